#!/usr/bin/env python3
"""Per-kernel / per-grid breakdown of a rocprofv3 --kernel-trace CSV directory (average us, ms per step)."""
import collections
import csv
import glob
import sys


def main():
    d, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    f = (glob.glob(d + "/*kernel_trace.csv") + glob.glob(d + "/*/*kernel_trace.csv"))[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    t0, t1 = None, None
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        agg[(n[:56], r["Grid_Size_X"])][0] += 1
        agg[(n[:56], r["Grid_Size_X"])][1] += (e - s) / 1e3
    tot = sum(v[1] for v in agg.values())
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
        print("%-58s grid %-8s calls/step %6.1f avg %7.1f us  ms/step %6.3f" % (k[0], k[1], v[0] / steps, v[1] / v[0], v[1] / steps / 1e3))
    print("sum of kernel time per step: %.3f ms" % (tot / steps / 1e3))


if __name__ == "__main__":
    main()
