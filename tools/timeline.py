#!/usr/bin/env python3
"""Critical-path view of one training step from a rocprofv3 kernel trace: per stream, kernel time by name, idle gaps."""
import collections
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    f = (glob.glob(d + "/*kernel_trace.csv") + glob.glob(d + "/*/*kernel_trace.csv"))[0]
    rows = []
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r["Stream_Id"], r["Grid_Size_X"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[2].startswith("adamw_kernel")]
    a, b = marks[-2], marks[-1]
    step = rows[a + 1:b + 1]
    t0, t1 = rows[a][1], rows[b][1]
    print("step span %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)))
    streams = collections.Counter(r[3] for r in step)
    main_stream = streams.most_common(1)[0][0]
    for sid, cnt in streams.most_common():
        ks = [r for r in step if r[3] == sid]
        busy = sum(r[1] - r[0] for r in ks)
        print("stream %s: %d kernels, busy %.3f ms%s" % (sid, cnt, busy / 1e6, "  (main)" if sid == main_stream else ""))
        agg = collections.defaultdict(lambda: [0, 0])
        for r in ks:
            agg[(r[2], r[4])][0] += 1
            agg[(r[2], r[4])][1] += r[1] - r[0]
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print("    %-50s grid %-8s x%-3d avg %7.1f us  total %6.3f ms" % (k[0], k[1], v[0], v[1] / v[0] / 1e3, v[1] / 1e6))
    # idle on the main stream
    ks = [r for r in step if r[3] == main_stream]
    gaps = []
    prev = t0
    for r in ks:
        if r[0] > prev:
            gaps.append((r[0] - prev, r[2]))
        prev = max(prev, r[1])
    print("main-stream idle %.3f ms in %d gaps; largest:" % (sum(g[0] for g in gaps) / 1e6, len(gaps)))
    for g in sorted(gaps, reverse=True)[:8]:
        print("    %.1f us before %s" % (g[0] / 1e3, g[1]))
    # forward / backward split: first attn_bwd marks the backward
    fb = next(r for r in step if r[2].startswith("attn2_bwd") or r[2].startswith("attn_bwd"))
    lf = [r for r in step if r[0] < fb[0] and (r[2].startswith("attn2_fwd") or r[2].startswith("attn_fwd"))][-1]
    print("forward+head span ~%.3f ms (to last attn_fwd end), backward+opt span ~%.3f ms" % ((lf[1] - t0) / 1e6, (t1 - lf[1]) / 1e6))


if __name__ == "__main__":
    main()
