#!/usr/bin/env python3
"""Kernels between the last attention forward and the first full-size backward LayerNorm of one training step (the
pooled last layer + task head + loss window) from a rocprofv3 kernel trace.  Usage: head_window.py <kernel_trace.csv>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
step = rows[idx[-2] + 1:idx[-1] + 1]
t0 = int(step[0]["Start_Timestamp"])
names = [r["Kernel_Name"] for r in step]
la = max(i for i, n in enumerate(names) if "attn2_fwd" in n)
fb = min(i for i, n in enumerate(names) if "attn2_bwd" in n)
tot = 0.0
for r in step[la:fb + 1]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    e = (int(r["End_Timestamp"]) - t0) / 1e3
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"])[:64]
    print("%9.1f %7.1f s%s g%8s %s" % (s, e - s, r["Stream_Id"], r["Grid_Size_X"], n))
    tot += e - s
w0, w1 = int(step[la]["Start_Timestamp"]), int(step[fb]["End_Timestamp"])
print("window %.1f us, kernels %d, busy %.1f us; step %.1f us" % ((w1 - w0) / 1e3, fb - la + 1, tot,
      (int(step[-1]["End_Timestamp"]) - t0) / 1e3))
