#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/...) into the small tracked files under profiles/.

    python tools/profile_summary.py --round r01 --stats gpurun_out/prof7 --steps 4 \
        --pmc gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3

* <round>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, per kernel (calls, total, average ns)
* <round>_summary.json       per-step milliseconds per kernel + PMC-derived HBM traffic per launch (FETCH_SIZE is
                             doubled on gfx950 as /opt/skills/guides/MI355X_MICROARCH.md prescribes; KiB units)
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", required=True)
    ap.add_argument("--stats", required=True)
    ap.add_argument("--steps", type=float, default=4, help="training steps covered by the stats run (warm-up + timed)")
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--out", default="profiles")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    f = glob.glob(os.path.join(args.stats, "*", "*kernel_stats.csv"))[0]
    shutil.copy(f, os.path.join(args.out, "%s_kernel_stats.csv" % args.round))
    rows = list(csv.DictReader(open(f)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    kernels = []
    for r in rows:
        kernels.append(dict(kernel=short(r["Name"]), calls_per_step=float(r["Calls"]) / args.steps,
                            ms_per_step=float(r["TotalDurationNs"]) / 1e6 / args.steps,
                            avg_us=float(r["AverageNs"]) / 1e3, pct=100 * float(r["TotalDurationNs"]) / total))
    pmc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for d in args.pmc:
        for f2 in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f2)):
                k = short(r["Kernel_Name"])
                pmc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[k][r["Counter_Name"]] += 1
    traffic = {}
    for k, c in pmc.items():
        n = max(cnt[k].values())
        fetch = c.get("FETCH_SIZE", 0.0) * 1024 * 2 / n   # gfx950: FETCH_SIZE counts 64 B per 128-B request
        write = c.get("WRITE_SIZE", 0.0) * 1024 / n
        hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
        traffic[k] = dict(launches=n, hbm_read_MB_per_launch=fetch / 1e6, hbm_write_MB_per_launch=write / 1e6,
                          l2_hit_rate=hit / max(hit + miss, 1.0),
                          lds_bank_conflict_frac=c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0),
                          wait_any_frac=c.get("SQ_WAIT_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0),
                          # rocprofv3's MfmaUtil: sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE of one XCD * 1024 SIMDs);
                          # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
                          mfma_busy_frac=(c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(c.get("GRBM_GUI_ACTIVE", 0.0) / 8 * 1024, 1.0)
                                          if "SQ_VALU_MFMA_BUSY_CYCLES" in c else None))
    out = dict(round=args.round, gpu_ms_per_step=total / 1e6 / args.steps, kernels=kernels, pmc=traffic)
    with open(os.path.join(args.out, "%s_summary.json" % args.round), "w") as fo:
        json.dump(out, fo, indent=1)
    print("wrote", args.out, "total %.2f ms/step" % out["gpu_ms_per_step"])


if __name__ == "__main__":
    main()
