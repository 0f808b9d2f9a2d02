#!/bin/bash
# Round profile on the GPU box: kernel-trace stats + separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one
# pass on gfx950; counters are never combined with other trace domains), condensed into gpurun_out/<round>/summary.
# Usage: bash tools/profile_round.sh r02 [workload]     (then copy gpurun_out/r02/summary/* into profiles/)
R=${1:-r02}
W=${2:-c2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$R
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1 || { echo "stats pass failed"; tail -3 $OUT/stats.log; exit 1; }
echo "stats done"
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i ($ctr) failed"; grep -m3 -i "error\|fail" $OUT/pmc$i.log; exit 1; }
  echo "pmc$i ($ctr) done"
done
python3 tools/profile_summary.py --round $R --stats $OUT/stats --steps 4 --pmc $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 --out $OUT/summary
ls -la $OUT/summary
