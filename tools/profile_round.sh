#!/bin/bash
# Round profile on the GPU box: kernel-trace stats + three separate PMC passes (never combined with other traces),
# condensed into profiles/<round>_*.  Usage: bash tools/profile_round.sh r01
set -e
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$R
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
echo "stats done" 
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc1 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc1.log 2>&1
echo "pmc1 done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc2 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc2.log 2>&1
echo "pmc2 done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc3 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc3.log 2>&1
echo "pmc3 done"
python3 tools/profile_summary.py --round $R --stats $OUT/stats --steps 4 --pmc $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 --out $OUT/summary
ls -la $OUT/summary
