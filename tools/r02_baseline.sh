#!/bin/bash
# Round-2 baseline on the GPU box (code of round 1): available counters, bench lines of the other BASELINE configs,
# one MFMA-counter pass.  Output under gpurun_out/r02a/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02a
mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || rocprofv3 -L > $OUT/avail.txt 2>&1
grep -i -n "mfma" $OUT/avail.txt | head -50 > $OUT/avail_mfma.txt
for w in c2 c3 c4 c5; do
  timeout -k 10 300 python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err || { echo "bench $w failed"; tail -5 $OUT/bench_$w.err; exit 1; }
  echo "bench $w done: $(cut -c1-160 $OUT/bench_$w.json)"
done
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_mfma.log 2>&1 || { echo "mfma pmc pass failed"; tail -5 $OUT/pmc_mfma.log; }
ls $OUT
