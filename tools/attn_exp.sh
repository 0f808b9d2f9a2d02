#!/bin/bash
# Where the attention forward kernel's time goes: timing builds with WRONG results (attention2.hip VL_EXP_ATTN bits: 1 = no input
# loads, 2 = no arithmetic between staging and the output stores, 4 = no output stores).  Build on the build host:
#   cd clg-vqa_amd/csrc && for v in 1 2 4 3 6 7; do make variant1 VARIANT=ax$v FILE=attention2.hip EXTRA=-DVL_EXP_ATTN=$v; done
# then on the GPU box: bash tools/attn_exp.sh   (forward columns only; the backward kernel is unchanged in these builds)
cd "$(dirname "$0")/.."
echo "== complete"; python3 tools/attn_bench.py 2>/dev/null | grep "p=0.1"
for v in 1 2 4 3 6 7; do
  echo "== VL_EXP_ATTN=$v"
  VLHIP_LIBRARY=clg-vqa_amd/csrc/ab_ax$v/libvlhip.so python3 tools/attn_bench.py 2>/dev/null | grep "p=0.1" | sed 's/  bwd.*//'
done
