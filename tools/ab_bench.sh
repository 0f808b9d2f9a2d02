# A/B of GEMM kernel variants inside one box / one device (VL_DEBUG = vl_debug_set key:value pairs)
for cfg in ${AB_CONFIGS:-"7:0" "7:1,8:0" "7:2,8:0" "7:1,8:1" "7:0" "7:1,8:1"}; do
  echo "== $cfg"
  VL_DEBUG=$cfg timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -E "gpu part|rror"
done
