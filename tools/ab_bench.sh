# A/B of variants inside one box / one device.  AB_CONFIGS: space-separated "ENV=VAL[;ENV=VAL]" settings.
# Prints ms / step and the live roofline fractions (forward 3-pass GEMM, fused attention op, backward dX GEMMs) per config.
for cfg in ${AB_CONFIGS:-"X=0"}; do
  echo "== $cfg"
  env $(echo $cfg | tr ';' ' ') timeout -k 10 300 python bench.py --steps ${AB_STEPS:-20} --warmup 5 --no-cpu-baseline --no-extras ${AB_ARGS:-} 2>/tmp/ab_err.txt | python3 -c "
import json, sys
try:
    d = json.loads(sys.stdin.read().strip().split('\n')[-1])
    r = d.get('roofline') or {}
    print('   %.3f ms/step  fwd3 %.4f (%.1f us)  qkv+attn %.4f  bwd1 %.4f' % (d['ms_per_step'], r.get('frac', 0), r.get('avg_launch_us', 0),
          (r.get('fused_attention') or {}).get('frac', 0), (r.get('backward_gemm') or {}).get('frac', 0)))
except Exception as e:
    print('   failed:', e)
"
  grep -E "rror|Traceback" /tmp/ab_err.txt | head -3
done
