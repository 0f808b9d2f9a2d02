# A/B of variants inside one box / one device.  AB_CONFIGS: space-separated "ENV=VAL[,ENV=VAL]" settings.
for cfg in ${AB_CONFIGS:-"X=0"}; do
  echo "== $cfg"
  env $(echo $cfg | tr ';' ' ') timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | grep -E "gpu part|rror"
done
