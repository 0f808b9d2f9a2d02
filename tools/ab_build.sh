#!/bin/bash
# Build libvlhip.so from the kernel sources of another commit into clg-vqa_amd/csrc/ab_<tag>/ for same-box A/B runs:
#   bash tools/ab_build.sh HEAD~1 old && AB_CONFIGS="VLHIP_LIBRARY=clg-vqa_amd/csrc/ab_old/libvlhip.so X=new ..." bash tools/ab_bench.sh
# (the C ABI must be the same in both commits; step-level A/B on one device is the only comparison that counts)
set -e
REV=${1:?commit}; TAG=${2:-old}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/clg-vqa_amd/csrc/ab_$TAG
rm -rf "$OUT" && mkdir -p "$OUT/src/clg-vqa_amd" "$OUT/src/include"
git -C "$ROOT" archive "$REV" clg-vqa_amd/csrc include | tar -x -C "$OUT/src"
make -C "$OUT/src/clg-vqa_amd/csrc" -j4 > "$OUT/build.log" 2>&1
cp "$OUT/src/clg-vqa_amd/csrc/libvlhip.so" "$OUT/libvlhip.so"
rm -rf "$OUT/src"
ls -la "$OUT/libvlhip.so"
