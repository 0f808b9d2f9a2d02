#!/bin/bash
# register / scratch usage of the kernels of one source file: bash tools/regs.sh dw.hip [filter] [extra flags]
cd "$(dirname "$0")/../clg-vqa_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $3 -Rpass-analysis=kernel-resource-usage -c $1 -o /tmp/regs_$$.o 2>&1 \
  | grep -E "error|Function Name|VGPRs:|ScratchSize|VGPRs Spill|Occupancy" | sed 's/\[-Rpass.*//; s/.*remark: *//' | paste - - - - - | grep -E "${2:-.}" | awk '{print $3, $4,$5, $6,$7,$8, $9,$10,$11,$12,$13,$14,$15,$16}'
rm -f /tmp/regs_$$.o
