#!/bin/bash
# Register, scratch and occupancy figures of every kernel of mdc_kernels.hip (compiler remarks).
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops \
  -I include -I dsp_amd/csrc -c dsp_amd/csrc/mdc_kernels.hip -o /tmp/mdc_k.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - |
  sed -E 's/Function Name: _ZN3mdc[0-9]+//; s/ENS_10TileParamsE//'
