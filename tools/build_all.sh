#!/bin/bash
# developer helper: product lib (+resource report) and the -DM3T_PHASE_TIMING variant
set -e
cd "$(dirname "$0")/../3dobjecttracking_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-function -Wno-unused-variable"
hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -o libm3t_hip.so m3t_hip_api.hip 2>&1 | grep -E "error|Function Name|VGPRs:|ScratchSize|SGPRs Spill" | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/.*remark: //' | paste - - - - | grep -E "error|tracking|region_corr|histogram|gradient" || true
hipcc $FLAGS -DM3T_PHASE_TIMING -o ../../tools/libm3t_hip_timing.so m3t_hip_api.hip 2>&1 | grep -E "error" || true
