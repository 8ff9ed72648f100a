#!/bin/bash
# developer helper: register / scratch use of every kernel of the product library (extra hipcc flags as arguments)
cd "$(dirname "$0")/../3dobjecttracking_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -w -Rpass-analysis=kernel-resource-usage "$@" -o /tmp/m3t_resources.so m3t_hip_api.hip 2>&1 | grep -E "error|Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: //; s/\[-Rpass.*//' | paste - - - - - - | awk '{printf "%-44s vgpr %-4s agpr %-4s scratch %-5s\n", $3, $5, $7, $10}'
