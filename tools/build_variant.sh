#!/bin/bash
# developer helper: tools/build_variant.sh <name> [extra hipcc flags] -> tools/variants/<name>/libm3t_hip.so (+ resource report)
set -e
name=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$REPO/tools/variants/$name" /tmp/m3t_build_$name
cd "$REPO/3dobjecttracking_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage -save-temps=obj "$@" -o "$REPO/tools/variants/$name/libm3t_hip.so" m3t_hip_api.hip > /tmp/m3t_build_$name/build.log 2>&1 || { grep -E "error" -A3 /tmp/m3t_build_$name/build.log | head -40; exit 1; }
mv "$REPO/tools/variants/$name"/*.s /tmp/m3t_build_$name/ 2>/dev/null || true
rm -f "$REPO/tools/variants/$name"/*.bc "$REPO/tools/variants/$name"/*.hipi "$REPO/tools/variants/$name"/*.o "$REPO/tools/variants/$name"/*.out "$REPO/tools/variants/$name"/*.txt "$REPO/tools/variants/$name"/*.hipfb
grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy" /tmp/m3t_build_$name/build.log | sed 's/.*remark: //; s/\[-Rpass.*//' | paste - - - - - | grep -E "tracking|rigid_opt|links|histogram_k" | awk '{print $3, "sgpr", $5, "vgpr", $7, "scratch", $10, "occ", $14}'
