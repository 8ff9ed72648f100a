#!/bin/bash
# developer helper: round 5, call 18 -- the bodies of a structure split over workgroups (tracking_step_tree_split_kernel)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${TAG:-r05r}
mkdir -p "$OUT"; cd "$REPO"
export M3T_INPUT_CACHE=${M3T_INPUT_CACHE:-/tmp/m3t_inputs_$(id -u)} M3T_INPUT_WORKERS=auto
(cd tests && timeout 900 python -m pytest test_gpu_multibody.py -m gpu -x -q 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -25) > "$OUT/multibody.log" 2>&1
tail -12 "$OUT/multibody.log"
NEW=3dobjecttracking_amd/csrc/libm3t_hip.so
for p in 1 2 4 8; do
  echo "parts $p: $(M3T_HIP_TREE_PARTS=$p timeout 300 python tools/chain_bench.py --oracle $NEW 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | grep "one launch")"
done | tee "$OUT/chain_parts.txt"
