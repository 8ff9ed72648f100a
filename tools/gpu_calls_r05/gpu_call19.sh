#!/bin/bash
# developer helper: round 5, call 19 -- the split tree kernel: tests, then the chain's part of the collection again
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r05s
mkdir -p "$OUT"; cd "$REPO"
export M3T_INPUT_CACHE=${M3T_INPUT_CACHE:-/tmp/m3t_inputs_$(id -u)} M3T_INPUT_WORKERS=auto
(cd tests && timeout 900 python -m pytest test_gpu_multibody.py test_gpu_multigpu.py test_gpu_bench_line.py -m gpu -q 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -25) > "$OUT/multibody.log" 2>&1
tail -6 "$OUT/multibody.log"
timeout 300 python tools/chain_bench.py --oracle --distributed tools/variants/r04/libm3t_hip.so 3dobjecttracking_amd/csrc/libm3t_hip.so 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tee "$OUT/chain_ab.txt"
ROUND=r05s CONFIGS=chain8 PMC_CONFIGS=chain8 RANKSHARE_CONFIGS=chain8 bash tools/collect_profiles.sh pmc bench stats phases rankshare 2>&1 | tail -40
