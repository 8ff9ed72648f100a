#!/bin/bash
# developer helper: round 5, call 2 -- tracking_step_split2_kernel (128 VGPRs, two workgroups per CU)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${TAG:-r05b}
mkdir -p "$OUT"; cd "$REPO"
export M3T_INPUT_CACHE=${M3T_INPUT_CACHE:-/tmp/m3t_inputs_$(id -u)} M3T_INPUT_WORKERS=auto
OLD=tools/variants/r04/libm3t_hip.so; NEW=3dobjecttracking_amd/csrc/libm3t_hip.so
# correctness of the split2 kernel: the split / parity / edge-case tests with it forced on
(cd tests && M3T_HIP_SPLIT2=1 timeout 900 python -m pytest test_gpu_split.py test_gpu_parity.py test_gpu_edge_cases.py test_gpu_benchmark_shape.py -m gpu -q 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -30) > "$OUT/tests_split2.log" 2>&1
tail -8 "$OUT/tests_split2.log"
timeout 600 python tools/quick_bench.py --objects 64 --env ";M3T_HIP_SPLIT2=1;M3T_HIP_THREADS=256,M3T_HIP_SPLIT_PARTS=8" $OLD $NEW $NEW > "$OUT/qb_rbot64.txt" 2>&1
cat "$OUT/qb_rbot64.txt"
timeout 600 python tools/quick_bench.py --objects 32,48,96,128 --env ";M3T_HIP_SPLIT2=1" $NEW > "$OUT/qb_rbot_other.txt" 2>&1
cat "$OUT/qb_rbot_other.txt"
timeout 600 python tools/quick_bench.py --ycb --objects 21,64 --env ";M3T_HIP_SPLIT2=1" $NEW > "$OUT/qb_ycb.txt" 2>&1
cat "$OUT/qb_ycb.txt"
timeout 300 python tools/chain_bench.py --oracle $OLD $NEW 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" > "$OUT/chain_ab.txt"
cat "$OUT/chain_ab.txt"
timeout 200 python tools/tree_timing.py tools/libm3t_hip_timing.so 2>&1 | grep -v amdgpu > "$OUT/tree_timing_new.txt"
cat "$OUT/tree_timing_new.txt"
M3T_HIP_SPLIT2=1 timeout 300 python tools/phase_timing.py tools/libm3t_hip_timing.so 64 2>&1 | grep -v amdgpu > "$OUT/phase_timing_rbot64_split2.txt"
head -32 "$OUT/phase_timing_rbot64_split2.txt"
