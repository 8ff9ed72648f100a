#!/bin/bash
# round 2, GPU call E: grouped occlusion windows; full GPU test suite
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r2e
mkdir -p "$OUT"
cd "$REPO"
(cd tests && timeout 1800 python -m pytest -m gpu -q --timeout=900 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -40) > "$OUT/tests_all.log" 2>&1
(timeout 300 python tools/phase_timing.py tools/libm3t_hip_timing.so 21 ycb 2>&1 | tail -34) > "$OUT/phase_ycb.log" 2>&1
(timeout 600 python tools/sweep_shapes.py rbot64 ycb 2>&1 | grep -E "^\{|Error|error|Traceback" ) > "$OUT/sweep.log" 2>&1
tail -12 "$OUT/tests_all.log"; cat "$OUT/sweep.log"; cat "$OUT/phase_ycb.log"
