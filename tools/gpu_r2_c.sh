#!/bin/bash
# round 2, GPU call C: wave-per-structure link kernels, chain leg, third kernel iteration
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r2c
mkdir -p "$OUT"
cd "$REPO"
(cd tests && timeout 900 python -m pytest -m gpu -q -x --timeout=600 test_gpu_multibody.py test_gpu_split.py test_gpu_parity.py test_gpu_generator.py 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -40) > "$OUT/tests_core.log" 2>&1
(timeout 300 python tools/phase_timing.py tools/libm3t_hip_timing.so 64 2>&1 | tail -34) > "$OUT/phase_split.log" 2>&1
(timeout 600 python tools/sweep_shapes.py rbot64 ycb 2>&1 | grep -E "^\{|Error|error|Traceback" ) > "$OUT/sweep.log" 2>&1
(timeout 600 python bench.py --config chain8 > "$OUT/bench_chain8.json" 2> "$OUT/bench_chain8.err")
(timeout 600 python bench.py --no-pcie > "$OUT/bench_default.json" 2> "$OUT/bench_default.err")
tail -5 "$OUT/tests_core.log"; cat "$OUT/sweep.log"; cat "$OUT/phase_split.log"; head -c 2500 "$OUT/bench_chain8.json"; echo; tail -5 "$OUT/bench_chain8.err"; head -c 400 "$OUT/bench_default.json"
