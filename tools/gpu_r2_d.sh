#!/bin/bash
# round 2, GPU call D: model generation with associated bodies, deferred occlusion vote, shared view search; big batches
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r2d
mkdir -p "$OUT"
cd "$REPO"
(cd tests && timeout 1500 python -m pytest -m gpu -q -x --timeout=900 test_gpu_model_generation.py test_gpu_split.py test_gpu_parity.py test_gpu_edge_cases.py test_gpu_benchmark_shape.py test_cpp_adapter.py 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -40) > "$OUT/tests_core.log" 2>&1
(timeout 300 python tools/phase_timing.py tools/libm3t_hip_timing.so 21 ycb 2>&1 | tail -34) > "$OUT/phase_ycb.log" 2>&1
(timeout 600 python tools/sweep_shapes.py rbot64 ycb 2>&1 | grep -E "^\{|Error|error|Traceback" ) > "$OUT/sweep.log" 2>&1
(timeout 900 python bench.py --no-pcie --no-cpu-baseline --sweep 256,512,1024,4096 > "$OUT/bench_sweep.json" 2> "$OUT/bench_sweep.err")
tail -8 "$OUT/tests_core.log"; cat "$OUT/sweep.log"; cat "$OUT/phase_ycb.log"; python -c "
import json; d=json.load(open('$OUT/bench_sweep.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d.get('batch_sweep'), indent=0))"
