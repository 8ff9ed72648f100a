#!/bin/bash
OUT=gpurun_out/r04b; mkdir -p $OUT
NEW=3dobjecttracking_amd/csrc/libm3t_hip.so; OLD=tools/variants/r03/libm3t_hip.so
(timeout 600 python -m pytest tests/test_gpu_roi.py tests/test_gpu_renderer.py tests/test_renderer_goldens.py "tests/test_gpu_benchmark_shape.py::test_compact_kernel_counts_saturated_background_pixels" "tests/test_gpu_benchmark_shape.py::test_headline_batch_is_bit_identical_to_the_oracle" tests/test_gpu_multibody.py -m gpu -q --timeout=600 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -40) > $OUT/tests.log 2>&1
tail -15 $OUT/tests.log
(timeout 200 python tools/phase_timing.py tools/libm3t_hip_timing.so 64 2>&1 | grep -v amdgpu) > $OUT/phase_timing_rbot64.txt; cat $OUT/phase_timing_rbot64.txt
(timeout 200 python tools/phase_timing.py tools/libm3t_hip_timing.so 21 ycb 2>&1 | grep -v amdgpu) > $OUT/phase_timing_ycb21.txt; head -32 $OUT/phase_timing_ycb21.txt
(timeout 200 python tools/tree_timing.py tools/libm3t_hip_timing.so 2>&1 | grep -v amdgpu) > $OUT/phase_timing_chain8.txt; cat $OUT/phase_timing_chain8.txt
(timeout 200 python tools/raster_probe.py --step $NEW 2>&1 | grep -v amdgpu) > $OUT/raster.txt
for b in 8 32; do (M3T_HIP_RASTER_BANDS=$b timeout 200 python tools/raster_probe.py --step $NEW 2>&1 | grep -v amdgpu | sed "s/^/bands $b: /") >> $OUT/raster.txt; done; cat $OUT/raster.txt
(timeout 200 python tools/quick_bench.py --objects 64 $NEW 2>&1 | grep -v amdgpu) > $OUT/quick_rbot64.txt; cat $OUT/quick_rbot64.txt
(timeout 400 python bench.py --busy-seconds 1 --no-cpu-baseline > $OUT/bench_rbot64.json 2> $OUT/bench_rbot64.err); python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r04b/bench_rbot64.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], json.dumps(d.get('pcie_inclusive'))[:1800])
except Exception as e:
    print('bench failed', e); print(open('gpurun_out/r04b/bench_rbot64.err').read()[-2000:])
PY
