#!/bin/bash
OUT=gpurun_out/r04f; mkdir -p $OUT
NEW=3dobjecttracking_amd/csrc/libm3t_hip.so
(timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_benchmark_shape.py tests/test_gpu_edge_cases.py tests/test_gpu_parity.py -m gpu -q --timeout=600 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -30) > $OUT/tests.log 2>&1
tail -8 $OUT/tests.log
(timeout 300 python tools/quick_bench.py --objects 64 tools/variants/pre/libm3t_hip.so $NEW 2>&1 | grep -v amdgpu) > $OUT/quick_rbot64.txt; cat $OUT/quick_rbot64.txt
(timeout 300 python tools/quick_bench.py --ycb --objects 21 tools/variants/pre/libm3t_hip.so $NEW 2>&1 | grep -v amdgpu) > $OUT/quick_ycb21.txt; cat $OUT/quick_ycb21.txt
(timeout 200 python tools/phase_timing.py tools/libm3t_hip_timing.so 64 2>&1 | grep -v amdgpu) > $OUT/phase_timing_rbot64.txt; cat $OUT/phase_timing_rbot64.txt
(timeout 200 python tools/raster_probe.py --step $NEW tools/variants/slices64/libm3t_hip.so 2>&1 | grep -v amdgpu) > $OUT/raster.txt; cat $OUT/raster.txt
