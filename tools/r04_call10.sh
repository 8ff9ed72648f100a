#!/bin/bash
OUT=gpurun_out/r04j; mkdir -p $OUT
NEW=3dobjecttracking_amd/csrc/libm3t_hip.so
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benchmark_shape.py tests/test_modality_goldens.py tests/test_gpu_generator.py -m gpu -q --timeout=600 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -30) > $OUT/tests.log 2>&1
tail -8 $OUT/tests.log
(timeout 600 python tools/quick_bench.py --ycb --objects 21,512 tools/variants/pre/libm3t_hip.so $NEW 2>&1 | grep -v amdgpu) > $OUT/quick_ycb.txt; cat $OUT/quick_ycb.txt
