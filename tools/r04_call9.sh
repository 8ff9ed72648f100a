#!/bin/bash
OUT=gpurun_out/r04i; mkdir -p $OUT
NEW=3dobjecttracking_amd/csrc/libm3t_hip.so
(timeout 900 python -m pytest tests/test_gpu_benchmark_shape.py tests/test_gpu_multibody.py tests/test_gpu_edge_cases.py -m gpu -q --timeout=600 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -30) > $OUT/tests.log 2>&1
tail -8 $OUT/tests.log
(timeout 600 python tools/quick_bench.py --objects 512,4096 tools/variants/pre/libm3t_hip.so $NEW 2>&1 | grep -v amdgpu) > $OUT/quick_rbot_large.txt; cat $OUT/quick_rbot_large.txt
(timeout 600 python tools/quick_bench.py --ycb --objects 512 tools/variants/pre/libm3t_hip.so $NEW 2>&1 | grep -v amdgpu) > $OUT/quick_synth512.txt; cat $OUT/quick_synth512.txt
(M3T_HIP_LIBRARY=$NEW timeout 300 python bench.py --config chain8 --no-cpu-baseline --repeats 5 2>&1 | grep -v amdgpu | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'], d['roofline'])
") > $OUT/chain8.txt 2>&1; cat $OUT/chain8.txt
