#!/bin/bash
OUT=gpurun_out/r04k; mkdir -p $OUT
NEW=3dobjecttracking_amd/csrc/libm3t_hip.so
(timeout 400 python tools/quick_bench.py --objects 64 $NEW 2>&1 | grep -v amdgpu) > $OUT/quick_w4.txt
(timeout 400 python tools/quick_bench.py --objects 64 --env "M3T_HIP_SPLIT_PER_CU=1;" tools/variants/w4/libm3t_hip.so 2>&1 | grep -v amdgpu) >> $OUT/quick_w4.txt
(timeout 400 python tools/quick_bench.py --objects 64,32 $NEW 2>&1 | grep -v amdgpu) >> $OUT/quick_w4.txt
cat $OUT/quick_w4.txt
