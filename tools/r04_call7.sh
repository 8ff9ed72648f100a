#!/bin/bash
OUT=gpurun_out/r04g; mkdir -p $OUT
NEW=3dobjecttracking_amd/csrc/libm3t_hip.so
(timeout 900 python -m pytest tests/test_gpu_renderer.py tests/test_renderer_goldens.py tests/test_gpu_benchmark_shape.py::test_headline_batch_is_bit_identical_to_the_oracle tests/test_gpu_benchmark_shape.py::test_ycb_batch_is_bit_identical_to_the_oracle tests/test_gpu_parity.py tests/test_gpu_split.py -m gpu -q --timeout=600 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -30) > $OUT/tests.log 2>&1
tail -12 $OUT/tests.log
(timeout 300 python tools/quick_bench.py --objects 64 tools/variants/pre/libm3t_hip.so $NEW 2>&1 | grep -v amdgpu) > $OUT/quick_rbot64.txt; cat $OUT/quick_rbot64.txt
(timeout 300 python tools/quick_bench.py --ycb --objects 21 tools/variants/pre/libm3t_hip.so $NEW 2>&1 | grep -v amdgpu) > $OUT/quick_ycb21.txt; cat $OUT/quick_ycb21.txt
(timeout 200 python tools/phase_timing.py tools/libm3t_hip_timing.so 64 2>&1 | grep -v amdgpu | head -30) > $OUT/phase_timing_rbot64.txt; grep -E "chain|solve \(|total" $OUT/phase_timing_rbot64.txt
(timeout 200 python tools/raster_probe.py --step $NEW 2>&1 | grep -v amdgpu) > $OUT/raster.txt; cat $OUT/raster.txt
rm -rf /tmp/rast; (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rast -- python $GRAFT_REPO_ROOT/tools/raster_probe.py --step $GRAFT_REPO_ROOT/$NEW > /dev/null 2>&1); cat /tmp/rast/*/*kernel_stats.csv | head -8 > $OUT/raster_stats.txt; cat $OUT/raster_stats.txt
