#!/bin/bash
OUT=gpurun_out/r04n; mkdir -p $OUT
NEW=3dobjecttracking_amd/csrc/libm3t_hip.so
(timeout 600 python -m pytest tests/test_gpu_renderer.py tests/test_renderer_goldens.py -m gpu -q --timeout=600 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -10) > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
(timeout 200 python tools/raster_probe.py --step $NEW 2>&1 | grep -v amdgpu) > $OUT/raster.txt; cat $OUT/raster.txt
rm -rf /tmp/rast; (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rast -- python $GRAFT_REPO_ROOT/tools/raster_probe.py --step $GRAFT_REPO_ROOT/$NEW > /dev/null 2>&1); cat /tmp/rast/*/*kernel_stats.csv | head -8 > $OUT/raster_stats.txt; cat $OUT/raster_stats.txt
