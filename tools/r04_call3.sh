#!/bin/bash
OUT=gpurun_out/r04c; mkdir -p $OUT
NEW=3dobjecttracking_amd/csrc/libm3t_hip.so
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benchmark_shape.py tests/test_gpu_split.py tests/test_gpu_edge_cases.py tests/test_gpu_multibody.py tests/test_cpp_adapter.py tests/test_gpu_roi.py -m gpu -q --timeout=600 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -40) > $OUT/tests.log 2>&1
tail -15 $OUT/tests.log
(timeout 300 python tools/quick_bench.py --objects 64 tools/variants/pre/libm3t_hip.so $NEW tools/variants/noguess/libm3t_hip.so 2>&1 | grep -v amdgpu) > $OUT/quick_rbot64.txt; cat $OUT/quick_rbot64.txt
(timeout 300 python tools/quick_bench.py --ycb --objects 21 tools/variants/pre/libm3t_hip.so $NEW 2>&1 | grep -v amdgpu) > $OUT/quick_ycb21.txt; cat $OUT/quick_ycb21.txt
(timeout 200 python tools/phase_timing.py tools/libm3t_hip_timing.so 64 2>&1 | grep -v amdgpu) > $OUT/phase_timing_rbot64.txt; head -30 $OUT/phase_timing_rbot64.txt
# ROI: does the pull kernel run while the step runs?
for L in $NEW tools/variants/noguess/libm3t_hip.so; do
  rm -rf /tmp/roitrace; (cd /tmp && export TMPDIR=/tmp && M3T_HIP_LIBRARY=$GRAFT_REPO_ROOT/$L timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/roitrace -- python $GRAFT_REPO_ROOT/tools/roi_trace.py 2>&1 | grep "roi loop")
  echo "== $L" >> $OUT/roi_trace.txt; python tools/roi_trace.py --report /tmp/roitrace >> $OUT/roi_trace.txt 2>&1
done; cat $OUT/roi_trace.txt
# renderer-fed step: per-kernel durations, band resolve vs the three-launch form
for mode in bands three; do
  rm -rf /tmp/rast; env=""; [ $mode = three ] && env="M3T_HIP_NO_LDS_RASTER=1"
  (cd /tmp && export TMPDIR=/tmp && env $env timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rast -- python $GRAFT_REPO_ROOT/tools/raster_probe.py --step $GRAFT_REPO_ROOT/$NEW > /dev/null 2>&1)
  echo "== $mode" >> $OUT/raster_stats.txt; cat /tmp/rast/*/*kernel_stats.csv | head -12 >> $OUT/raster_stats.txt
done; cat $OUT/raster_stats.txt
