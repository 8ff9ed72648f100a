#!/bin/bash
OUT=gpurun_out/r04h; mkdir -p $OUT
NEW=3dobjecttracking_amd/csrc/libm3t_hip.so
(timeout 900 python -m pytest tests/test_gpu_multibody.py tests/test_gpu_renderer.py tests/test_renderer_goldens.py tests/test_multibody_oracle.py -m gpu -q --timeout=600 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -40) > $OUT/tests.log 2>&1
tail -15 $OUT/tests.log
(timeout 200 python tools/tree_timing.py tools/libm3t_hip_timing.so 2>&1 | grep -v amdgpu) > $OUT/phase_timing_chain8.txt; cat $OUT/phase_timing_chain8.txt
for L in $NEW; do
  (M3T_HIP_LIBRARY=$L timeout 300 python bench.py --config chain8 --no-cpu-baseline --repeats 5 2>&1 | grep -v amdgpu | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$L', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['repeats'])
    else: print(l.rstrip()[:300])
") >> $OUT/chain8.txt 2>&1
done; cat $OUT/chain8.txt
(timeout 200 python tools/raster_probe.py --step $NEW 2>&1 | grep -v amdgpu) > $OUT/raster.txt

for b in 16; do (M3T_HIP_RASTER_BANDS=$b timeout 200 python tools/raster_probe.py --step $NEW 2>&1 | grep -v amdgpu | sed "s/^/bands $b: /") >> $OUT/raster.txt; done; cat $OUT/raster.txt
rm -rf /tmp/rast; (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rast -- python $GRAFT_REPO_ROOT/tools/raster_probe.py --step $GRAFT_REPO_ROOT/$NEW > /dev/null 2>&1); cat /tmp/rast/*/*kernel_stats.csv | head -8 > $OUT/raster_stats.txt; cat $OUT/raster_stats.txt

