#!/bin/bash
# round 4, first GPU call: the merged r04-prep work against round 3's library (tools/variants/r03)
OUT=gpurun_out/r04a; mkdir -p $OUT
NEW=3dobjecttracking_amd/csrc/libm3t_hip.so; OLD=tools/variants/r03/libm3t_hip.so
(timeout 900 python -m pytest tests -m gpu -q --timeout=600 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -40) > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
(timeout 300 python tools/quick_bench.py --objects 64 $OLD $NEW 2>&1 | grep -v amdgpu) > $OUT/quick_rbot64.txt; cat $OUT/quick_rbot64.txt
(timeout 300 python tools/quick_bench.py --ycb --objects 21 $OLD $NEW 2>&1 | grep -v amdgpu) > $OUT/quick_ycb21.txt; cat $OUT/quick_ycb21.txt
(timeout 300 python tools/raster_probe.py --step $OLD $NEW 2>&1 | grep -v amdgpu) > $OUT/raster.txt
(M3T_HIP_NO_LDS_RASTER=1 timeout 300 python tools/raster_probe.py --step $NEW 2>&1 | grep -v amdgpu) >> $OUT/raster.txt; cat $OUT/raster.txt
for L in $OLD $NEW; do
  (M3T_HIP_LIBRARY=$L timeout 300 python bench.py --config chain8 --no-cpu-baseline --repeats 5 2>&1 | grep -v amdgpu | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$L', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['repeats'])
    else: print(l.rstrip()[:300])
") >> $OUT/chain8.txt 2>&1
done; cat $OUT/chain8.txt
(timeout 600 python bench.py --busy-seconds 2 --cpu-seconds 3 --no-cpu-parallel > $OUT/bench_rbot64.json 2> $OUT/bench_rbot64.err); python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r04a/bench_rbot64.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], json.dumps(d.get('pcie_inclusive'))[:1500])
except Exception as e:
    print('bench failed', e); print(open('gpurun_out/r04a/bench_rbot64.err').read()[-2000:])
PY
