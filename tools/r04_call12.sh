#!/bin/bash
OUT=gpurun_out/r04l; mkdir -p $OUT
NEW=3dobjecttracking_amd/csrc/libm3t_hip.so
for p in 4 8 16; do (M3T_HIP_SPLIT_PARTS=$p timeout 200 python tools/raster_probe.py --step $NEW 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/parts $p: /") >> $OUT/raster_parts.txt; done; cat $OUT/raster_parts.txt
(timeout 600 python bench.py --extras --no-pcie --no-cpu-baseline --busy-seconds 1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps(d['extras'])); print(d['value'], d['ms_per_step'], json.dumps(d['roofline']))
") > $OUT/extras.txt 2>&1; cat $OUT/extras.txt
