#!/bin/bash
OUT=gpurun_out/r04m; mkdir -p $OUT
NEW=3dobjecttracking_amd/csrc/libm3t_hip.so
for cfg in "32 64" "64 64" "100 64" "32 128" "64 128" "64 256"; do set -- $cfg
  (M3T_HIP_RASTER_BANDS=$1 M3T_HIP_RASTER_SLICES=$2 M3T_HIP_SPLIT_PARTS=16 timeout 200 python tools/raster_probe.py --step $NEW 2>&1 | grep -v amdgpu | tr '\n' ' ' | sed "s/^/bands $1 slices $2 parts 16: /"; echo) >> $OUT/raster_knobs.txt
done; cat $OUT/raster_knobs.txt
