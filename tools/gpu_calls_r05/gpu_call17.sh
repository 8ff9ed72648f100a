#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${TAG:-r05q}
mkdir -p "$OUT"; cd "$REPO"
for knobs in "2 8" "2 4" "2 5" "2 10" "2 16" "3 8" "4 8"; do
  set -- $knobs
  echo "slices $1 bands $2: $(M3T_HIP_RASTER_SLICES=$1 M3T_HIP_RASTER_BANDS=$2 timeout 300 python tools/render64_trace.py 64 20 2>&1 | grep objects)" | tee -a "$OUT/render64_knobs2.txt"
done
