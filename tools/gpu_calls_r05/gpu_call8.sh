#!/bin/bash
# developer helper: round 5, call 8 -- work spread of the focused renderers at 128 pairs (64-object renderer-fed step)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${TAG:-r05h}
mkdir -p "$OUT"; cd "$REPO"
for knobs in "32 8" "2 8" "4 8" "8 8" "2 4" "2 2" "2 16" "4 4" "1 8"; do
  set -- $knobs
  echo "slices $1 bands $2: $(M3T_HIP_RASTER_SLICES=$1 M3T_HIP_RASTER_BANDS=$2 timeout 300 python tools/render64_trace.py 64 8 2>&1 | grep objects)" | tee -a "$OUT/render64_knobs.txt"
done
