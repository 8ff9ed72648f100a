#!/bin/bash
# developer helper: round 5, call 10 -- resolve with the transposed grid and packed stores; set-up with two workgroups per CU;
# renderer goldens
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${TAG:-r05j}
mkdir -p "$OUT"; cd "$REPO"
(cd tests && timeout 900 python -m pytest test_gpu_renderer.py -m gpu -x -q 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -5) | tee "$OUT/renderer_tests.log"
for v in "product 0" "setup4 4" "setup4 8"; do
  set -- $v
  lib=""; [ $1 != product ] && lib=$REPO/tools/variants/$1/libm3t_hip.so
  export M3T_HIP_RASTER_SLICES=$2; [ $2 = 0 ] && unset M3T_HIP_RASTER_SLICES
  (cd /tmp && export TMPDIR=/tmp && M3T_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/r_$1$2" -- python "$REPO/tools/render64_trace.py" 64 5 > "$OUT/r_$1$2.log" 2>&1)
  echo "$1 slices $2: $(grep objects $OUT/r_$1$2.log)"
  grep -E "focused|tracking" "$OUT"/r_$1$2/*/*kernel_stats.csv | cut -d, -f1-4
  rm -rf "$OUT/r_$1$2"
done | tee "$OUT/render64_after.txt"
for b in 4 16 32; do echo "bands $b: $(M3T_HIP_RASTER_BANDS=$b timeout 300 python tools/render64_trace.py 64 8 2>&1 | grep objects)"; done | tee -a "$OUT/render64_after.txt"
echo "1 object: $(timeout 300 python tools/render64_trace.py 1 20 2>&1 | grep objects)" | tee -a "$OUT/render64_after.txt"
