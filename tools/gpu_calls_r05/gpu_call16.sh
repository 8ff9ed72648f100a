#!/bin/bash
# developer helper: round 5, call 16 -- all GPU tests; the renderer-fed steps with the round's rasteriser changes
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${TAG:-r05p}
mkdir -p "$OUT"; cd "$REPO"
export M3T_INPUT_CACHE=${M3T_INPUT_CACHE:-/tmp/m3t_inputs_$(id -u)} M3T_INPUT_WORKERS=auto
(cd tests && timeout 1500 python -m pytest -m gpu -q --timeout=900 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -12) > "$OUT/gpu_tests.log" 2>&1
tail -4 "$OUT/gpu_tests.log"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/r" -- python "$REPO/tools/render64_trace.py" 64 10 > "$OUT/r.log" 2>&1)
echo "under rocprofv3: $(grep objects $OUT/r.log)" | tee "$OUT/render64.txt"
grep -E "focused|tracking|histogram" "$OUT"/r/*/*kernel_stats.csv | cut -d, -f1-4 | tee -a "$OUT/render64.txt"
cp "$OUT"/r/*/*kernel_stats.csv "$OUT/render64_kernel_stats.csv"; rm -rf "$OUT/r"
echo "$(timeout 300 python tools/render64_trace.py 64 20 2>&1 | grep objects)" | tee -a "$OUT/render64.txt"
echo "$(timeout 300 python tools/render64_trace.py 1 20 2>&1 | grep objects)" | tee -a "$OUT/render64.txt"
