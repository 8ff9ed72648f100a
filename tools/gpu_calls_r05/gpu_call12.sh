#!/bin/bash
# developer helper: round 5, call 12 -- set-up kernel with the vertex table in LDS
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${TAG:-r05l}
mkdir -p "$OUT"; cd "$REPO"
(cd tests && timeout 900 python -m pytest test_gpu_renderer.py test_gpu_model_generation.py -m gpu -x -q 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -5) | tee "$OUT/tests.log"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/r" -- python "$REPO/tools/render64_trace.py" 64 10 > "$OUT/r.log" 2>&1)
echo "product: $(grep objects $OUT/r.log)" | tee "$OUT/render64_after3.txt"
grep -E "focused|tracking|histogram" "$OUT"/r/*/*kernel_stats.csv | cut -d, -f1-4 | tee -a "$OUT/render64_after3.txt"
cp "$OUT"/r/*/*kernel_stats.csv "$OUT/render64_kernel_stats.csv"; rm -rf "$OUT/r"
echo "no profiler: $(timeout 300 python tools/render64_trace.py 64 20 2>&1 | grep objects)" | tee -a "$OUT/render64_after3.txt"
echo "no vertex table: $(M3T_HIP_NO_VERTEX_TABLE=1 timeout 300 python tools/render64_trace.py 64 20 2>&1 | grep objects)" | tee -a "$OUT/render64_after3.txt"
echo "8 objects: $(timeout 300 python tools/render64_trace.py 8 20 2>&1 | grep objects)" | tee -a "$OUT/render64_after3.txt"
echo "8 objects, no vertex table: $(M3T_HIP_NO_VERTEX_TABLE=1 timeout 300 python tools/render64_trace.py 8 20 2>&1 | grep objects)" | tee -a "$OUT/render64_after3.txt"
echo "1 object: $(timeout 300 python tools/render64_trace.py 1 20 2>&1 | grep objects)" | tee -a "$OUT/render64_after3.txt"
