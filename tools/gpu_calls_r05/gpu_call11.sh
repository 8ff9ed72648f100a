#!/bin/bash
# developer helper: round 5, call 11 -- set-up kernel with the two-deep load pipeline, bulk pose reset; renderer + ROI tests
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${TAG:-r05k}
mkdir -p "$OUT"; cd "$REPO"
export M3T_INPUT_CACHE=${M3T_INPUT_CACHE:-/tmp/m3t_inputs_$(id -u)} M3T_INPUT_WORKERS=auto
(cd tests && timeout 900 python -m pytest test_gpu_renderer.py test_gpu_roi.py test_gpu_model_generation.py -m gpu -x -q 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -5) | tee "$OUT/tests.log"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/r" -- python "$REPO/tools/render64_trace.py" 64 10 > "$OUT/r.log" 2>&1)
echo "product: $(grep objects $OUT/r.log)" | tee "$OUT/render64_after2.txt"
grep -E "focused|tracking|histogram" "$OUT"/r/*/*kernel_stats.csv | cut -d, -f1-4 | tee -a "$OUT/render64_after2.txt"
cp "$OUT"/r/*/*kernel_stats.csv "$OUT/render64_kernel_stats.csv"; rm -rf "$OUT/r"
echo "no profiler: $(timeout 300 python tools/render64_trace.py 64 20 2>&1 | grep objects)" | tee -a "$OUT/render64_after2.txt"
echo "1 object: $(timeout 300 python tools/render64_trace.py 1 20 2>&1 | grep objects)" | tee -a "$OUT/render64_after2.txt"
M3T_BENCH_RESERVE_CUS=32 timeout 900 python bench.py --config rbot64 --no-cpu-baseline --no-buckets --busy-seconds 1 > "$OUT/bench_rbot64_roi.json" 2> "$OUT/bench_rbot64_roi.err"
python - "$OUT/bench_rbot64_roi.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"])
    print(json.dumps(d["pcie_inclusive"]["roi_rectangles"]["reserved_cus"], indent=1))
except Exception as e:
    print("bench:", e, open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
