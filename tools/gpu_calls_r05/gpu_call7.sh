#!/bin/bash
# developer helper: round 5, call 7 -- adaptive margins again, per-kernel times of the 64-object renderer-fed step
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${TAG:-r05g}
mkdir -p "$OUT"; cd "$REPO"
export M3T_INPUT_CACHE=${M3T_INPUT_CACHE:-/tmp/m3t_inputs_$(id -u)} M3T_INPUT_WORKERS=auto
(cd tests && timeout 900 python -m pytest test_gpu_roi.py -m gpu -x -q 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -15) > "$OUT/roi_tests.log" 2>&1
tail -5 "$OUT/roi_tests.log"
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
for n in 64 1; do
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/render$n" -- python "$REPO/tools/render64_trace.py" $n 5 > "$OUT/render$n.log" 2>&1)
tail -1 "$OUT/render$n.log"
cp "$OUT"/render$n/*/*kernel_stats.csv "$OUT/render${n}_kernel_stats.csv" 2>/dev/null
head -8 "$OUT/render${n}_kernel_stats.csv"
rm -rf "$OUT/render$n"
done
