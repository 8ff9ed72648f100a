#!/bin/bash
# developer helper: round 5, call 6 -- the ROI guard / repeat (tests), the ROI legs of the bench line, 1-object phase timing
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${TAG:-r05f}
mkdir -p "$OUT"; cd "$REPO"
export M3T_INPUT_CACHE=${M3T_INPUT_CACHE:-/tmp/m3t_inputs_$(id -u)} M3T_INPUT_WORKERS=auto
(cd tests && timeout 900 python -m pytest test_gpu_roi.py test_gpu_generator.py -m gpu -x -q -s 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -40) > "$OUT/roi_tests.log" 2>&1
tail -25 "$OUT/roi_tests.log"
timeout 300 python tools/phase_timing.py tools/libm3t_hip_timing.so 1 2>&1 | grep -v amdgpu > "$OUT/phase_timing_rbot1.txt"
head -29 "$OUT/phase_timing_rbot1.txt"
timeout 900 python bench.py --config rbot64 --no-cpu-baseline --no-buckets --busy-seconds 1 > "$OUT/bench_rbot64_roi.json" 2> "$OUT/bench_rbot64_roi.err"
python - "$OUT/bench_rbot64_roi.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"])
    print(json.dumps(d.get("pcie_inclusive"), indent=1))
except Exception as e:
    print("bench:", e, open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
