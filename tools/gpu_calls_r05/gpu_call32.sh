#!/bin/bash
# round 5, call 32 -- ROI rectangles with the ellipsoid bound: tests, the ROI legs at margins 24 / 20 / 16
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/r05u
mkdir -p "$OUT"; cd "$REPO"
export M3T_INPUT_CACHE=${M3T_INPUT_CACHE:-/tmp/m3t_inputs_$(id -u)} M3T_INPUT_WORKERS=auto
(cd tests && timeout 900 python -m pytest test_gpu_roi.py test_gpu_generator.py -m gpu -q 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl|Could not|Set up)" | tail -6) | tee "$OUT/roi_tests.log"
for m in 24 20 16; do
M3T_BENCH_ROI_MARGIN=$m M3T_BENCH_RESERVE_CUS=32,64 timeout 900 python bench.py --config rbot64 --no-cpu-baseline --no-buckets --busy-seconds 1 > "$OUT/bench_roi_m$m.json" 2> "$OUT/bench_roi_m$m.err"
python - "$OUT/bench_roi_m$m.json" $m <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["pcie_inclusive"]["roi_rectangles"]
    print("margin", sys.argv[2], "no reservation:", r["pose_updates_per_s"], r["ms_per_step"], "repeated", r["bodies_outside_their_rectangle"], r["bit_identical_to_whole_frames"])
    for e in r["reserved_cus"]:
        print("   ", e["cus_for_the_pull"], "CUs:", e["pose_updates_per_s"], e["ms_per_step"], "repeated", e["bodies_outside_their_rectangle"], e["bit_identical_to_whole_frames"], "| adaptive", e["adaptive_margins"]["pose_updates_per_s"], e["adaptive_margins"]["bodies_repeated_on_whole_frames"])
except Exception as e:
    print("bench:", e, open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done | tee "$OUT/roi_margins.txt"
