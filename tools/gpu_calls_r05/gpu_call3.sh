#!/bin/bash
# developer helper: round 5, call 3
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${TAG:-r05c}
mkdir -p "$OUT"; cd "$REPO"
export M3T_INPUT_CACHE=${M3T_INPUT_CACHE:-/tmp/m3t_inputs_$(id -u)} M3T_INPUT_WORKERS=auto
OLD=tools/variants/r04/libm3t_hip.so; NEW=3dobjecttracking_amd/csrc/libm3t_hip.so
(cd tests && timeout 600 python -m pytest test_gpu_multibody.py -m gpu -x -q 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -12) > "$OUT/multibody.log" 2>&1
tail -3 "$OUT/multibody.log"
timeout 300 python tools/chain_bench.py --oracle $OLD $NEW 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" > "$OUT/chain_ab.txt"
cat "$OUT/chain_ab.txt"
timeout 200 python tools/tree_timing.py tools/libm3t_hip_timing.so 2>&1 | grep -v amdgpu > "$OUT/tree_timing_new.txt"
cat "$OUT/tree_timing_new.txt"
# tree sums experiment
timeout 600 python tools/quick_bench.py --objects 64 $NEW tools/variants/treesums/libm3t_hip.so $NEW tools/variants/treesums/libm3t_hip.so > "$OUT/qb_treesums.txt" 2>&1
cat "$OUT/qb_treesums.txt"
timeout 600 python tools/tree_sums_deviation.py tools/variants/treesums/libm3t_hip.so 64 50 2>&1 | grep -v amdgpu > "$OUT/tree_sums_deviation.txt"
cat "$OUT/tree_sums_deviation.txt"
# rank-share projections (item 2): code path check + first numbers
for c in ycb21 chain8; do
  timeout 600 python bench.py --config $c --rank-share 1,2,4,8 --no-cpu-baseline --no-pcie --no-buckets --busy-seconds 1 > "$OUT/rank_share_$c.json" 2> "$OUT/rank_share_$c.err"
  python - "$OUT/rank_share_$c.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["metric"], d["ms_per_step"], d["config"].get("rccl_ranks"))
    for p in (d.get("projected_scaling") or {}).get("points", []):
        print("  ", p)
except Exception as e:
    print("rank-share:", e, open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done
