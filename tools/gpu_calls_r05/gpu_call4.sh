#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${TAG:-r05d}
mkdir -p "$OUT"; cd "$REPO"
export M3T_INPUT_CACHE=${M3T_INPUT_CACHE:-/tmp/m3t_inputs_$(id -u)} M3T_INPUT_WORKERS=auto
OLD=tools/variants/r04/libm3t_hip.so; NEW=3dobjecttracking_amd/csrc/libm3t_hip.so
(cd tests && timeout 600 python -m pytest test_gpu_multibody.py -m gpu -q 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -25) > "$OUT/multibody.log" 2>&1
tail -12 "$OUT/multibody.log"
timeout 300 python tools/chain_bench.py --oracle --distributed $OLD $NEW 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" > "$OUT/chain_ab.txt"
cat "$OUT/chain_ab.txt"
timeout 600 python bench.py --config chain8 --rank-share 1,2,4,8 --no-cpu-baseline --busy-seconds 1 > "$OUT/rank_share_chain8.json" 2> "$OUT/rank_share_chain8.err"
python - "$OUT/rank_share_chain8.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["metric"], d["ms_per_step"], d["config"].get("rccl_ranks"), d.get("distributed_path_world1"))
    for p in (d.get("projected_scaling") or {}).get("points", []):
        print("  ", p)
except Exception as e:
    print("rank-share:", e, open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
(cd tests && timeout 1500 python -m pytest -m gpu -q --timeout=900 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -12) > "$OUT/gpu_tests.log" 2>&1
tail -4 "$OUT/gpu_tests.log"
