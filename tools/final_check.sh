#!/bin/bash
# what the driver runs at the end of a round, in one gpurun call: pytest -m gpu, smoke(), the default bench line
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${TAG:-final}
mkdir -p "$OUT"; cd "$REPO"
export M3T_INPUT_CACHE=${M3T_INPUT_CACHE:-${XDG_CACHE_HOME:-$HOME/.cache}/m3t_inputs} M3T_INPUT_WORKERS=${M3T_INPUT_WORKERS:-auto}
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -8) | tee "$OUT/gpu_tests_final.log"
(timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) | tee "$OUT/smoke.log"
t0=$(date +%s.%N)
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "python bench.py: $(echo "$(date +%s.%N) - $t0" | bc 2>/dev/null) s wall" | tee -a "$OUT/smoke.log"
tail -1 "$OUT/bench_default.json" | cut -c1-400
