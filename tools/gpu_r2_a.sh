#!/bin/bash
# round 2, GPU call A: correctness of the restructured kernels + first timings
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r2a
mkdir -p "$OUT"
cd "$REPO"
(timeout 300 python __graft_entry__.py smoke 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -12) > "$OUT/smoke.log" 2>&1
(cd tests && timeout 1500 python -m pytest -m gpu -q -x --timeout=600 test_gpu_parity.py test_gpu_split.py test_gpu_edge_cases.py 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -40) > "$OUT/tests_core.log" 2>&1
(timeout 600 python tools/sweep_shapes.py rbot ycb 2>&1 | grep -E "^\{|Error|error|Traceback" ) > "$OUT/sweep.log" 2>&1
for v in "split:64:" "nosplit:64:" "ycb:21:ycb"; do
  IFS=: read name n ycb <<< "$v"
  if [ "$name" = nosplit ]; then export M3T_HIP_NO_SPLIT=1; else unset M3T_HIP_NO_SPLIT; fi
  (timeout 300 python tools/phase_timing.py tools/libm3t_hip_timing.so $n $ycb 2>&1 | tail -30) > "$OUT/phase_$name.log" 2>&1
done
unset M3T_HIP_NO_SPLIT
(timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err")
(cd tests && timeout 1200 python -m pytest -m gpu -q --timeout=600 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -40) > "$OUT/tests_all.log" 2>&1
tail -3 "$OUT/smoke.log"; tail -5 "$OUT/tests_core.log"; cat "$OUT/sweep.log"; tail -3 "$OUT/tests_all.log"; head -c 1500 "$OUT/bench_default.json"
