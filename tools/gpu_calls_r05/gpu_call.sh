#!/bin/bash
# developer helper: one gpurun call of round 5 (which = the sections to run)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${TAG:-r05a}
OUT=$REPO/gpurun_out/$TAG
WHAT=${@:-multibody chain timing qb cal tests}
mkdir -p "$OUT"
cd "$REPO"
export M3T_INPUT_CACHE=${M3T_INPUT_CACHE:-/tmp/m3t_inputs_$(id -u)}
export M3T_INPUT_WORKERS=${M3T_INPUT_WORKERS:-auto}
OLD=${OLD:-tools/variants/r04/libm3t_hip.so}
NEW=${NEW:-3dobjecttracking_amd/csrc/libm3t_hip.so}
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has multibody; then
  (cd tests && timeout 600 python -m pytest test_gpu_multibody.py -m gpu -x -q 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -25) > "$OUT/multibody.log" 2>&1
  tail -3 "$OUT/multibody.log"
fi
if has chain; then
  timeout 400 python tools/chain_bench.py --oracle --distributed $OLD $NEW $EXTRA_LIBS 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" > "$OUT/chain_ab.txt"
  cat "$OUT/chain_ab.txt"
fi
if has timing; then
  [ -f tools/variants/r04/libm3t_hip_timing.so ] && timeout 200 python tools/tree_timing.py tools/variants/r04/libm3t_hip_timing.so 2>&1 | grep -v amdgpu > "$OUT/tree_timing_r04.txt"
  timeout 200 python tools/tree_timing.py tools/libm3t_hip_timing.so 2>&1 | grep -v amdgpu > "$OUT/tree_timing_new.txt"
  cat "$OUT/tree_timing_new.txt"
fi
if has qb; then
  timeout 500 python tools/quick_bench.py --objects 64 $OLD $NEW $OLD $NEW $EXTRA_LIBS > "$OUT/qb_rbot64.txt" 2>&1
  cat "$OUT/qb_rbot64.txt"
  timeout 500 python tools/quick_bench.py --ycb --objects 21 $OLD $NEW $OLD $NEW $EXTRA_LIBS > "$OUT/qb_ycb21.txt" 2>&1
  cat "$OUT/qb_ycb21.txt"
fi
if has cal; then
  mkdir -p "$OUT/cal"
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/cal/fetch" -- "$REPO/tools/bin/ubench_counters" > "$OUT/cal/known.txt" 2> "$OUT/cal/fetch.log"
   timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/cal/write" -- "$REPO/tools/bin/ubench_counters" > "$OUT/cal/known_w.txt" 2> "$OUT/cal/write.log")
  python tools/counter_calibration.py "$OUT/cal" "$OUT/cal/known.txt" "$OUT/counter_calibration.txt"
  find "$OUT/cal" -name "*.csv" -size +2M -delete
fi
if has tests; then
  (cd tests && timeout 1500 python -m pytest -m gpu -q --timeout=900 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -12) > "$OUT/gpu_tests.log" 2>&1
  tail -3 "$OUT/gpu_tests.log"
fi
