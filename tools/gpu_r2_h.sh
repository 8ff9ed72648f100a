#!/bin/bash
# view list in closest_view: GPU tests, bench, phase timing; instruction breakdown of the YCB configuration
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r2h
mkdir -p "$OUT"
cd "$REPO"
(cd tests && timeout 1500 python -m pytest -m gpu -q -x --timeout=900 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -15) > "$OUT/gpu_tests.log" 2>&1
(timeout 600 python bench.py --no-pcie --cpu-seconds 3 --no-cpu-parallel > "$OUT/bench_default.json" 2> "$OUT/bench_default.err")
(timeout 600 python bench.py --config ycb21 --no-pcie --cpu-seconds 3 --no-cpu-parallel > "$OUT/bench_ycb21.json" 2> "$OUT/bench_ycb21.err")
(timeout 300 python tools/phase_timing.py tools/libm3t_hip_timing.so 64 2>&1 | tail -34) > "$OUT/phase_timing_rbot64.txt" 2>&1
(timeout 900 python bench.py --no-pcie --no-cpu-baseline --sweep 512,4096 > "$OUT/bench_sweep.json" 2> "$OUT/bench_sweep.err")
cd /tmp && export TMPDIR=/tmp
PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
for mode in ycb_region ycb_region_noocc ycb_depth; do
  M3T_HIP_NO_SPLIT=1 M3T_HIP_THREADS=256 timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/$mode" -- python $REPO/tools/instr_breakdown.py 21 $mode > "$OUT/$mode.log" 2>&1
  python $REPO/tools/instr_breakdown_summary.py "$OUT/$mode" 21 > "$OUT/$mode.txt" 2>&1
done
cd "$REPO"
tail -4 "$OUT/gpu_tests.log"; head -c 400 "$OUT/bench_default.json"; echo; head -c 300 "$OUT/bench_ycb21.json"; echo
cat "$OUT/phase_timing_rbot64.txt"; cat "$OUT"/ycb_*.txt
python - <<PY
import json
d = json.load(open("$OUT/bench_sweep.json"))
for p in d.get("batch_sweep", []): print(p)
print(json.load(open("$OUT/bench_default.json")).get("parity"))
print(json.load(open("$OUT/bench_ycb21.json")).get("parity"))
PY
