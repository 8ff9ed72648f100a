#!/bin/bash
# developer helper: round 5, call 5 -- inputs of profiles/r05_headline_floor.txt: per-phase cycles of
# the 1-object and the 64-object step (timing build) and per-setting instruction counts (PMC) of the same two steps
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${TAG:-r05e}
mkdir -p "$OUT"; cd "$REPO"
export M3T_INPUT_CACHE=${M3T_INPUT_CACHE:-/tmp/m3t_inputs_$(id -u)} M3T_INPUT_WORKERS=auto
for n in 1 64; do
  timeout 300 python tools/phase_timing.py tools/libm3t_hip_timing.so $n 2>&1 | grep -v amdgpu > "$OUT/phase_timing_rbot$n.txt"
  head -30 "$OUT/phase_timing_rbot$n.txt"
done
for n in 1 64; do
  for p in a b; do
    [ $p = a ] && CNT="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
    [ $p = b ] && CNT="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE"
    (cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$OUT/instr_${n}_$p" -- python "$REPO/tools/instr_breakdown.py" $n rbot > "$OUT/instr_${n}_$p.log" 2>&1)
    python tools/instr_breakdown_summary.py "$OUT/instr_${n}_$p" $n > "$OUT/instr_breakdown_rbot${n}_$p.txt" 2>&1
    cat "$OUT/instr_breakdown_rbot${n}_$p.txt"
    rm -rf "$OUT/instr_${n}_$p"
  done
done
