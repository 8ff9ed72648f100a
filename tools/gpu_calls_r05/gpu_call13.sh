#!/bin/bash
# developer helper: round 5, call 13 -- where the set-up kernel's time goes at 128 pairs (probe builds)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${TAG:-r05m}
mkdir -p "$OUT"; cd "$REPO"
for v in sprobe1 sprobe2; do
  lib=$REPO/tools/variants/$v/libm3t_hip.so
  (cd /tmp && export TMPDIR=/tmp && M3T_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/r_$v" -- python "$REPO/tools/render64_trace.py" 64 5 > "$OUT/r_$v.log" 2>&1)
  echo "$v: $(grep objects $OUT/r_$v.log)"
  grep -E "focused|tracking" "$OUT"/r_$v/*/*kernel_stats.csv | cut -d, -f1-4
  rm -rf "$OUT/r_$v"
done | tee "$OUT/setup_probes.txt"
