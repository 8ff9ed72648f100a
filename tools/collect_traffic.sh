#!/bin/bash
# HBM traffic of the bench kernels from the rocprofv3 PMC counters, one counter per pass
# (MI355X_MICROARCH.md, HBM section): writes gpurun_out/traffic/{fetch,write}/ and the summary
# gpurun_out/hbm_traffic.json, which is committed as profiles/rNN_hbm_traffic.json.
set -e
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/traffic
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
python "$REPO/tools/pmc_summary.py" "$OUT" "$REPO/gpurun_out/hbm_traffic.json"
