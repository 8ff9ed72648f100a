#!/bin/bash
# Round profile set: GPU test log, the full bench line (sweep + YCB + extras), rocprofv3 kernel stats of the
# default bench command.  Outputs under gpurun_out/; copied to profiles/rNN_* by hand.
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
(cd "$REPO/tests" && timeout 1200 python -m pytest -m gpu -q 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -8) > "$OUT/gpu_tests.log"
(cd "$REPO" && python bench.py --sweep 1,8,256,512,1024,4096,32768 --ycb 21 --extras > "$OUT/bench_full.json" 2> "$OUT/bench_full.err")
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python "$REPO/bench.py" --no-cpu-baseline > "$OUT/prof.log" 2>&1
cp "$OUT"/prof/*/*kernel_stats.csv "$OUT/bench_kernel_stats.csv"
cp "$OUT"/prof/*/*domain_stats.csv "$OUT/bench_domain_stats.csv" 2>/dev/null
tail -1 "$OUT/gpu_tests.log"; head -c 600 "$OUT/bench_full.json"; echo; head -4 "$OUT/bench_kernel_stats.csv"
