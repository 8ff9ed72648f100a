#!/bin/bash
# Round profile set (outputs under gpurun_out/r02/; the ones to be judged are copied to profiles/r02_* by hand):
# GPU test log, the bench lines of all four configurations, the batch sweep, rocprofv3 kernel stats of the default
# bench command, the per-phase cycle breakdown of the developer build, the PMC passes (tools/collect_pmc.sh).
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r02
mkdir -p "$OUT"
cd "$REPO"
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; rocm-smi --showproductname 2>/dev/null | head -12) > "$OUT/host.log" 2>&1
(cd tests && timeout 1500 python -m pytest -m gpu -q --timeout=900 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -8) > "$OUT/gpu_tests.log" 2>&1
(timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err")
(timeout 600 python bench.py --config ycb21 > "$OUT/bench_ycb21.json" 2> "$OUT/bench_ycb21.err")
(timeout 900 python bench.py --config synth512 --steps 10 --warmup 3 > "$OUT/bench_synth512.json" 2> "$OUT/bench_synth512.err")
(timeout 600 python bench.py --config chain8 > "$OUT/bench_chain8.json" 2> "$OUT/bench_chain8.err")
(timeout 1200 python bench.py --no-pcie --no-cpu-baseline --sweep 1,8,32,256,512,1024,4096 --extras > "$OUT/bench_sweep.json" 2> "$OUT/bench_sweep.err")
for v in "rbot64:64:" "ycb21:21:ycb"; do
  IFS=: read name n ycb <<< "$v"
  (timeout 300 python tools/phase_timing.py tools/libm3t_hip_timing.so $n $ycb 2>&1 | tail -34) > "$OUT/phase_timing_$name.txt" 2>&1
done
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python "$REPO/bench.py" --no-cpu-baseline --no-pcie > "$OUT/prof.log" 2>&1)
cp "$OUT"/prof/*/*kernel_stats.csv "$OUT/bench_kernel_stats.csv" 2>/dev/null
bash tools/collect_pmc.sh rbot64 > "$OUT/pmc_rbot64.log" 2>&1
bash tools/collect_pmc.sh ycb21 > "$OUT/pmc_ycb21.log" 2>&1
tail -2 "$OUT/gpu_tests.log"; for f in default ycb21 synth512 chain8; do head -c 300 "$OUT/bench_$f.json"; echo; done; head -5 "$OUT/bench_kernel_stats.csv"; tail -25 "$OUT/pmc_rbot64.log"
