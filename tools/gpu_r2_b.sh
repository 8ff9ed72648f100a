#!/bin/bash
# round 2, GPU call B: second kernel iteration + new entry points + config legs
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r2b
mkdir -p "$OUT"
cd "$REPO"
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; python -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())"; lscpu | head -20; uptime) > "$OUT/host.log" 2>&1
(cd tests && timeout 1500 python -m pytest -m gpu -q -x --timeout=600 test_gpu_parity.py test_gpu_split.py test_gpu_edge_cases.py test_gpu_multibody.py 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -40) > "$OUT/tests_core.log" 2>&1
(timeout 600 python tools/sweep_shapes.py rbot ycb 2>&1 | grep -E "^\{|Error|error|Traceback" ) > "$OUT/sweep.log" 2>&1
for v in "split:64:" "ycb:21:ycb"; do
  IFS=: read name n ycb <<< "$v"
  (timeout 300 python tools/phase_timing.py tools/libm3t_hip_timing.so $n $ycb 2>&1 | tail -34) > "$OUT/phase_$name.log" 2>&1
done
(timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err")
(timeout 600 python bench.py --config ycb21 > "$OUT/bench_ycb21.json" 2> "$OUT/bench_ycb21.err")
(timeout 900 python bench.py --config synth512 --steps 10 --warmup 3 > "$OUT/bench_synth512.json" 2> "$OUT/bench_synth512.err")
tail -5 "$OUT/tests_core.log"; cat "$OUT/sweep.log"; cat "$OUT/host.log" | head -8; for f in default ycb21 synth512; do head -c 700 "$OUT/bench_$f.json"; echo; tail -3 "$OUT/bench_$f.err"; done
