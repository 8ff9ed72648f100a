#!/bin/bash
# Round profile set, one script, one commit (outputs under gpurun_out/$ROUND/; tools/publish_profiles.py copies the
# ones to be judged to profiles/${ROUND}_*):
#   * GPU test log
#   * the bench line of every configuration (bench.py --config ...), the large-batch line (4096 objects) and a sweep
#   * rocprofv3 --kernel-trace --stats of every configuration's bench command
#   * PMC passes (SQ x 2, TA, TCC, FETCH_SIZE, WRITE_SIZE: MI355X_MICROARCH.md -- 8 SQ / 4 TCC slots per pass, the two
#     size counters in passes of their own, never together with the hip / hsa trace domains) of the same commands
#   * per-phase cycles of the developer build (tools/phase_timing.py)
# Order: the PMC passes first -- their HBM traffic blocks go into profiles/ on the box, so the bench lines taken after
# them quote this run's traffic (roofline.traffic_source).
#   * (round 5) counter calibration (FETCH_SIZE / WRITE_SIZE against known byte counts, before the PMC passes that use it),
#     rank-share projections (one rank's share at 1 / 2 / 4 / 8 ranks, alone on this GPU), per-kernel statistics of the
#     64-object renderer-fed step
# usage: tools/collect_profiles.sh [what ...]   what = tests cal pmc bench extras stats phases rankshare render sweep (default: all)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUND=${ROUND:-r06}
OUT=$REPO/gpurun_out/$ROUND
WHAT=${@:-tests cal pmc bench extras stats phases rankshare render sweep}
CONFIGS=${CONFIGS:-rbot64 rbot4096 ycb21 synth512 chain8}
mkdir -p "$OUT"
cd "$REPO"
# every bench.py invocation below would regenerate its configuration's inputs (rbot64: 39 s of numpy on the box's CPU,
# synth512: 66 s -- about 14 of the 23.5 minutes of round 4's collection): generate once per (configuration, frame count)
export M3T_INPUT_CACHE=${M3T_INPUT_CACHE:-${XDG_CACHE_HOME:-$HOME/.cache}/m3t_inputs}  # (a private directory: the cache holds pickles, bench_inputs.py refuses one that others can write to)
export M3T_INPUT_WORKERS=${M3T_INPUT_WORKERS:-auto}  # ... and the first generation on all cores (same bits: tests/test_input_cache.py)
has() { [[ " $WHAT " == *" $1 "* ]]; }
args_of() {  # bench.py arguments of a profile configuration
  case $1 in
    rbot4096) echo "--config rbot64 --objects 4096" ;;
    *) echo "--config $1" ;;
  esac
}
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; rocm-smi --showproductname 2>/dev/null | head -12; git -C "$REPO" rev-parse HEAD 2>/dev/null) > "$OUT/host.log" 2>&1

if has tests; then
  (cd tests && timeout 1500 python -m pytest -m gpu -q --timeout=900 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -8) > "$OUT/gpu_tests.log" 2>&1
  tail -2 "$OUT/gpu_tests.log"
fi
PROF="--steps 10 --warmup 2 --no-cpu-baseline --no-pcie --no-buckets --repeats 1"
if has cal; then  # what FETCH_SIZE / WRITE_SIZE report for known byte counts in the tracking kernels' access patterns
  mkdir -p "$OUT/cal" tools/bin
  [ -x tools/bin/ubench_counters ] || hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_counters tools/ubench_counters.hip > "$OUT/cal/build.log" 2>&1
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/cal/fetch" -- "$REPO/tools/bin/ubench_counters" > "$OUT/cal/known.txt" 2> "$OUT/cal/fetch.log"
   timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/cal/write" -- "$REPO/tools/bin/ubench_counters" > "$OUT/cal/known_w.txt" 2> "$OUT/cal/write.log")
  python tools/counter_calibration.py "$OUT/cal" "$OUT/cal/known.txt" "$OUT/counter_calibration.txt" | tail -10
  [ -s "$OUT/counter_calibration.txt" ] && grep -q "^factor" "$OUT/counter_calibration.txt" && cp "$OUT/counter_calibration.txt" "$REPO/profiles/${ROUND}_counter_calibration.txt"
  rm -rf "$OUT/cal"
fi
if has pmc; then
  declare -A PASS
  PASS[sq1]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
  PASS[sq2]="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"
  PASS[ta]="TA_TA_BUSY_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
  PASS[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
  PASS[fetch]="FETCH_SIZE"
  PASS[write]="WRITE_SIZE"
  for c in ${PMC_CONFIGS:-rbot64 rbot4096 ycb21 synth512 chain8}; do
    mkdir -p "$OUT/pmc_$c"
    # all passes for the configurations in PMC_FULL (default: the headline), FETCH_SIZE / WRITE_SIZE only for the
    # others (every bench line gets its roofline.traffic; a pass is one more run of the bench command)
    passes="fetch write"
    [[ " ${PMC_FULL:-rbot64 rbot4096 chain8} " == *" $c "* ]] && passes="sq1 sq2 ta tcc fetch write"
    for p in ${PMC_PASSES:-$passes}; do
      (cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --pmc ${PASS[$p]} --output-format csv -d "$OUT/pmc_$c/$p" -- python "$REPO/bench.py" $(args_of $c) $PROF > "$OUT/pmc_$c/$p.log" 2>&1)
    done
    python tools/pmc_summary3.py "$OUT/pmc_$c" "$OUT/pmc_$c.json" "$c" "bench.py $(args_of $c) $PROF" | tail -24
    rm -rf "$OUT/pmc_$c"
  done
fi

# the bench lines below quote this run's traffic blocks (bench.py looks them up under profiles/)
for f in "$OUT"/hbm_traffic_*.json; do [ -f "$f" ] && cp "$f" "$REPO/profiles/${ROUND}_$(basename "$f")"; done
if has bench; then
  for c in $CONFIGS; do
    extra=""
    [ "$c" = rbot4096 ] && extra="--no-pcie --cpu-seconds 4 --no-cpu-parallel"
    (timeout 900 python bench.py $(args_of $c) $extra > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err")
    head -c 400 "$OUT/bench_$c.json"; echo
  done
  # (round 6) the 4096-object line again with the kernel that gathers its pairs from L2: the LDS table's before / after
  (M3T_HIP_COMPACT_TABLE=0 timeout 900 python bench.py --config rbot64 --objects 4096 --no-pcie --no-cpu-baseline --no-buckets > "$OUT/bench_rbot4096_plain_kernel.json" 2> "$OUT/bench_rbot4096_plain_kernel.err")
  head -c 300 "$OUT/bench_rbot4096_plain_kernel.json"; echo
fi
if has extras; then
  (timeout 900 python bench.py --extras --no-pcie --no-cpu-baseline --no-buckets --busy-seconds 1 > "$OUT/bench_extras.json" 2> "$OUT/bench_extras.err")
  python - "$OUT/bench_extras.json" <<'PY'
import json, sys
print(json.dumps(json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]).get("extras"))[:600])
PY
fi
if has stats; then
  for c in $CONFIGS; do
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$c" -- python "$REPO/bench.py" $(args_of $c) $PROF > "$OUT/stats_$c.log" 2>&1)
    cp "$OUT"/stats_$c/*/*kernel_stats.csv "$OUT/kernel_stats_$c.csv" 2>/dev/null
    rm -rf "$OUT/stats_$c"  # (the raw traces of a 4096-object run are tens of MB; gpurun merges at most 64 MB back)
    head -4 "$OUT/kernel_stats_$c.csv"
  done
fi
if has phases; then
  for v in "rbot64:64:" "rbot1:1:" "ycb21:21:ycb"; do
    IFS=: read name n ycb <<< "$v"
    (timeout 300 python tools/phase_timing.py tools/libm3t_hip_timing.so $n $ycb 2>&1 | grep -v amdgpu) > "$OUT/phase_timing_$name.txt" 2>&1
  done
  (timeout 300 python tools/tree_timing.py tools/libm3t_hip_timing.so 2>&1 | grep -v amdgpu) > "$OUT/phase_timing_chain8.txt" 2>&1
  head -12 "$OUT/phase_timing_rbot64.txt"
fi
if has rankshare; then  # what one GPU can say about N: rank 0's share at N = 1, 2, 4, 8 ranks, alone on this GPU (a projection)
  for c in ${RANKSHARE_CONFIGS:-rbot64 synth512 ycb21 chain8}; do
    (timeout 900 python bench.py --config $c --rank-share 1,2,4,8 --no-cpu-baseline --no-pcie --no-buckets --busy-seconds 1 > "$OUT/rank_share_$c.json" 2> "$OUT/rank_share_$c.err")
    python - "$OUT/rank_share_$c.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["metric"], d["ms_per_step"], "rccl_ranks", d["config"].get("rccl_ranks"))
    for p in (d.get("projected_scaling") or {}).get("points", []):
        print("  ", {k: v for k, v in p.items() if k in ("n_gpus", "objects_on_rank_0", "bodies_with_modalities_on_rank_0", "rank_0_ms_per_step", "projected_pose_updates_per_s", "projected_pose_updates_per_s_before_transport")})
except Exception as e:
    print("rank-share:", e)
PY
  done
fi
if has render; then  # the renderer-fed step of the reference's test scene, 64 times in one context and once: per-kernel times
  for n in 64 1; do
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/render$n" -- python "$REPO/tools/render64_trace.py" $n 10 > "$OUT/render$n.log" 2>&1)
    cp "$OUT"/render$n/*/*kernel_stats.csv "$OUT/render${n}_kernel_stats.csv" 2>/dev/null
    rm -rf "$OUT/render$n"
    (grep objects "$OUT/render$n.log"; echo "without the profiler: $(timeout 300 python tools/render64_trace.py $n 20 2>&1 | grep objects)") | tee "$OUT/render$n.txt"
    rm -f "$OUT/render$n.log"
  done
fi
if has sweep; then
  (timeout 1200 python bench.py --no-pcie --no-cpu-baseline --no-buckets --repeats 3 --sweep 1,8,32,256,512,1024,4096 > "$OUT/bench_sweep.json" 2> "$OUT/bench_sweep.err")
  # SURVEY 8(d) also names 32768 objects (shared cameras above 4096: batch_point): its own run and time limit
  (timeout 700 python bench.py --no-pcie --no-cpu-baseline --no-buckets --repeats 1 --busy-seconds 1 --sweep 32768 > "$OUT/bench_sweep_32768.json" 2> "$OUT/bench_sweep_32768.err")
  python - "$OUT/bench_sweep.json" <<'PY'
import json, sys
import os
for path in (sys.argv[1], sys.argv[1].replace(".json", "_32768.json")):
    if os.path.exists(path) and os.path.getsize(path):
        for s in json.load(open(path)).get("batch_sweep", []):
            print(s["objects"], s["pose_updates_per_s"], s["frac_of_hbm_roofline"], s.get("kernel"))
PY
fi
