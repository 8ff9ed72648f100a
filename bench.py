#!/usr/bin/env python
"""bench.py — pose-updates/s of the M3T per-frame pose-optimisation hot path on MI355X.

One "step" = Tracker::ExecuteTrackingStep (the whole loop nest of correspondence iterations x Newton updates +
histogram update) for every object of the batch on the next synthetic frame.  Configurations (BASELINE.json):

  --config rbot64   (default, configs[1]) 64 batched RBOT-geometry objects PER GPU, RegionModality, 200 lines x 7 x 2,
                    640x512 BGR8, 32 bins.  Objects are independent: ranks share nothing on the data path (weak scaling).
  --config ycb21    (configs[2]) 21 objects, Region + Depth fused modalities, YCB parameters, 640x480.
  --config synth512 (configs[3]) 512 synthetic objects (random sparse viewpoint models), Region + Depth, sharded
                    over the GPUs (strong scaling: 512 / N objects per GPU).
  --config chain8   (configs[4]) kinematic chain of 8 bodies / 13 dof tracked by 8 RegionModalities, the structure's
                    joint Hessian summed over the GPUs with ONE all-reduce per Newton step; plus the sweep over
                    1..50 bodies of examples/optimization_time.cpp.

Frames, models and histograms are resident in HBM before the timed region starts (the reference's evaluators also
exclude image loading, rbot_evaluator.cpp:354-414).  Prints ONE JSON line on rank 0 (roofline,
cpu_baseline and parity blocks).
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
ORACLE_DIR = os.path.join(ROOT, "oracle")  # the checker / CPU baseline: never in the timed GPU path

# algorithmic bytes per pose-update (Region only, RBOT parameters)
B_PIXELS = 49400 * 3
B_HIST_READ = 2 * 32 ** 3 * 4
B_VIEW_SCAN = 7 * 2562 * 12
B_VIEW_POINTS = 7 * 200 * 152
B_HIST_RMW = 2 * (131072 + 131072)
B_HIST_PIXELS = 200 * 40 * 3
B_POSE = 64
B_ALG = B_PIXELS + B_HIST_READ + B_VIEW_SCAN + B_VIEW_POINTS + B_HIST_RMW + B_HIST_PIXELS + B_POSE  # 1 386 704
B_ALG_TRACK_KERNEL = B_PIXELS + B_HIST_READ + B_VIEW_SCAN + B_VIEW_POINTS + B_POSE  # fused tracking kernel only
# Region + Depth with measured occlusions, YCB parameters (16 bins, 4 correspondence iterations)
B_ALG_YCB = 1094456
B_ALG_YCB_HIST = 65536 + 24000 + 122976 // 4 + 121600 // 4  # the histogram update's share when it is a launch of its own
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)

CONFIGS = {
    "rbot64": dict(metric="pose-updates/sec (64 objects, 200 lines, 7 it)", scaling="weak", with_depth=False,
                   objects=64, models=18, alg=B_ALG, alg_track=B_ALG_TRACK_KERNEL, newton=14,
                   workload="BASELINE configs[1]: %(n)d batched RBOT-geometry objects per GPU, RegionModality only, "
                            "200 lines x 7 corr-iterations x 2 updates, 640x512 BGR8, 32-bin histograms, "
                            "%(views)d views x 200 points models (%(models)d distinct)"),
    "ycb21": dict(metric="pose-updates/sec (YCB-Video scene, 21 objects, Region+Depth)", scaling="strong",
                  with_depth=True, objects=21, models=6, alg=B_ALG_YCB, alg_track=B_ALG_YCB - B_ALG_YCB_HIST, newton=8,
                  workload="BASELINE configs[2]: %(n)d objects in one scene, Region + Depth fused modalities (ICG), YCB "
                           "parameters (scales 7/4/2, 16 bins, measured occlusions, 4 corr-iterations x 2 updates), "
                           "640x480 BGR8 + u16 depth, %(views)d views x 200 points models (%(models)d distinct)"),
    "synth512": dict(metric="pose-updates/sec (512 synthetic objects, Region+Depth)", scaling="strong",
                     with_depth=True, objects=512, models=16, alg=B_ALG_YCB, alg_track=B_ALG_YCB - B_ALG_YCB_HIST,
                     newton=8,
                     workload="BASELINE configs[3]: %(n)d synthetic objects on this GPU (512 over all GPUs; random "
                              "star-shaped bodies, %(models)d distinct sparse viewpoint models of %(views)d views x 200 "
                              "points), Region + Depth, YCB parameters, every object its own 640x480 frame ring "
                              "(64 distinct rendered streams)"),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--config", default="rbot64", choices=sorted(CONFIGS) + ["chain8"])
    p.add_argument("--objects", type=int, default=0, help="objects (per GPU for rbot64, in total otherwise); 0 = the config's")
    p.add_argument("--models", type=int, default=0, help="distinct sparse viewpoint models; 0 = the config's")
    p.add_argument("--n-divides", type=int, default=4, help="geodesic subdivisions (4 -> 2562 views)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline sample budget (1 thread)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-cpu-parallel", action="store_true", help="skip the all-cores (OpenMP) CPU leg")
    p.add_argument("--repeats", type=int, default=0,
                   help="how often the timed region of K steps is repeated; 0 = until --busy-seconds of timed regions")
    p.add_argument("--busy-seconds", type=float, default=6.0,
                   help="with --repeats 0: keep repeating the K-step region until this much GPU time has been timed "
                        "(at least 5 regions): a K = 20 region lasts ~3 ms, too short for device-level telemetry")
    p.add_argument("--no-pcie", action="store_true", help="skip the host-buffer (PCIe-inclusive) legs")
    p.add_argument("--no-buckets", action="store_true",
                   help="skip the one-launch-per-sub-step leg (profiling runs: only the timed kernels are launched)")
    p.add_argument("--sweep", type=str, default="", help="comma separated object counts for a batch sweep (extra)")
    p.add_argument("--rank-share", type=str, default="",
                   help="comma separated GPU counts N (e.g. 1,2,4,8): measure rank 0's share of the configuration -- the "
                        "objects rank_plan() gives it at N ranks, or the chain's distributed path with the modalities "
                        "of bodies i mod N == 0 -- ALONE on this one GPU and emit `projected_scaling` (a projection: "
                        "no transport, no barrier skew; never `value`)")
    p.add_argument("--extras", action="store_true",
                   help="extra legs (never the headline): model generation without OpenGL and a tracking step with "
                        "all renderer-fed branches, on the reference's own test fixture")
    return p.parse_args()


def open_oracle(native=False):
    """oracle/libm3t_oracle.so (the bit-exact checker: x86-64-v3, no FMA contraction) or, native=True,
    oracle/libm3t_oracle_native.so: the same source built the way the reference builds (-O3 -march=native, default
    contraction, M3T/CMakeLists.txt:73-80) on THIS host -- faster, not bit-identical, used for timing only."""
    import subprocess
    pkg = importlib.import_module("3dobjecttracking_amd")
    name = "libm3t_oracle_native.so" if native else "libm3t_oracle.so"
    path = os.path.join(ORACLE_DIR, name)
    if native:  # always rebuilt: -march=native belongs to the host that runs it
        subprocess.check_call(["make", "-s", "-B", "-C", ORACLE_DIR, name])
    elif not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, name])
    return pkg.CApi(path, "m3t_oracle_")


def share_one_gpu():
    """Developer switch M3T_BENCH_SHARE_ONE_GPU=1: a DRY RUN of the N-rank code path on a box with one GPU -- every
    rank uses device 0, the process group runs on gloo, one workgroup per object (co-resident split launches of two
    processes on one GPU could wait for each other).  The line says so (`dry_run`) and its value measures nothing."""
    return os.environ.get("M3T_BENCH_SHARE_ONE_GPU") == "1"


def visible_gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n_gpus, argv, n_visible=None):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks, one per GPU, under
    torch.distributed.run on 127.0.0.1 (what the driver does itself for its scaling runs).  Fails loudly when fewer
    than N devices are visible -- a run that asked for N GPUs never reports a smaller n_gpus.  Returns the launcher's
    command line (the caller execs it)."""
    n_visible = visible_gpus() if n_visible is None else n_visible
    if share_one_gpu() and n_visible >= 1:
        n_visible = n_gpus
    if n_visible < n_gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node; refusing to run fewer ranks than "
                         "asked for" % (n_gpus, n_visible))
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)


def rank_plan(config, rank, world, objects=0, models=0):
    """Which objects and how many distinct models rank `rank` of `world` holds: weak scaling (rbot64) =
    `objects` per GPU, rank r owns the global objects [r n, (r + 1) n); strong scaling (ycb21, synth512) = a fixed
    total, object i -> GPU i mod G.  A rank builds and uploads only the models its own objects use."""
    pkg = importlib.import_module("3dobjecttracking_amd")
    cfg = CONFIGS[config]
    total = objects or cfg["objects"]
    if cfg["scaling"] == "weak":
        n_obj = total
        ids = pkg.sharding.shard_objects(n_obj * world, rank, world, mode="block")
        total = n_obj * world
        first = int(ids[0])
    else:
        ids = pkg.sharding.shard_objects(total, rank, world, mode="round_robin")
        n_obj = len(ids)
        first = rank * 1000  # seeds of this rank's rendered streams
    n_streams = min(n_obj, 64)
    n_models = min(models or cfg["models"], n_streams)
    return {"global_ids": [int(i) for i in ids], "n_obj": n_obj, "total_objects": total, "first_object": first,
            "n_streams": n_streams, "n_models": n_models,
            "model_of": [(i % n_streams) % n_models for i in range(n_obj)] if n_obj else []}


def live_rccl_ranks(dist, hip=None):
    """How many ranks RCCL really spans, asked of the live communicator -- never echoed from argv / WORLD_SIZE: the
    library's own communicator (chain8: ncclCommCount through m3t_hip_comm_get_rank_count) when `hip` is given, else
    torch.distributed's process group if its backend is nccl (= RCCL on ROCm); 0 under gloo or without a group."""
    if hip is not None:
        n = C.c_int(0)
        hip.call("comm_get_rank_count", C.byref(n))
        return int(n.value)
    if dist is None or not dist.is_initialized():
        return 0
    return int(dist.get_world_size()) if str(dist.get_backend()).lower() == "nccl" else 0


def rank_share_points(pkg, scenes, inputs_full, cfg, config, counts, objects, models, use_depth, K=10, W=3, regions=7):
    """`--rank-share`: rank 0's share at N ranks (rank_plan: the objects it would own), run ALONE on this GPU.  The
    projected whole-job rate is (objects of all ranks) x K / (rank 0's time): what N GPUs deliver if every rank takes
    as long as rank 0 does alone -- no transport (the rigid configurations have none on the data path), no barrier
    skew, no host contention between N processes.  A projection, labelled as such; the driver's N-GPU runs measure."""
    n_frames = K + W + 1
    points, base_rate = [], None
    for n in counts:
        plan = rank_plan(config, 0, n, objects, models)
        if cfg["scaling"] == "weak":
            ids = list(range(plan["n_obj"]))  # every rank holds the same number of objects
        else:
            ids = [i for i in plan["global_ids"]]  # object i -> GPU i mod N: rank 0's are 0, N, 2 N, ...
        if not ids:
            continue
        hip = pkg.open_context(0)
        sub = scenes.subset(inputs_full, ids)
        inst = scenes.Instance(hip, sub, use_depth=use_depth)
        scenes.stage_frames(hip, inst, sub, n_frames)
        hip.call("cameras_select_slot", 0)
        hip.call("start_modalities", 0)

        def run(first, count):
            for k in range(first, first + count):
                hip.call("cameras_select_slot", k)
                hip.call("execute_tracking_step", k)

        run(1, W)
        hip.call("sync")
        restart = np.stack([np.ascontiguousarray(sub.gt[i][W].T, np.float32).reshape(16) for i in range(len(ids))])
        times = []
        for _ in range(regions):
            hip.call("bodies_set_poses", restart.ctypes.data_as(C.POINTER(C.c_float)), len(ids))
            hip.call("sync")
            t = time.perf_counter()
            run(1 + W, K)
            hip.call("sync")
            times.append(time.perf_counter() - t)
        el = float(np.median(times))
        shape = (C.c_int * 4)()
        hip.call("get_step_shape", shape)
        name = C.create_string_buffer(64)
        hip.call("get_step_kernel", name, 64)
        total = plan["total_objects"]
        rate = total * K / el
        if base_rate is None:
            base_rate = rate / n
        points.append({"n_gpus": n, "objects_on_rank_0": len(ids), "objects_in_all": total,
                       "rank_0_ms_per_step": round(el / K * 1e3, 4), "kernel": name.value.decode(),
                       "workgroups_per_object": int(shape[1]),
                       "projected_pose_updates_per_s": round(rate, 1),
                       "projected_efficiency_vs_n_times_the_first_point": round(rate / (base_rate * n), 4)})
        del inst, hip
    return {"what": "PROJECTION, not a measurement of N GPUs: rank 0's share (bench.rank_plan) timed alone on ONE GPU; "
                    "projected rate = objects of all ranks x steps / rank 0's time; no transport (none on the data path "
                    "of the rigid configurations), no barrier skew, no contention between N host processes",
            "steps": K, "warmup": W, "regions": regions, "points": points}


def flush_stdio():
    sys.stdout.flush()
    try:
        C.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        cmd = launch_ranks(args.gpus, sys.argv[1:])
        os.execv(cmd[0], cmd)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (args.gpus, world))
    import torch
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dry_run = share_one_gpu() and world > 1
    if dry_run:
        local_rank = 0  # (chain8: RCCL refuses two ranks on one device -- the link sums cross over gloo, bench_chain.run)
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (rank, torch.cuda.device_count()))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if dry_run:
            import datetime
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))  # (a mismatched collective fails, not hangs)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # N ranks build their inputs (rendering, model sampling: numpy) side by side on one host: every rank keeps to
        # its share of the usable cores instead of N thread pools of full width fighting over the container's quota
        try:
            from threadpoolctl import threadpool_limits
            threadpool_limits(limits=max(1, int(usable_cpus()["usable"]) // world))
        except Exception:  # noqa: BLE001 (a convenience, not a requirement)
            pass

    # the synthetic inputs are generated on worker processes (bench_inputs.py: one task per model / per
    # object stream, the same bits as in one process, serial fallback; at N > 1 every rank keeps to its share of the cores)
    os.environ.setdefault("M3T_INPUT_WORKERS", "auto" if world == 1 else str(max(1, int(usable_cpus()["usable"]) // world)))
    pkg = importlib.import_module("3dobjecttracking_amd")
    if args.config == "chain8":
        import bench_chain
        out = bench_chain.run(args, pkg, rank, local_rank, world, dist, torch, open_oracle, measured_traffic, dry_run)
        if out is not None and dry_run:
            out["dry_run"] = ("M3T_BENCH_SHARE_ONE_GPU=1: %d ranks on ONE GPU, the link sums summed over gloo through "
                              "m3t_hip_comm_set_reduce_callback -- the N-rank code path (tracking_step_tree_segment_kernel with "
                              "partial ownership), not a measurement" % world)
            out["metric"] = "[DRY RUN, not a measurement] " + out["metric"]
    else:
        import bench_inputs  # (beside this file: the synthetic batches, their worker processes and cache)
        out = run_objects(args, pkg, bench_inputs, rank, local_rank, world, dist, torch, dry_run)
        if out is not None and dry_run:
            out["dry_run"] = "M3T_BENCH_SHARE_ONE_GPU=1: %d ranks on ONE GPU over gloo, one workgroup per object -- the N-rank code path, not a measurement" % world
            out["metric"] = "[DRY RUN, not a measurement] " + out["metric"]
    # the JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio when a communicator is
    # made, which a redirected stdout holds back until the process ends -- behind the line.  Push it out first (every
    # rank, before the last barrier), take the process group down, then print.
    flush_stdio()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    flush_stdio()
    if rank == 0:
        print(json.dumps(out), flush=True)


def run_objects(args, pkg, scenes, rank, local_rank, world, dist, torch, dry_run=False):
    cfg = CONFIGS[args.config]
    syn = pkg.synthetic
    hip = pkg.open_context(local_rank)
    if dry_run:
        hip.call("set_object_split", 0)
    K, W = args.steps, args.warmup
    n_frames = K + W + 1
    use_depth = cfg["with_depth"]
    t0 = time.time()
    plan = rank_plan(args.config, rank, world, args.objects, args.models)
    n_obj, total_objects, n_streams, n_models = plan["n_obj"], plan["total_objects"], plan["n_streams"], plan["n_models"]
    base = scenes.Inputs(n_streams, n_frames, n_divides=args.n_divides, n_models=n_models, with_depth=use_depth,
                         first_object=plan["first_object"])
    inputs = scenes.replicate(base, n_obj)
    inst = scenes.Instance(hip, inputs, use_depth=use_depth)
    scenes.stage_frames(hip, inst, inputs, n_frames)
    setup_s = time.time() - t0

    def barrier():
        hip.call("sync")
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(first_frame, count):
        for k in range(first_frame, first_frame + count):
            hip.call("cameras_select_slot", k)
            hip.call("execute_tracking_step", k)

    def get_poses():
        poses = np.zeros((n_obj, 16), np.float32)
        hip.call("bodies_get_poses", poses.ctypes.data_as(C.POINTER(C.c_float)), n_obj)
        return poses

    hip.call("cameras_select_slot", 0)
    hip.call("start_modalities", 0)
    run(1, W)
    barrier()
    t = time.perf_counter()
    run(1 + W, K)
    barrier()
    elapsed = pkg.sharding.max_over_ranks(time.perf_counter() - t, dist, device="cuda")
    poses = get_poses()
    # the same K steps again, args.repeats - 1 more times (restarted from the ground-truth pose of frame W; the
    # trajectory checked below is the first one): min / median of the timed region, MAX over ranks each
    times = [elapsed]
    restart = np.stack([np.ascontiguousarray(inputs.gt[i][W].T, np.float32).reshape(16) for i in range(n_obj)])
    # (every rank repeats the same number of times: the count is fixed from rank 0's first region)
    n_repeats = args.repeats if args.repeats > 0 else int(min(4000, max(5, args.busy_seconds / max(elapsed, 1e-6))))
    if dist is not None:
        tn = torch.tensor([n_repeats], device="cuda")
        dist.broadcast(tn, 0)
        n_repeats = int(tn.item())
    for _ in range(max(0, n_repeats - 1)):
        hip.call("bodies_set_poses", restart.ctypes.data_as(C.POINTER(C.c_float)), n_obj)
        barrier()
        t = time.perf_counter()
        run(1 + W, K)
        barrier()
        times.append(pkg.sharding.max_over_ranks(time.perf_counter() - t, dist, device="cuda"))
    elapsed = float(np.median(times))
    tracked, adds_gt = 0, []
    for i in range(n_obj):
        e = syn.pose_errors(poses[i].reshape(4, 4).T, inputs.gt[i][W + K])
        tracked += int(e[0] < np.deg2rad(5) and e[1] < 0.05)  # rbot_evaluator.cpp:416-433
        if i < 64:
            adds_gt.append(syn.add_s(inputs.vertices[i], poses[i].reshape(4, 4).T, inputs.gt[i][W + K]))

    # ---- roofline leg: HIP events around the kernels on the context stream (rank 0) ----
    roofline = None
    if rank == 0:
        # (a) ONE event pair around the K launches of a timed region (nothing between the launches): the device time
        # per launch, which is what `achieved` is computed from; (b) an event pair around every launch, as round 3
        # measured it: each pair also times the launch gap in front of its kernel and the markers slow the stream, so
        # its mean comes out above (a) -- and above the host-clocked ms_per_step -- by the marker overhead reported
        hip.call("bodies_set_poses", restart.ctypes.data_as(C.POINTER(C.c_float)), n_obj)
        ms = (C.c_float * 2)()
        cnt = (C.c_int * 2)()
        regions = []
        for _ in range(15):  # (the median of 15 regions, like ms_per_step is a median over its regions)
            hip.call("bodies_set_poses", restart.ctypes.data_as(C.POINTER(C.c_float)), n_obj)
            hip.call("set_kernel_timing", 2)
            run(1 + W, K)
            hip.call("get_kernel_timing", ms, cnt)
            regions.append(float(ms[0]))
        region_ms, region_launches = float(np.median(regions)), int(cnt[0]) + int(cnt[1])
        fused_hist = cnt[1] == 0  # the histogram update rode in the tracking launch
        hip.call("bodies_set_poses", restart.ctypes.data_as(C.POINTER(C.c_float)), n_obj)
        hip.call("set_kernel_timing", 1)
        run(1 + W, K)
        hip.call("get_kernel_timing", ms, cnt)
        hip.call("set_kernel_timing", 0)
        shape = (C.c_int * 4)()
        hip.call("get_step_shape", shape)  # objects, workgroups per object, threads, histogram update fused
        name = C.create_string_buffer(64)
        hip.call("get_step_kernel", name, 64)
        kernel = name.value.decode()
        pair_ms = ms[0] / max(cnt[0], 1)
        hist_ms = ms[1] / max(cnt[1], 1)
        # one launch per step when the histogram update is fused: the region mean IS the kernel's mean duration (plus
        # the launch gap); with a separate histogram kernel the pairs' split of the region is used
        track_ms = region_ms / max(region_launches, 1) if fused_hist else region_ms / K * pair_ms / max(pair_ms + hist_ms, 1e-9)
        alg = cfg["alg"] if fused_hist else cfg["alg_track"]
        achieved = alg * n_obj / (track_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(args.config, kernel, n_obj, fused_hist)
        roofline = {"bound": "hbm",
                    "kernel": kernel + (" (whole step incl. histogram update)" if fused_hist else ""),
                    "workgroups_per_object": shape[1], "threads_per_workgroup": shape[2],
                    "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                    "kernel_ms": round(track_ms, 4), "algorithmic_bytes_per_launch": alg * n_obj,
                    "kernel_ms_how": "HIP events on the context's stream: one pair around the %d launches of a timed "
                                     "region / %d, median of 15 regions (min %.4f)" %
                                     (region_launches, region_launches, min(regions) / max(region_launches, 1)),
                    "kernel_ms_event_pair_per_launch": round(pair_ms, 4),
                    "event_pair_overhead_ms": round(pair_ms - track_ms, 4)}
        if not fused_hist:
            roofline["histogram_kernel_ms"] = round(hist_ms, 4)

    # ---- the evaluators' four time buckets (rbot_evaluator.cpp:354-414) on the device, one launch per sub-step
    # (m3t_hip_set_fused_step(0)), each bucket synchronised and timed on the host: what the fused launch replaces ----
    buckets = None
    if rank == 0 and world == 1 and args.config in ("rbot64", "ycb21") and not args.no_buckets:
        try:  # (extra legs never cost the line: the headline above is already measured)
            buckets = device_buckets(hip, restart, n_obj, W, min(K, 5), cfg)
        except Exception as e:  # noqa: BLE001
            buckets = {"error": str(e)[:300]}

    # ---- host-buffer (PCIe-inclusive) rate: every step first receives its frames from host memory; never `value` ----
    pcie = None
    if rank == 0 and world == 1 and args.config == "rbot64" and not args.no_pcie:
        try:
            pcie = pcie_legs(hip, inst, inputs, n_obj, W, K)
        except Exception as e:  # noqa: BLE001
            pcie = {"error": str(e)[:300]}

    # ---- optional batch sweep (extra, not the headline) ----
    sweep = []
    if rank == 0 and args.sweep:
        for n in [int(x) for x in args.sweep.split(",") if x]:
            sweep.append(batch_point(pkg, scenes, base, n, use_depth, cfg))

    # ---- CPU baseline: the oracle restatement, bounded sample (rank 0); its first pass over the frames is also
    # the parity check of the benchmarked trajectory: the same objects, the same frames, free running ----
    cpu, parity = None, None
    if rank == 0 and not args.no_cpu_baseline:
        # N > 1: rank 0 still checks its own first objects against the oracle and times a short 1-thread sample (the
        # other ranks wait at the final barrier); the all-cores and native-build legs run at N = 1 only
        if world > 1:
            args.cpu_seconds, args.no_cpu_parallel = min(args.cpu_seconds, 4.0), True
        ora = open_oracle()
        n_cpu = min(8, n_obj)
        sub = scenes.subset(inputs, list(range(n_cpu)))
        oinst = scenes.Instance(ora, sub, use_depth=use_depth)
        oinst.upload_frame(0)
        oinst.tracker.StartModalities(0)
        done, spent, first_pass = 0, 0.0, True
        while spent < args.cpu_seconds or first_pass:
            for k in range(1, n_frames):
                oinst.upload_frame(k)  # excluded from the timed region (as for the GPU)
                tc = time.perf_counter()
                oinst.tracker.ExecuteTrackingStep(k)
                spent += time.perf_counter() - tc
                done += n_cpu
                if first_pass and k == W + K:
                    # pose after the last timed frame: HIP (first timed run) vs oracle, ADD-S as
                    # ycb_evaluator.cpp:816-831, rotation / translation as rbot_evaluator.cpp:416-433
                    op = oinst.poses()
                    errs = [syn.pose_errors(poses[i].reshape(4, 4).T, op[i]) for i in range(n_cpu)]
                    adds = [syn.add_s(inputs.vertices[i], poses[i].reshape(4, 4).T, op[i]) for i in range(n_cpu)]
                    parity = {"rot_max": float(max(e[0] for e in errs)), "trans_max": float(max(e[1] for e in errs)),
                              "add_s_max": float(max(adds)), "n": n_cpu, "frames": W + K,
                              "bit_identical": bool(all(np.array_equal(poses[i].reshape(4, 4).T, op[i])
                                                        for i in range(n_cpu))),
                              "what": "body2world after %d free-running frames, HIP (benchmarked launch shape) vs "
                                      "oracle, objects 0..%d" % (W + K, n_cpu - 1)}
                if spent >= args.cpu_seconds and not first_pass:
                    break
            first_pass = False
            oinst.set_poses([inputs.gt[i][0] for i in range(n_cpu)])
        # parity over ALL timed objects (round 6): the whole batch in one oracle context, stepped with the OpenMP loop
        # over objects (independent objects: the same bits as the serial loop), same frames, free running.  Batches
        # whose oracle pass would take more than ~10 s of host time are sampled (64 objects, the last one included).
        try:
            all_parity = parity_all_objects(scenes, inputs, poses, n_obj, W, K, use_depth)
            if all_parity is not None:
                all_parity["first_objects_serial"] = {k: parity[k] for k in ("rot_max", "trans_max", "add_s_max", "n", "bit_identical")}
                parity = all_parity
        except Exception as e:  # noqa: BLE001 (the 8-object check above stands)
            parity["all_objects_error"] = str(e)[:200]
        cpu_parallel = None
        if not args.no_cpu_parallel:
            cpu_parallel = cpu_all_cores(scenes, scenes.replicate(base, min(n_obj, 64)), min(n_obj, 64), n_frames,
                                         use_depth)
        native = None
        try:  # the same restatement built as the reference builds (M3T/CMakeLists.txt:73-80), on this host
            if world > 1:
                raise RuntimeError("N = 1 only")
            nat = open_oracle(native=True)
            ninst = scenes.Instance(nat, sub, use_depth=use_depth)
            ninst.upload_frame(0)
            ninst.tracker.StartModalities(0)
            n_done, n_spent = 0, 0.0
            while n_spent < min(args.cpu_seconds, 6.0):
                for k in range(1, n_frames):
                    ninst.upload_frame(k)
                    tc = time.perf_counter()
                    ninst.tracker.ExecuteTrackingStep(k)
                    n_spent += time.perf_counter() - tc
                    n_done += n_cpu
                ninst.set_poses([inputs.gt[i][0] for i in range(n_cpu)])
            native = {"value": round(n_done / n_spent, 1), "unit": "pose-updates/s", "cores": 1,
                      "build": "g++ -O3 -march=native (FMA contraction on): timing only, not bit-identical"}
            if not args.no_cpu_parallel:
                native["all_cores"] = cpu_all_cores(scenes, scenes.replicate(base, min(n_obj, 64)), min(n_obj, 64),
                                                    n_frames, use_depth, seconds=5.0, native=True)
        except Exception as e:  # noqa: BLE001 (no compiler on the box: the baseline above stands alone)
            native = {"error": str(e)[:200]}
        cpu = {"value": round(done / spent, 1), "unit": "pose-updates/s", "cores": 1, "kind": "port",
               "all_cores": cpu_parallel, "native_build": native,
               "sample": "%d pose-updates of %d of the same objects, same frames, oracle/libm3t_oracle.so "
                         "(g++ -O3 -march=x86-64-v3), 1 thread, host has %d cores" % (done, n_cpu, os.cpu_count())}

    if rank != 0:
        return None
    total = total_objects * K if cfg["scaling"] == "weak" else total_objects * K
    n_views = inputs.region_models[0][1].shape[0]
    out = {
        "metric": cfg["metric"], "value": round(total / elapsed, 1),
        "unit": "pose-updates/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": cfg["scaling"],
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["workload"] % dict(n=n_obj, views=n_views, models=len(inputs.region_models)),
                   "objects_per_gpu": n_obj,
                   "parallelism": "objects sharded over %d GPU(s) (%s), no collective on the data path" %
                                  (world, "rank r owns objects [r n, (r + 1) n)" if cfg["scaling"] == "weak"
                                   else "object i on GPU i mod %d" % world),
                   "ranks": world, "rccl_ranks": live_rccl_ranks(dist),
                   "process_group_backend": (str(dist.get_backend()) if dist is not None else None),
                   "tracked_within_5cm_5deg": "%d/%d" % (tracked, n_obj),
                   "mean_add_s_vs_ground_truth_m": round(float(np.mean(adds_gt)), 6), "setup_s": round(setup_s, 1)},
        "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
        "repeats": {"n": len(times), "ms_per_step_min": round(min(times) / K * 1e3, 4),
                    "ms_per_step_median": round(elapsed / K * 1e3, 4),
                    "ms_per_step_p95": round(float(np.percentile(times, 95)) / K * 1e3, 4),
                    "timed_seconds_total": round(float(np.sum(times)), 3),
                    "ms_per_step_first_5": [round(x / K * 1e3, 4) for x in times[:5]]},
        "pcie_inclusive": pcie, "device_buckets_unfused": buckets,
        "frac_of_hbm_roofline_whole_step": round(total / elapsed * cfg["alg"] / (HBM_PEAK_GBS * 1e9 * world), 5),
        "newton_steps_per_s": round(total / elapsed * cfg["newton"], 1),  # corr-iterations x updates
    }
    if sweep:
        out["batch_sweep"] = sweep
    if args.rank_share and world == 1:
        counts = [int(x) for x in args.rank_share.split(",") if x]
        out["projected_scaling"] = rank_share_points(pkg, scenes, inputs, cfg, args.config, counts, args.objects,
                                                     args.models, use_depth)
    if args.extras:
        out["extras"] = extras_point(pkg)
    return out


def device_buckets(hip, restart, n_obj, W, K, cfg):
    n_corr, n_update = (7, 2) if cfg["newton"] == 14 else (4, 2)
    hip.call("bodies_set_poses", restart.ctypes.data_as(C.POINTER(C.c_float)), n_obj)
    hip.call("set_fused_step", 0)
    t = [0.0, 0.0, 0.0, 0.0]

    def timed(i, name, *a):
        t0 = time.perf_counter()
        hip.call(name, *a)
        hip.call("sync")
        t[i] += time.perf_counter() - t0

    for k in range(1 + W, 1 + W + K):
        hip.call("cameras_select_slot", k)
        for c in range(n_corr):
            timed(0, "calculate_correspondences", k, c)
            for u in range(n_update):
                timed(1, "calculate_gradient_and_hessian", k, c, u)
                timed(2, "calculate_optimization", k, c, u)
        timed(3, "calculate_results", k)
    hip.call("set_fused_step", 1)
    tot = sum(t)
    return {"ms_per_step": {"correspondences": round(t[0] / K * 1e3, 4), "gradient_hessian": round(t[1] / K * 1e3, 4),
                            "optimization": round(t[2] / K * 1e3, 4), "results": round(t[3] / K * 1e3, 4),
                            "total": round(tot / K * 1e3, 4)},
            "share": {"correspondences": round(t[0] / tot, 3), "gradient_hessian": round(t[1] / tot, 3),
                      "optimization": round(t[2] / tot, 3), "results": round(t[3] / tot, 3)},
            "note": "one launch per sub-step (%d launches per frame), host-timed with a sync after every launch; the "
                    "fused launch does the same work in ms_per_step" % (n_corr * (1 + 2 * n_update) + 1)}


def pcie_legs(hip, inst, inputs, n_obj, W, K):
    n_up = max(1, min(5, K - 1))  # the asynchronous leg stages one frame ahead
    hip.call("cameras_select_slot", 0)
    hip.call("sync")
    tu = time.perf_counter()
    for k in range(1 + W, 1 + W + n_up):
        for i, cam in enumerate(inst.color_cams):
            f = inputs.color[i][k]
            hip.call("camera_upload", cam.id, f.ctypes.data_as(C.c_void_p), f.strides[0])
        hip.call("execute_tracking_step", k)
    hip.call("sync")
    el = time.perf_counter() - tu
    frame_bytes = sum(inputs.color[i][0].nbytes for i in range(n_obj))
    pcie = {"pose_updates_per_s": round(n_obj * n_up / el, 1), "ms_per_step": round(el / n_up * 1e3, 3),
            "host_bytes_per_step": frame_bytes, "upload_GBs": round(frame_bytes * n_up / el / 1e9, 2),
            "note": "pageable host frames, synchronous m3t_hip_camera_upload per camera, then the step (%d steps)" % n_up}
    # the same with ONE page-locked slab per batch-frame and the double-buffered asynchronous ingest: the 64 frames of
    # step k+1 cross PCIe as one DMA on the copy stream while step k runs (m3t_hip_cameras_upload_batch_async)
    # (the streaming legs run longer than the pageable one: their pipelines -- the copy of frame k + 1 beside step k --
    # take a step to fill, which a 5-step loop would charge with a fifth of a step)
    n_up = max(1, min(12, K - 1))
    blocks = [np.stack([inputs.color[i][k] for i in range(n_obj)]) for k in range(1 + W, 2 + W + n_up)]
    ids = (C.c_int * n_obj)(*[cam.id for cam in inst.color_cams])
    for b in blocks:
        inst.tracker.register_host_buffer(b)

    def upload(slot, block):
        hip.call("cameras_upload_batch_async", ids, n_obj, slot, block.ctypes.data_as(C.c_void_p),
                 block.strides[0], block.strides[1])

    hip.call("cameras_set_ring", ids, n_obj, 2)  # one ring for the batch: slot s of all cameras is one block
    upload(0, blocks[0])
    hip.call("ingest_sync")
    hip.call("sync")
    tu = time.perf_counter()
    for j in range(n_up):
        hip.call("cameras_select_slot", j % 2)
        hip.call("execute_tracking_step", 1 + W + j)
        upload((j + 1) % 2, blocks[j + 1])
    hip.call("ingest_sync")
    hip.call("sync")
    el = time.perf_counter() - tu
    el_full = el
    # the same again with ROI ingest: only the trackers' rectangle of every frame is pulled out of the slab (one kernel
    # per batch-frame on the copy stream, m3t_hip_cameras_upload_batch_roi_async); poses must not change, and the
    # device-side check must not report a body that left its rectangle
    roi = None
    if "set_roi_ingest" in hip._fn:
        ref = (C.c_float * (16 * n_obj))()
        hip.call("bodies_get_poses", ref, n_obj)
        ref = np.array(ref)
        restart = np.stack([np.ascontiguousarray(inputs.gt[i][W].T, np.float32).reshape(16) for i in range(n_obj)])

        start_block = np.stack([inputs.color[i][W] for i in range(n_obj)])
        inst.tracker.register_host_buffer(start_block)
        # pixels: two frames of this workload's motion (the rectangle comes from the pose two frames back); a body that
        # outruns it is repeated on the whole frame inside the step, so the margin is a cost knob, not a correctness one
        margin = float(os.environ.get("M3T_BENCH_ROI_MARGIN", "24"))  # (developer: other margins)

        def timed(roi_on):
            # both runs from the same state: poses of frame W, histograms initialised on frame W (whole frames)
            # (roi_on = 2: adaptive margins -- three times what a body's rectangle moved over the last step, at most `margin`)
            hip.call("set_roi_ingest", int(roi_on), C.c_float(margin))
            hip.call("bodies_set_poses", restart.ctypes.data_as(C.POINTER(C.c_float)), n_obj)
            upload(1, start_block)
            hip.call("ingest_sync")
            hip.call("cameras_select_slot", 1)
            hip.call("start_modalities", W)
            upload(0, blocks[0])  # (whole frames: no step recorded yet)
            hip.call("ingest_sync")
            hip.call("sync")
            fn = "cameras_upload_batch_roi_async" if roi_on else "cameras_upload_batch_async"
            t0 = time.perf_counter()
            for j in range(n_up):
                hip.call("cameras_select_slot", j % 2)
                hip.call("execute_tracking_step", 1 + W + j)
                b = blocks[j + 1]
                hip.call(fn, ids, n_obj, (j + 1) % 2, b.ctypes.data_as(C.c_void_p), b.strides[0], b.strides[1])
            hip.call("ingest_sync")
            hip.call("sync")
            dt = time.perf_counter() - t0
            out = (C.c_float * (16 * n_obj))()
            hip.call("bodies_get_poses", out, n_obj)
            return dt, np.array(out)

        def median_of(roi_on, n=5):
            runs = [timed(roi_on) for _ in range(n)]
            assert all(np.array_equal(r[1], runs[0][1]) for r in runs)
            return float(np.median([r[0] for r in runs])), runs[0][1]

        dt_full, poses_full = median_of(False)
        dt_roi, poses_roi = median_of(True)
        bodies = (C.c_int * 64)()
        n_miss, pulls = C.c_int(0), C.c_longlong(0)
        hip.call("roi_get_status", bodies, 64, C.byref(n_miss), C.byref(pulls))
        roi = {"pose_updates_per_s": round(n_obj * n_up / dt_roi, 1), "ms_per_step": round(dt_roi / n_up * 1e3, 3),
               "whole_frames_same_loop_ms_per_step": round(dt_full / n_up * 1e3, 3),
               "rectangle_uploads": int(pulls.value), "bodies_outside_their_rectangle": int(n_miss.value),
               "bit_identical_to_whole_frames": bool(np.array_equal(poses_roi, poses_full)), "margin_px": margin,
               "timed": "median of 5 runs of %d steps" % n_up,
               "note": "m3t_hip_cameras_upload_batch_roi_async: one pull kernel per batch-frame over the mapped slab; the "
                       "steps run the guarded kernels (a body that leaves its rectangle is repeated on the whole frame inside "
                       "the step: counted in bodies_outside_their_rectangle)"}
        # ... and with CUs of its own for the pull kernel (m3t_hip_reserve_ingest_cus: CU-masked streams): frame k + 1
        # crosses PCIe WHILE step k runs on the other CUs
        if "reserve_ingest_cus" in hip._fn:
            roi["reserved_cus"] = []
            for n_cus in [int(x) for x in os.environ.get("M3T_BENCH_RESERVE_CUS", "32,64").split(",")]:  # (developer: other counts)
                hip.call("reserve_ingest_cus", n_cus)
                dt_res, poses_res = median_of(True)
                shape = (C.c_int * 4)()
                hip.call("get_step_shape", shape)
                hip.call("roi_get_status", bodies, 64, C.byref(n_miss), C.byref(pulls))
                entry = {"cus_for_the_pull": n_cus, "pose_updates_per_s": round(n_obj * n_up / dt_res, 1),
                         "ms_per_step": round(dt_res / n_up * 1e3, 3), "workgroups_per_object": int(shape[1]),
                         "bodies_outside_their_rectangle": int(n_miss.value),
                         "bit_identical_to_whole_frames": bool(np.array_equal(poses_res, poses_full))}
                # ... and with per-body margins from the motion over the last step (fewer bytes)
                dt_ad, poses_ad = median_of(2)
                hip.call("roi_get_status", bodies, 64, C.byref(n_miss), C.byref(pulls))
                entry["adaptive_margins"] = {"pose_updates_per_s": round(n_obj * n_up / dt_ad, 1),
                                             "ms_per_step": round(dt_ad / n_up * 1e3, 3),
                                             "bodies_repeated_on_whole_frames": int(n_miss.value),
                                             "bit_identical_to_whole_frames": bool(np.array_equal(poses_ad, poses_full))}
                roi["reserved_cus"].append(entry)
            hip.call("reserve_ingest_cus", 0)
        hip.call("set_roi_ingest", 0, C.c_float(0.0))
    for b in blocks:
        inst.tracker.unregister_host_buffer(b)
    el = el_full
    pcie["roi_rectangles"] = roi
    pcie["async_pinned"] = {"pose_updates_per_s": round(n_obj * n_up / el, 1),
                            "ms_per_step": round(el / n_up * 1e3, 3),
                            "upload_GBs": round(frame_bytes * n_up / el / 1e9, 2),
                            "note": "one page-locked slab per batch-frame, m3t_hip_cameras_upload_batch_async on the "
                                    "copy stream, two ring slots; the copy of frame k+1 overlaps step k (%d steps)" % n_up}
    return pcie


def usable_cpus():
    """cores this process may really use: the scheduler affinity, capped by the cgroup CPU quota of the container
    (os.cpu_count() reports the host's hardware threads, which a container with a quota cannot all run on)"""
    info = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "cgroup_cpus": None}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    info["cgroup_cpus"] = float(txt[0]) / float(txt[1])
            else:
                quota = float(txt[0])
                period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    info["cgroup_cpus"] = quota / period
            break
        except Exception:
            continue
    n = info["affinity"]
    if info["cgroup_cpus"]:
        n = min(n, max(1, int(info["cgroup_cpus"])))
    info["usable"] = max(1, n)
    return info


def parity_all_objects(scenes, inputs, poses, n_obj, W, K, use_depth, budget_pose_updates=40000):
    """HIP (the first timed trajectory, `poses` after frame W + K) against the oracle over every object of the batch:
    one oracle context holding the checked objects, m3t_oracle_execute_tracking_step_parallel over the same W + K frames.
    rot / trans as rbot_evaluator.cpp:416-433, ADD-S as ycb_evaluator.cpp:816-831."""
    syn = importlib.import_module("3dobjecttracking_amd").synthetic
    frames = W + K
    if n_obj * frames <= budget_pose_updates:
        ids = list(range(n_obj))
    else:
        ids = sorted(set(np.linspace(0, n_obj - 1, 64).astype(int).tolist()))
    ora = open_oracle()
    f = ora.lib.m3t_oracle_execute_tracking_step_parallel
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    sub = inputs if len(ids) == n_obj else scenes.subset(inputs, ids)
    inst = scenes.Instance(ora, sub, use_depth=use_depth)
    inst.upload_frame(0)
    inst.tracker.StartModalities(0)
    n_threads = min(usable_cpus()["usable"], len(ids))
    buckets = (C.c_double * 4)()
    t = time.perf_counter()
    for k in range(1, frames + 1):
        inst.upload_frame(k)
        rc = f(ora.ctx, k, n_threads, buckets)
        assert rc == 0, ora.last_error()
    spent = time.perf_counter() - t
    op = inst.poses()
    mine = [poses[i].reshape(4, 4).T for i in ids]
    errs = [syn.pose_errors(mine[j], op[j]) for j in range(len(ids))]
    adds = [syn.add_s(inputs.vertices[i], mine[j], op[j]) for j, i in enumerate(ids)]
    return {"rot_max": float(max(e[0] for e in errs)), "trans_max": float(max(e[1] for e in errs)),
            "add_s_max": float(max(adds)), "n": len(ids), "of": n_obj, "frames": frames,
            "bit_identical": bool(all(np.array_equal(mine[j], op[j]) for j in range(len(ids)))),
            "oracle_seconds": round(spent, 2), "oracle_threads": n_threads,
            "what": "body2world after %d free-running frames, HIP (benchmarked launch shape, first timed run) vs oracle, "
                    "%s" % (frames, "ALL %d timed objects" % n_obj if len(ids) == n_obj else
                            "%d of the %d timed objects (evenly spaced, first and last included)" % (len(ids), n_obj))}


def cpu_all_cores(scenes, inputs, n_obj, n_frames, use_depth, seconds=8.0, native=False):
    """SURVEY 8(d) CPU baseline (ii): the batch (at most 64 objects) in ONE oracle context, stepped with an OpenMP
    `parallel for` over the objects at nproc threads (m3t_oracle_execute_tracking_step_parallel; what the
    reference's evaluators do over sequences, rbot_evaluator.cpp:144).  Also the evaluators' four time buckets
    (rbot_evaluator.cpp:354-414), summed over threads."""
    ora = open_oracle(native)
    f = ora.lib.m3t_oracle_execute_tracking_step_parallel
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    inst = scenes.Instance(ora, inputs, use_depth=use_depth)
    inst.upload_frame(0)
    inst.tracker.StartModalities(0)
    host = usable_cpus()
    n_threads = min(host["usable"], n_obj)
    buckets = (C.c_double * 4)()
    done, spent = 0, 0.0
    while spent < seconds:
        for k in range(1, n_frames):
            inst.upload_frame(k)
            tc = time.perf_counter()
            rc = f(ora.ctx, k, n_threads, buckets)
            spent += time.perf_counter() - tc
            assert rc == 0, ora.last_error()
            done += n_obj
            if spent >= seconds:
                break
        inst.set_poses([inputs.gt[i][0] for i in range(n_obj)])
    tot = sum(buckets) or 1.0
    return {"value": round(done / spent, 1), "unit": "pose-updates/s", "cores": n_threads, "host": host,
            "sample": "%d pose-updates, %d objects in one oracle context, OpenMP parallel for over objects, "
                      "%d threads" % (done, n_obj, n_threads),
            "bucket_share": {"correspondences": round(buckets[0] / tot, 3), "gradient_hessian": round(buckets[1] / tot, 3),
                             "optimization": round(buckets[2] / tot, 3), "results": round(buckets[3] / tot, 3)},
            "thread_us_per_pose_update": round(tot / done * 1e6, 1)}


def measured_traffic(config, kernel, n_obj, fused_histogram=False):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/rNN_hbm_traffic_*.json:
    FETCH_SIZE and WRITE_SIZE in separate runs of this same command; the read side corrected by the factor measured on
    this hardware in the kernels' own access pattern, profiles/rNN_counter_calibration.txt -- MI355X_MICROARCH.md §HBM
    calibrates wide streams only).  The counters cannot be collected from inside this process; null when no profile of this
    configuration, batch size, kernel and launch structure exists."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic*.json")), reverse=True):
        try:
            d = json.load(open(path))
            if d.get("config", "rbot64") != config or d.get("objects_per_launch") != n_obj:
                continue
            if bool(d.get("histogram_update_fused", False)) != fused_histogram or kernel not in d["kernels"]:
                continue
            src = os.path.relpath(path, ROOT)
            if d.get("correction"):  # (round 5: the factor is measured, tools/ubench_counters.hip)
                src += "; correction: FETCH_SIZE x %.3f (%s); bracket [raw, read side doubled] = %s" % (
                    d["correction"]["FETCH_SIZE"], d["correction"]["how"],
                    d["kernels"][kernel].get("hbm_bytes_per_launch_bracket"))
            return d["kernels"][kernel]["hbm_bytes_per_launch_corrected"], src
        except Exception:
            continue
    return None, None


def extras_point(pkg):
    """Rows f-1 / a14 on the reference's own fixture (tests/golden): wall time of generating the default
    region + depth model of the triangle body (2 x 2562 views at 2000 x 2000) and of one tracking step of
    Region + Depth modality with region checking, silhouette checking and modelled occlusions behind the
    20 950-triangle bottle (4 focused renderers, refreshed before each of the 7 correspondence searches)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))  # (this extra leg runs on the reference's own test fixture)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import golden_scene as gs
    import util
    from util import host
    api = pkg.open_context(0)
    tv, tf = pkg.config.load_obj(os.path.join(util.GOLDEN, "_body/triangle.obj"))
    body = host.Body(api, gs.mtv.body2world())
    body.set_geometry(tv, tf, np.asarray(gs.mtv.GEOMETRY2BODY, np.float32), body_id=150, region_id=150)
    t0 = time.perf_counter()
    host.RegionModel.generate(api, body)
    t1 = time.perf_counter()
    host.DepthModel.generate(api, body)
    t2 = time.perf_counter()
    api = pkg.open_context(0)
    f = gs.TrackerFixture(api, measure_occlusions=False, region_params=dict(n_unoccluded_iterations=0),
                          depth_params=dict(n_unoccluded_iterations=0))
    geometry, _ = gs.fixture_renderer_geometry(api, f.body)
    cd = host.FocusedBasicDepthRenderer(api, geometry, f.color_camera)
    cs = host.FocusedSilhouetteRenderer(api, geometry, f.color_camera, id_type=1)
    dd = host.FocusedBasicDepthRenderer(api, geometry, f.depth_camera)
    ds = host.FocusedSilhouetteRenderer(api, geometry, f.depth_camera, id_type=0)
    for r in (cd, cs, dd, ds):
        r.AddReferencedBody(f.body)
    f.region.ModelOcclusions(cd)
    f.region.UseRegionChecking(cs)
    f.depth.ModelOcclusions(dd)
    f.depth.UseSilhouetteChecking(ds)
    start = f.body.body2world_pose()
    f.tracker.StartModalities(0)
    f.tracker.ExecuteTrackingStep(0)
    api.call("sync")
    n = 20
    t3 = time.perf_counter()
    for _ in range(n):
        f.body.set_body2world_pose(start)
        f.tracker.ExecuteTrackingStep(0)
    api.call("sync")
    t4 = time.perf_counter()
    name = C.create_string_buffer(64)
    api.call("get_step_kernel", name, 64)
    shape1 = (C.c_int * 4)()
    api.call("get_step_shape", shape1)
    # the same scene 64 times in one context (64 bodies behind 64 bottles, 256 focused renderers = 128 pairs of twins
    # drawn once each, refreshed before each of the 7 searches): the renderer-fed step of a batch
    api64 = pkg.open_context(0)
    fixtures = []
    for _ in range(64):
        g = gs.TrackerFixture(api64, measure_occlusions=False, region_params=dict(n_unoccluded_iterations=0),
                              depth_params=dict(n_unoccluded_iterations=0))
        geo, _ = gs.fixture_renderer_geometry(api64, g.body)
        rs = (host.FocusedBasicDepthRenderer(api64, geo, g.color_camera),
              host.FocusedSilhouetteRenderer(api64, geo, g.color_camera, id_type=1),
              host.FocusedBasicDepthRenderer(api64, geo, g.depth_camera),
              host.FocusedSilhouetteRenderer(api64, geo, g.depth_camera, id_type=0))
        for r in rs:
            r.AddReferencedBody(g.body)
        g.region.ModelOcclusions(rs[0])
        g.region.UseRegionChecking(rs[1])
        g.depth.ModelOcclusions(rs[2])
        g.depth.UseSilhouetteChecking(rs[3])
        fixtures.append((g, rs))
    tracker64 = fixtures[0][0].tracker
    start64 = (C.c_float * (16 * 128))()  # every fixture made two bodies: the tracked triangle and the bottle
    api64.call("bodies_get_poses", start64, 128)
    tracker64.StartModalities(0)
    tracker64.ExecuteTrackingStep(0)
    api64.call("sync")
    poses64 = np.stack([g.body.body2world_pose() for g, _ in fixtures])
    n64 = 10
    t5 = time.perf_counter()
    # (all 64 poses go back with ONE call: 64 set_body2world_pose calls cost 0.3 ms of Python per step, which is not
    # the tracker's time)
    for _ in range(n64):
        api64.call("bodies_set_poses", start64, 128)
        tracker64.ExecuteTrackingStep(0)
    api64.call("sync")
    t6 = time.perf_counter()
    name64 = C.create_string_buffer(64)
    api64.call("get_step_kernel", name64, 64)
    shape64 = (C.c_int * 4)()
    api64.call("get_step_shape", shape64)
    return {"region_model_generation_s": round(t1 - t0, 2), "depth_model_generation_s": round(t2 - t1, 2),
            "renderer_fed_kernel": name.value.decode(), "renderer_fed_shape": list(shape1),
            "renderer_fed_64_objects": {
                "ms_per_step": round((t6 - t5) / n64 * 1e3, 3),
                "pose_updates_per_s": round(64 * n64 / (t6 - t5), 1),
                "kernel": name64.value.decode(), "shape": list(shape64),
                "all_objects_end_on_the_single_object_pose": bool(all(np.array_equal(p, poses64[0]) for p in poses64)),
                "note": "64 x (Region + Depth with region / silhouette checking + modelled occlusions, 4 focused renderers "
                        "of 20 958 triangles at 200 x 200), one context, 7 x 2 iterations"},
            "model": "2562 views x 200 points, 2000 x 2000 renderings, data/_body/triangle.obj",
            "renderer_fed_tracking_step_ms": round((t4 - t3) / n * 1e3, 3),
            "renderer_fed_config": "Region + Depth, region / silhouette checking + modelled occlusions, 4 focused "
                                   "renderers of 20 958 triangles at 200 x 200, 7 x 2 iterations, 1 object"}


def batch_point(pkg, scenes, base, n_obj, use_depth, cfg):
    """pose-updates/s at another batch size (few frames; every object its own camera and frame ring up to 4096
    objects, beyond that the objects look at the 64 rendered streams through 64 shared cameras)"""
    hip = pkg.open_context(0)
    K, W = 6, 2
    n_frames = K + W + 1
    rep = scenes.replicate(base, n_obj)
    if n_obj > 4096:
        rep.camera_of = [i % base.n_objects for i in range(n_obj)]
    inst = scenes.Instance(hip, rep, use_depth=use_depth)
    scenes.stage_frames(hip, inst, rep, n_frames)
    hip.call("cameras_select_slot", 0)
    hip.call("start_modalities", 0)
    for k in range(1, 1 + W):
        hip.call("cameras_select_slot", k)
        hip.call("execute_tracking_step", k)
    hip.call("sync")
    t = time.perf_counter()
    for k in range(1 + W, 1 + W + K):
        hip.call("cameras_select_slot", k)
        hip.call("execute_tracking_step", k)
    hip.call("sync")
    el = time.perf_counter() - t
    rate = n_obj * K / el
    shape = (C.c_int * 4)()
    hip.call("get_step_shape", shape)
    name = C.create_string_buffer(64)
    hip.call("get_step_kernel", name, 64)
    return {"objects": n_obj, "pose_updates_per_s": round(rate, 1), "ms_per_step": round(el / K * 1e3, 4),
            "kernel": name.value.decode(), "histogram_update_in_the_launch": bool(shape[3]),
            "frac_of_hbm_roofline": round(rate * cfg["alg"] / (HBM_PEAK_GBS * 1e9), 5),
            "workgroups_per_object": shape[1], "threads_per_workgroup": shape[2],
            "distinct_frame_rings": len({c.id for c in inst.color_cams})}


if __name__ == "__main__":
    main()
