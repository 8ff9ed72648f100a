#!/usr/bin/env python
"""bench.py — pose-updates/s of the M3T per-frame pose-optimisation hot path on MI355X.

One "step" = Tracker::ExecuteTrackingStep (7 correspondence iterations x 2 Newton updates +
histogram update, RBOT parameters, 200 lines) for every object of the batch on the next
synthetic 640x512 frame.  Workload at every N: 64 batched RBOT-geometry objects PER GPU
(BASELINE.json configs[1]); objects are independent, so ranks share nothing on the data
path (weak scaling, no collective).  Frames, models and histograms are resident in HBM
before the timed region starts (the reference's evaluators also exclude image loading,
rbot_evaluator.cpp:354-414).

Prints ONE JSON line on rank 0 (see DESIGN.md §6 for the roofline and cpu_baseline legs).
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
sys.path.insert(0, os.path.join(ROOT, "tests"))

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
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--objects", type=int, default=64, help="objects per GPU")
    p.add_argument("--models", type=int, default=8, help="distinct sparse viewpoint models per GPU")
    p.add_argument("--n-divides", type=int, default=4, help="geodesic subdivisions (4 -> 2562 views)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline sample budget (1 thread)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-cpu-parallel", action="store_true", help="skip the all-cores (OpenMP) CPU leg")
    p.add_argument("--repeats", type=int, default=5, help="how often the timed region of K steps is repeated")
    p.add_argument("--sweep", type=str, default="", help="comma separated object counts for a batch sweep (extra)")
    p.add_argument("--extras", action="store_true",
                   help="extra legs (never the headline): model generation without OpenGL and a tracking step with "
                        "all renderer-fed branches, on the reference's own test fixture")
    p.add_argument("--ycb", type=int, default=0,
                   help="extra leg: N objects with Region+Depth fused modalities, YCB parameters (BASELINE configs[2])")
    return p.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    n_gpus = world

    pkg = importlib.import_module("3dobjecttracking_amd")
    import scenes
    syn = pkg.synthetic
    hip = pkg.open_context(local_rank)

    n_obj, K, W = args.objects, args.steps, args.warmup
    n_frames = K + W + 1
    t0 = time.time()
    # weak scaling: rank r owns the global objects [r * n_obj, (r + 1) * n_obj)
    my_objects = pkg.sharding.shard_objects(n_obj * world, rank, world, mode="block")
    inputs = scenes.Inputs(n_obj, n_frames, n_divides=args.n_divides, n_models=min(args.models, n_obj),
                           first_object=int(my_objects[0]))
    inst = scenes.Instance(hip, inputs)
    for cam in inst.color_cams:
        hip.call("camera_set_ring", cam.id, n_frames)
    for i, cam in enumerate(inst.color_cams):
        for k in range(n_frames):
            f = inputs.color[i][k]
            hip.call("camera_upload_slot", cam.id, k, f.ctypes.data_as(C.c_void_p), f.strides[0])
    setup_s = time.time() - t0

    def barrier():
        hip.call("sync")
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(first, count):
        for k in range(first, first + count):
            hip.call("cameras_select_slot", k)
            hip.call("execute_tracking_step", k)

    hip.call("cameras_select_slot", 0)
    hip.call("start_modalities", 0)
    run(1, W)
    barrier()
    t = time.perf_counter()
    run(1 + W, K)
    barrier()
    elapsed = time.perf_counter() - t
    elapsed = pkg.sharding.max_over_ranks(elapsed, dist, device="cuda")
    poses = np.zeros((n_obj, 16), np.float32)
    hip.call("bodies_get_poses", poses.ctypes.data_as(C.POINTER(C.c_float)), n_obj)
    # the same K steps again, args.repeats - 1 more times (restarted from the ground-truth pose of frame W; the
    # trajectory checked below is the first one): min / median of the timed region, MAX over ranks each
    times = [elapsed]
    restart = np.stack([np.ascontiguousarray(inputs.gt[i][W].T, np.float32).reshape(16) for i in range(n_obj)])
    for _ in range(max(0, args.repeats - 1)):
        hip.call("bodies_set_poses", restart.ctypes.data_as(C.POINTER(C.c_float)), n_obj)
        barrier()
        t = time.perf_counter()
        run(1 + W, K)
        barrier()
        times.append(pkg.sharding.max_over_ranks(time.perf_counter() - t, dist, device="cuda"))
    elapsed = float(np.median(times))
    tracked = 0
    for i in range(n_obj):
        e = syn.pose_errors(poses[i].reshape(4, 4).T, inputs.gt[i][W + K])
        tracked += int(e[0] < np.deg2rad(5) and e[1] < 0.05)  # rbot_evaluator.cpp:416-433

    # ---- roofline leg: HIP events around the kernels on the context stream (rank 0) ----
    roofline = None
    if rank == 0:
        hip.call("bodies_set_poses", np.stack([np.ascontiguousarray(inputs.gt[i][W].T, np.float32).reshape(16)
                                               for i in range(n_obj)]).ctypes.data_as(C.POINTER(C.c_float)), n_obj)
        hip.call("set_kernel_timing", 1)
        run(1 + W, K)
        ms = (C.c_float * 2)()
        cnt = (C.c_int * 2)()
        hip.call("get_kernel_timing", ms, cnt)
        hip.call("set_kernel_timing", 0)
        shape = (C.c_int * 4)()
        hip.call("get_step_shape", shape)  # objects, workgroups per object, threads, histogram update fused
        kernel = "tracking_step_split_kernel" if shape[1] > 1 else "tracking_step_kernel"
        track_ms = ms[0] / max(cnt[0], 1)
        fused_hist = cnt[1] == 0  # the histogram update rode in the tracking launch (one workgroup per CU)
        hist_ms = ms[1] / max(cnt[1], 1)
        alg = B_ALG if fused_hist else B_ALG_TRACK_KERNEL
        achieved = alg * n_obj / (track_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(kernel, n_obj, fused_hist)
        roofline = {"bound": "hbm",
                    "kernel": kernel + (" (whole step incl. histogram update)" if fused_hist else ""),
                    "workgroups_per_object": shape[1], "threads_per_workgroup": shape[2],
                    "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                    "kernel_ms": round(track_ms, 4), "algorithmic_bytes_per_launch": alg * n_obj}
        if not fused_hist:
            roofline["histogram_kernel_ms"] = round(hist_ms, 4)
            roofline["histogram_kernel_GBs"] = round((B_HIST_RMW + B_HIST_PIXELS + B_VIEW_SCAN // 7 + B_VIEW_POINTS // 7) *
                                                     n_obj / (hist_ms * 1e-3) / 1e9, 2)

    # ---- host-buffer (PCIe-inclusive) rate: every step first uploads its 64 frames from host memory
    # through m3t_hip_camera_upload (the boundary's Camera::UpdateImage); never the headline value ----
    pcie = None
    if rank == 0 and world == 1:
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
                "note": "pageable host frames, synchronous m3t_hip_camera_upload per camera, then the step"}
        # the same with page-locked frames and the double-buffered asynchronous ingest: frame k+1 crosses
        # PCIe on the copy stream while step k runs (m3t_hip_camera_upload_slot_async)
        blocks = [np.stack([inputs.color[i][k] for i in range(n_obj)]) for k in range(1 + W, 2 + W + n_up)]
        for b in blocks:
            inst.tracker.register_host_buffer(b)
        for i, cam in enumerate(inst.color_cams):
            cam.upload_slot(0, blocks[0][i], asynchronous=True)
        hip.call("ingest_sync")
        hip.call("sync")
        tu = time.perf_counter()
        for j in range(n_up):
            hip.call("cameras_select_slot", j % 2)
            hip.call("execute_tracking_step", 1 + W + j)
            for i, cam in enumerate(inst.color_cams):
                cam.upload_slot((j + 1) % 2, blocks[j + 1][i], asynchronous=True)
        hip.call("ingest_sync")
        hip.call("sync")
        el = time.perf_counter() - tu
        for b in blocks:
            inst.tracker.unregister_host_buffer(b)
        pcie["async_pinned"] = {"pose_updates_per_s": round(n_obj * n_up / el, 1),
                                "ms_per_step": round(el / n_up * 1e3, 3),
                                "upload_GBs": round(frame_bytes * n_up / el / 1e9, 2),
                                "note": "page-locked frames, m3t_hip_camera_upload_slot_async on the copy stream, "
                                        "two ring slots; the copy of frame k+1 overlaps step k"}

    # ---- optional batch sweep (extra lines on stderr, not the headline) ----
    sweep = []
    if rank == 0 and args.sweep:
        for n in [int(x) for x in args.sweep.split(",") if x]:
            sweep.append(batch_point(pkg, scenes, n, args))

    ycb = None
    if rank == 0 and args.ycb:
        ycb = ycb_point(pkg, scenes, args.ycb, args)

    # ---- CPU baseline: the oracle restatement, bounded sample (rank 0); its first pass over the frames is also
    # the parity check of the benchmarked trajectory: the same 8 objects, the same frames, free running ----
    cpu, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # N = 1 only (the other ranks would idle)
        import util
        ora = util.open_oracle()
        n_cpu = min(8, n_obj)
        sub = scenes.Inputs.__new__(scenes.Inputs)
        sub.__dict__.update(inputs.__dict__)
        sub.n_objects = n_cpu
        oinst = scenes.Instance(ora, sub)
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
        cpu_parallel = None
        if not args.no_cpu_parallel:
            cpu_parallel = cpu_all_cores(ora, scenes, inputs, n_obj, n_frames)
        cpu = {"value": round(done / spent, 1), "unit": "pose-updates/s", "cores": 1, "kind": "port",
               "all_cores": cpu_parallel,
               "sample": "%d pose-updates of %d of the same objects, same frames, oracle/libm3t_oracle.so "
                         "(g++ -O3 -march=x86-64-v3), 1 thread, host has %d cores" % (done, n_cpu, os.cpu_count())}

    if rank == 0:
        total = n_obj * n_gpus * K
        out = {
            "metric": "pose-updates/sec (64 objects, 200 lines, 7 it)", "value": round(total / elapsed, 1),
            "unit": "pose-updates/s", "n_gpus": n_gpus, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d batched RBOT-geometry objects per GPU, RegionModality only, "
                                   "200 lines x 7 corr-iterations x 2 updates, 640x512 BGR8, 32-bin histograms, "
                                   "%d views x 200 points models (%d distinct)" %
                                   (n_obj, inputs.region_models[0][1].shape[0], len(inputs.region_models)),
                       "objects_per_gpu": n_obj, "parallelism": "objects sharded over %d GPU(s), no collective" % n_gpus,
                       "tracked_within_5cm_5deg": "%d/%d" % (tracked, n_obj), "setup_s": round(setup_s, 1)},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
            "repeats": {"n": len(times), "ms_per_step_min": round(min(times) / K * 1e3, 4),
                        "ms_per_step_median": round(elapsed / K * 1e3, 4),
                        "ms_per_step_all": [round(x / K * 1e3, 4) for x in times]},
            "pcie_inclusive": pcie,
            "frac_of_hbm_roofline_whole_step": round(total / elapsed * B_ALG / (HBM_PEAK_GBS * 1e9 * n_gpus), 5),
            "newton_steps_per_s": round(total / elapsed * 14, 1),  # 7 correspondence iterations x 2 updates
        }
        if sweep:
            out["batch_sweep"] = sweep
        if ycb:
            out["ycb_region_depth"] = ycb
        if args.extras:
            out["extras"] = extras_point(pkg)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_all_cores(ora_unused, scenes, inputs, n_obj, n_frames, seconds=8.0):
    """SURVEY 8(d) CPU baseline (ii): the whole batch in ONE oracle context, stepped with an OpenMP `parallel for`
    over the objects at nproc threads (m3t_oracle_execute_tracking_step_parallel; what the reference's evaluators
    do over sequences, rbot_evaluator.cpp:144).  Also the evaluators' four time buckets
    (rbot_evaluator.cpp:354-414), summed over threads."""
    import util
    ora = util.open_oracle()
    f = ora.lib.m3t_oracle_execute_tracking_step_parallel
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    inst = scenes.Instance(ora, inputs)
    inst.upload_frame(0)
    inst.tracker.StartModalities(0)
    n_threads = os.cpu_count() or 1
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
    return {"value": round(done / spent, 1), "unit": "pose-updates/s", "cores": n_threads,
            "sample": "%d pose-updates, all %d objects in one oracle context, OpenMP parallel for over objects, "
                      "%d threads" % (done, n_obj, n_threads),
            "bucket_share": {"correspondences": round(buckets[0] / tot, 3), "gradient_hessian": round(buckets[1] / tot, 3),
                             "optimization": round(buckets[2] / tot, 3), "results": round(buckets[3] / tot, 3)},
            "thread_seconds_per_pose_update_us": round(tot / done * 1e6, 1)}


def measured_traffic(kernel, n_obj, fused_histogram=False):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/rNN_hbm_traffic.json: FETCH_SIZE and WRITE_SIZE in separate runs of this same
    command, read side doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950).  The counters
    cannot be collected from inside this process; null when no profile for this batch size exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        if d.get("objects_per_launch") != n_obj or bool(d.get("histogram_update_fused", False)) != fused_histogram:
            return None, None
        return d["kernels"][kernel]["hbm_bytes_per_launch_corrected"], os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def extras_point(pkg):
    """Rows f-1 / a14 on the reference's own fixture (tests/golden): wall time of generating the default
    region + depth model of the triangle body (2 x 2562 views at 2000 x 2000) and of one tracking step of
    Region + Depth modality with region checking, silhouette checking and modelled occlusions behind the
    20 950-triangle bottle (4 focused renderers, refreshed before each of the 7 correspondence searches)."""
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
    return {"region_model_generation_s": round(t1 - t0, 2), "depth_model_generation_s": round(t2 - t1, 2),
            "model": "2562 views x 200 points, 2000 x 2000 renderings, data/_body/triangle.obj",
            "renderer_fed_tracking_step_ms": round((t4 - t3) / n * 1e3, 3),
            "renderer_fed_config": "Region + Depth, region / silhouette checking + modelled occlusions, 4 focused "
                                   "renderers of 20 958 triangles at 200 x 200, 7 x 2 iterations, 1 object"}


B_ALG_YCB = 1094456  # Region + Depth with measured occlusions, YCB parameters


def ycb_point(pkg, scenes, n_obj, args):
    """BASELINE configs[2]-shaped leg: n_obj objects, Region + Depth (ICG), YCB parameters, 640x480"""
    import util
    hip = pkg.open_context(0)
    K, W = 10, 3
    n_frames = K + W + 1
    inputs = scenes.Inputs(n_obj, n_frames, n_divides=args.n_divides, n_models=min(8, n_obj), with_depth=True)
    inst = scenes.Instance(hip, inputs, use_depth=True)
    for cams, frames in ((inst.color_cams, inputs.color), (inst.depth_cams, inputs.depth)):
        for i, cam in enumerate(cams):
            hip.call("camera_set_ring", cam.id, n_frames)
            for k in range(n_frames):
                f = frames[i][k]
                hip.call("camera_upload_slot", cam.id, k, f.ctypes.data_as(C.c_void_p), f.strides[0])
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
    poses = np.zeros((n_obj, 16), np.float32)
    hip.call("bodies_get_poses", poses.ctypes.data_as(C.POINTER(C.c_float)), n_obj)
    adds = [pkg.synthetic.add_s(inputs.vertices[i], poses[i].reshape(4, 4).T, inputs.gt[i][W + K]) for i in range(n_obj)]
    # CPU restatement on the same objects / frames (1 thread)
    ora = util.open_oracle()
    oinst = scenes.Instance(ora, inputs, use_depth=True)
    oinst.upload_frame(0)
    oinst.tracker.StartModalities(0)
    tc = 0.0
    for k in range(1, 1 + W + K):
        oinst.upload_frame(k)
        t0 = time.perf_counter()
        oinst.tracker.ExecuteTrackingStep(k)
        tc += time.perf_counter() - t0
    rate = n_obj * K / el
    return {"objects": n_obj, "pose_updates_per_s": round(rate, 1), "ms_per_step": round(el / K * 1e3, 4),
            "frac_of_hbm_roofline": round(rate * B_ALG_YCB / (HBM_PEAK_GBS * 1e9), 5),
            "mean_add_s_vs_gt_m": round(float(np.mean(adds)), 5),
            "cpu_port_pose_updates_per_s": round(n_obj * (W + K) / tc, 1)}


def batch_point(pkg, scenes, n_obj, args):
    """pose-updates/s at another batch size (few frames, models shared)"""
    hip = pkg.open_context(0)
    K, W = 6, 2
    n_frames = K + W + 1
    inputs = scenes.Inputs(min(n_obj, 64), n_frames, n_divides=args.n_divides, n_models=min(8, n_obj))
    # replicate the 64 rendered streams to reach n_obj objects
    rep = scenes.Inputs.__new__(scenes.Inputs)
    rep.__dict__.update(inputs.__dict__)
    idx = [i % inputs.n_objects for i in range(n_obj)]
    rep.n_objects = n_obj
    for name in ("scenes", "model_of", "gt", "color", "depth", "start", "vertices"):
        rep.__dict__[name] = [inputs.__dict__[name][i] for i in idx]
    if n_obj > 4096:  # beyond that the replicas look at the 64 frame streams through 64 shared cameras
        rep.camera_of = idx
    inst = scenes.Instance(hip, rep)
    staged = set()
    for i, cam in enumerate(inst.color_cams):
        if cam.id in staged:
            continue
        staged.add(cam.id)
        hip.call("camera_set_ring", cam.id, n_frames)
        for k in range(n_frames):
            f = rep.color[i][k]
            hip.call("camera_upload_slot", cam.id, k, f.ctypes.data_as(C.c_void_p), f.strides[0])
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
    return {"objects": n_obj, "pose_updates_per_s": round(rate, 1), "ms_per_step": round(el / K * 1e3, 4),
            "frac_of_hbm_roofline": round(rate * B_ALG / (HBM_PEAK_GBS * 1e9), 5)}


if __name__ == "__main__":
    main()
