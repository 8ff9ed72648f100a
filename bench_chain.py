"""bench.py --config chain8 (BASELINE configs[4]): one kinematic structure -- a chain of 8 bodies, a free root and seven
1-dof revolute joints (13 dof) -- tracked by one RegionModality per body; with N GPUs every rank holds the whole link
tree but only the modalities of its own bodies (body i -> rank i mod N), and each Newton step sums the stacked link
sums (6 + 36 floats per link) with ONE ncclAllReduce inside the library (m3t_hip_comm_init_rank: while a communicator
is set, ExecuteTrackingStep runs link sums -> all-reduce -> project + solve by itself; the sum is exact, N ranks
compute the poses of one process bit for bit).  Also the sweep of
examples/optimization_time.cpp:14-79: Optimizer::CalculateOptimization for chains of 1..50 one-dof links."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
B_ALG = 1386704  # bytes per pose-update of a RegionModality with RBOT parameters


def chain_inputs(scenes, syn, n_bodies, n_frames, n_divides):
    """bodies, models and rendered frames of a serial chain: body 0 free, body j rotates about its joint's z axis"""
    inputs = scenes.Inputs(n_bodies, 1, n_divides=n_divides, n_models=min(n_bodies, 4))
    rng = np.random.default_rng(11)
    joints = [syn.make_pose(syn.rot_vec(rng.uniform(-0.3, 0.3, 3)), [0.02 * (1 + j % 3), 0.01 * (j % 2), 0.0])
              for j in range(n_bodies - 1)]
    pose_root = inputs.gt[0][0].copy()
    angles = rng.uniform(-0.2, 0.2, n_bodies - 1)
    gt = []
    inputs.color = [[] for _ in range(n_bodies)]
    for k in range(n_frames):
        pose_root = syn.perturb_pose(pose_root, rng, rot_deg=0.7, trans=0.002)
        angles = angles + rng.uniform(-0.03, 0.03, n_bodies - 1)
        poses = [pose_root.copy()]
        for j in range(n_bodies - 1):
            poses.append(poses[-1] @ joints[j] @ syn.make_pose(syn.rot_vec([0, 0, angles[j]]), [0, 0, 0]))
        gt.append(([p.copy() for p in poses], angles.copy()))
        for i in range(n_bodies):
            inputs.color[i].append(inputs.scenes[i].render(poses[i]))
    return inputs, joints, gt


class Chain:
    def __init__(self, api, host, syn, inputs, joints, start_root, start_angles, owned):
        rp = dict(syn.RBOT_REGION_PARAMS, measure_occlusions=0)
        n = inputs.n_objects
        self.api = api
        self.models = [host.RegionModel(api, data_points=m[0], orientations=m[1], contour_lengths=m[2])
                       for m in inputs.region_models]
        self.bodies = [host.Body(api, np.eye(4)) for _ in range(n)]
        self.cams = [host.ColorCamera(api, **inputs.intr) for _ in range(n)]
        self.mods = {i: host.RegionModality(api, self.bodies[i], self.cams[i], self.models[inputs.model_of[i]], **rp)
                     for i in owned}
        self.links = [host.Link(api, body=self.bodies[0])]
        for j in range(1, n):
            self.links.append(host.Link(
                api, body=self.bodies[j], parent=self.links[j - 1],
                joint2parent_pose=joints[j - 1] @ syn.make_pose(syn.rot_vec([0, 0, start_angles[j - 1]]), [0, 0, 0]),
                free_directions=(0, 0, 1, 0, 0, 0)))
        for i in owned:
            self.links[i].AddModality(self.mods[i])
        self.opt = host.Optimizer(api, root_link=self.links[0])
        self.tracker = host.Tracker(api, 7, 2)
        self.bodies[0].set_body2world_pose(start_root)
        assert self.tracker.CalculateConsistentPoses()
        self.owned = list(owned)

    def upload(self, inputs, k):
        for i in self.owned:
            self.cams[i].UpdateImage(inputs.color[i][k])

    def poses(self):
        return np.stack([b.body2world_pose() for b in self.bodies])


def optimization_time_structures(api, host, n_bodies, n_structures):
    """examples/optimization_time.cpp:23-57 (test_constraints = false): a body-less root without free directions and
    n_bodies chained links with one free direction each, joint2parent = translate(0.01, 0, 0); no modalities"""
    for _ in range(n_structures):
        root = host.Link(api, free_directions=(0, 0, 0, 0, 0, 0))
        parent = root
        for j in range(n_bodies):
            t = np.eye(4)
            t[0, 3] = 0.01
            parent = host.Link(api, parent=parent, joint2parent_pose=t, free_directions=(1, 0, 0, 0, 0, 0))
        host.Optimizer(api, root_link=root)
    return host.Tracker(api, 1, 1)


def rank_share_chain(args, pkg, scenes, syn, host, inputs, joints, start_root, start_angles, n_bodies, local_rank,
                     K=10, W=3, regions=7):
    """`--rank-share`: what ONE rank of N pays per tracking step before the transport is added -- the whole link tree
    but only the modalities of the bodies i mod N == 0 (sharding.place_bodies), through the distributed path (library
    communicator at world size 1: link sums -> ncclAllReduce -> project + solve per Newton step).  The other ranks'
    link sums are missing, so the poses are not the chain's: a timing of the code path, never a result."""
    counts = [int(x) for x in args.rank_share.split(",") if x]
    n_frames = K + W + 1
    points = []
    for n in counts:
        owned = [i for i, r in enumerate(pkg.sharding.place_bodies(n_bodies, n)) if r == 0]
        ctx = pkg.open_context(local_rank)
        ch = Chain(ctx, host, syn, inputs, joints, start_root, start_angles, owned)
        buf = C.create_string_buffer(128)
        ctx.call("comm_get_unique_id", buf, 128)
        ctx.call("comm_init_rank", buf, 128, 1, 0)
        for i in owned:
            ctx.call("camera_set_ring", ch.cams[i].id, n_frames)
            for k in range(n_frames):
                f = inputs.color[i][k]
                ctx.call("camera_upload_slot", ch.cams[i].id, k, f.ctypes.data_as(C.c_void_p), f.strides[0])

        def steps(first, count):
            for k in range(first, first + count):
                ctx.call("cameras_select_slot", k)
                ctx.call("execute_tracking_step", k)

        ctx.call("cameras_select_slot", 0)
        ctx.call("start_modalities", 0)
        steps(1, W)
        ctx.call("sync")
        times = []
        for _ in range(regions):
            t = time.perf_counter()
            steps(1 + W, K)
            ctx.call("sync")
            times.append(time.perf_counter() - t)
        el = float(np.median(times))
        calls = C.c_longlong(0)
        ctx.call("comm_get_allreduce_count", C.byref(calls))
        ranks = C.c_int(0)
        ctx.call("comm_get_rank_count", C.byref(ranks))
        points.append({"n_gpus": n, "bodies_with_modalities_on_rank_0": len(owned), "rank_0_ms_per_step": round(el / K * 1e3, 4),
                       "projected_pose_updates_per_s_before_transport": round(n_bodies * K / el, 1),
                       "allreduce_calls_per_step": round(calls.value / (W + regions * K), 2),
                       "communicator_ranks_in_this_measurement": int(ranks.value)})
        ctx.call("comm_destroy")
        del ch, ctx
    return {"what": "PROJECTION, not a measurement of N GPUs: the distributed path of rank 0 (modalities of the bodies "
                    "i mod N == 0 only, whole link tree) timed alone on ONE GPU with the library's communicator at world "
                    "size 1; the all-reduce's transport latency over xGMI (14 per step) is NOT in it",
            "steps": K, "warmup": W, "regions": regions, "points": points,
            "thread_ranks": thread_ranks_chain(pkg, syn, host, inputs, joints, start_root, start_angles, n_bodies, local_rank,
                                               [n for n in counts if n > 1], K, W)}


def thread_ranks_chain(pkg, syn, host, inputs, joints, start_root, start_angles, n_bodies, device, counts, K, W):
    """The N-rank step with REAL partial sums on one GPU: N contexts, one host thread each, body i's modality in context
    i mod N, the link sums added through m3t_hip_comm_set_reduce_callback (sharding.ThreadRanks).  Every context takes
    tracking_step_tree_segment_kernel with its own bodies only and must end on the poses of one context that owns the
    whole chain (the one-launch step), bit for bit.  The time is N contexts sharing ONE GPU plus a host-side sum per
    Newton step (stream sync, two copies, two thread barriers): an upper bound of the code path, not of N GPUs."""
    n_frames = K + W + 1

    def stage(ctx, ch):
        for i in ch.owned:
            ctx.call("camera_set_ring", ch.cams[i].id, n_frames)
            for k in range(n_frames):
                f = inputs.color[i][k]
                ctx.call("camera_upload_slot", ch.cams[i].id, k, f.ctypes.data_as(C.c_void_p), f.strides[0])

    def steps(ctx, first, count):
        for k in range(first, first + count):
            ctx.call("cameras_select_slot", k)
            ctx.call("execute_tracking_step", k)

    one = pkg.open_context(device)
    ch1 = Chain(one, host, syn, inputs, joints, start_root, start_angles, range(n_bodies))
    stage(one, ch1)
    one.call("cameras_select_slot", 0)
    one.call("start_modalities", 0)
    steps(one, 1, W + K)
    expected = ch1.poses()
    del ch1, one
    out = []
    for n in counts:
        placed = pkg.sharding.place_bodies(n_bodies, n)
        ctxs = [pkg.open_context(device) for _ in range(n)]
        chains = [Chain(ctxs[r], host, syn, inputs, joints, start_root, start_angles, [i for i, p in enumerate(placed) if p == r])
                  for r in range(n)]
        ranks = pkg.sharding.ThreadRanks(ctxs)
        elapsed = [0.0] * n

        def work(rank, ctx):
            stage(ctx, chains[rank])
            ctx.call("cameras_select_slot", 0)
            ctx.call("start_modalities", 0)
            steps(ctx, 1, W)
            ctx.call("sync")
            t = time.perf_counter()
            steps(ctx, 1 + W, K)
            ctx.call("sync")
            elapsed[rank] = time.perf_counter() - t
            return chains[rank].poses()

        poses = ranks.run(work)
        kernels, calls = [], []
        for ctx in ctxs:
            name, c = C.create_string_buffer(96), C.c_longlong(0)
            ctx.call("get_step_kernel", name, 96)
            ctx.call("comm_get_allreduce_count", C.byref(c))
            kernels.append(name.value.decode())
            calls.append(round(c.value / (W + K), 2))
        ranks.close()
        out.append({"n_ranks": n, "bodies_per_rank": [len(ch.owned) for ch in chains],
                    "kernel": sorted(set(kernels)), "reductions_per_step": sorted(set(calls)),
                    "bit_identical_to_one_context_owning_the_chain": bool(all(np.array_equal(p, expected) for p in poses)),
                    "ms_per_step_all_ranks_sharing_one_gpu_host_summed": round(max(elapsed) / K * 1e3, 4)})
        del chains, ctxs
    return out


def run(args, pkg, rank, local_rank, world, dist, torch, open_oracle, measured_traffic=None, dry_run=False):
    import bench_inputs as scenes
    syn, host = pkg.synthetic, pkg.host
    n_bodies, K, W = args.objects or 8, args.steps, args.warmup
    n_frames = K + W + 1
    n_div = min(args.n_divides, 2)  # the link kernels, not the model size, are the subject of this leg
    t0 = time.time()
    inputs, joints, gt = chain_inputs(scenes, syn, n_bodies, n_frames, n_div)
    start_root = syn.perturb_pose(gt[0][0][0], np.random.default_rng(5), rot_deg=0.5, trans=0.001)
    start_angles = gt[0][1] + 0.01
    hip = pkg.open_context(local_rank)
    owned = [i for i, r in enumerate(pkg.sharding.place_bodies(n_bodies, world)) if r == rank]
    ch = Chain(hip, host, syn, inputs, joints, start_root, start_angles, owned)
    host_reduce = None
    if world > 1 and dry_run:  # all ranks on one GPU: the collective through the reduce seam, gloo as the transport
        host_reduce = pkg.sharding.HostReduce(hip, dist)
    elif world > 1:  # the library's own RCCL communicator: rank 0 creates the id, torch.distributed carries it
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            buf = C.create_string_buffer(128)
            hip.call("comm_get_unique_id", buf, 128)
            uid = torch.tensor(list(buf.raw), dtype=torch.uint8, device="cuda")
        dist.broadcast(uid, 0)
        raw = bytes(uid.cpu().tolist())
        hip.call("comm_init_rank", C.create_string_buffer(raw, 128), 128, world, rank)
    for cam_i in owned:
        hip.call("camera_set_ring", ch.cams[cam_i].id, n_frames)
        for k in range(n_frames):
            f = inputs.color[cam_i][k]
            hip.call("camera_upload_slot", ch.cams[cam_i].id, k, f.ctypes.data_as(C.c_void_p), f.strides[0])
    setup_s = time.time() - t0

    def barrier():
        hip.call("sync")
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(first, count):
        for k in range(first, first + count):
            hip.call("cameras_select_slot", k)
            hip.call("execute_tracking_step", k)

    hip.call("cameras_select_slot", 0)
    hip.call("start_modalities", 0)
    run_steps(1, W)
    barrier()
    times = []
    first_poses = None
    for r in range(max(1, args.repeats)):
        t = time.perf_counter()
        run_steps(1 + W, K)
        barrier()
        times.append(pkg.sharding.max_over_ranks(time.perf_counter() - t, dist, device="cuda"))
        if first_poses is None:
            first_poses = ch.poses()
    elapsed = float(np.median(times))
    # roofline leg: ONE HIP event pair on the context's stream around the K launches of one more region.  EVERY rank
    # runs these steps: at N > 1 each of them reduces 14 times per step, and a rank that had already left for the final
    # barrier would leave rank 0 waiting in its first all-reduce for ever (found by the two-rank dry run of round 6)
    ms, cnt = (C.c_float * 2)(), (C.c_int * 2)()
    hip.call("set_kernel_timing", 2)
    run_steps(1 + W, K)
    hip.call("get_kernel_timing", ms, cnt)
    hip.call("set_kernel_timing", 0)
    barrier()
    if rank != 0:
        return None
    kernel_ms = float(ms[0]) / max(int(cnt[0]), 1) if cnt[0] else None
    # ---- CPU restatement of the same chain (all bodies in one process) + parity of the first timed trajectory ----
    cpu, parity = None, None
    if not args.no_cpu_baseline:
        ora = open_oracle()
        oc = Chain(ora, host, syn, inputs, joints, start_root, start_angles, range(n_bodies))
        oc.upload(inputs, 0)
        assert oc.tracker.StartModalities(0)
        spent = 0.0
        for k in range(1, 1 + W + K):
            oc.upload(inputs, k)
            tc = time.perf_counter()
            assert oc.tracker.ExecuteTrackingStep(k)
            spent += time.perf_counter() - tc
        op = oc.poses()
        errs = [syn.pose_errors(first_poses[i], op[i]) for i in range(n_bodies)]
        parity = {"rot_max": float(max(e[0] for e in errs)), "trans_max": float(max(e[1] for e in errs)),
                  "add_s_max": float(max(syn.add_s(inputs.vertices[i], first_poses[i], op[i]) for i in range(n_bodies))),
                  "n": n_bodies, "frames": W + K, "bit_identical": bool(np.array_equal(first_poses, op)),
                  "what": "body2world of every body after %d free-running frames, HIP vs oracle" % (W + K)}
        cpu = {"value": round(n_bodies * (W + K) / spent, 1), "unit": "pose-updates/s", "cores": 1, "kind": "port",
               "sample": "%d tracking steps of the same %d-body chain, oracle/libm3t_oracle.so, 1 thread" % (W + K, n_bodies)}
    gt_err = [syn.pose_errors(first_poses[i], gt[W + K][0][i]) for i in range(n_bodies)]
    # ---- the path a rank runs when the structure spans GPUs, at world size 1: per Newton step link sums -> the
    # library's ncclAllReduce -> project + solve, one launch per sub-step (what N > 1 costs per rank before the
    # transport's latency is added); the poses must be the fused launch's
    distributed = None
    if world == 1:
        try:
            ctx2 = pkg.open_context(local_rank)
            ch2 = Chain(ctx2, host, syn, inputs, joints, start_root, start_angles, range(n_bodies))
            buf = C.create_string_buffer(128)
            ctx2.call("comm_get_unique_id", buf, 128)
            ctx2.call("comm_init_rank", buf, 128, 1, 0)
            for cam in ch2.cams:
                ctx2.call("camera_set_ring", cam.id, n_frames)
                for k in range(n_frames):
                    f = inputs.color[ch2.cams.index(cam)][k]
                    ctx2.call("camera_upload_slot", cam.id, k, f.ctypes.data_as(C.c_void_p), f.strides[0])
            ctx2.call("cameras_select_slot", 0)
            ctx2.call("start_modalities", 0)

            def steps2(first, count):
                for k in range(first, first + count):
                    ctx2.call("cameras_select_slot", k)
                    ctx2.call("execute_tracking_step", k)
            steps2(1, W)
            ctx2.call("sync")
            t = time.perf_counter()
            steps2(1 + W, K)
            ctx2.call("sync")
            dt = time.perf_counter() - t
            calls = C.c_longlong(0)
            ctx2.call("comm_get_allreduce_count", C.byref(calls))
            distributed = {"ms_per_step": round(dt / K * 1e3, 4), "allreduce_calls_per_step": round(calls.value / (W + K), 2),
                           "floats_per_allreduce": 42 * n_bodies,
                           "bit_identical_to_the_one_launch_step": bool(np.array_equal(ch2.poses(), first_poses)),
                           "kernel": (lambda b: (ctx2.call("get_step_kernel", b, 64), b.value.decode())[1])(C.create_string_buffer(64)),
                           "note": "library communicator at world size 1: one launch of tracking_step_tree_segment_kernel "
                                   "(solve from the summed link sums | search or line state | products, link sums) and one "
                                   "ncclAllReduce per Newton step"}
            ctx2.call("comm_destroy")
        except Exception as e:  # (no RCCL on the box: the leg is reported as missing, the bench line stands)
            distributed = {"error": str(e)[:200]}
    # ---- examples/optimization_time.cpp: CalculateOptimization for chains of 1..50 one-dof links ----
    sweep = []
    n_structures = 256
    for nb in (1, 2, 4, 8, 16, 32, 50):
        api = pkg.open_context(local_rank)
        tracker = optimization_time_structures(api, host, nb, n_structures)
        assert tracker.CalculateOptimization(0, 0, 0)
        api.call("sync")
        reps = 20
        t = time.perf_counter()
        for _ in range(reps):
            api.call("calculate_optimization", 0, 0, 0)
        api.call("sync")
        gpu_us = (time.perf_counter() - t) / reps * 1e6
        ora = open_oracle()
        otr = optimization_time_structures(ora, host, nb, 1)
        assert otr.CalculateOptimization(0, 0, 0)
        t = time.perf_counter()
        for _ in range(200):
            ora.call("calculate_optimization", 0, 0, 0)
        cpu_us = (time.perf_counter() - t) / 200 * 1e6
        sweep.append({"bodies": nb, "gpu_us_per_launch_of_%d_structures" % n_structures: round(gpu_us, 1),
                      "gpu_us_per_structure": round(gpu_us / n_structures, 3), "cpu_port_us_per_structure": round(cpu_us, 2)})
    total = n_bodies * K
    rate = total / elapsed
    import bench
    rccl_ranks = bench.live_rccl_ranks(dist, hip)  # ncclCommCount of the library's communicator: 0 at N = 1 (none set)
    projected = rank_share_chain(args, pkg, scenes, syn, host, inputs, joints, start_root, start_angles, n_bodies, local_rank) \
        if (getattr(args, "rank_share", "") and world == 1) else None
    collectives = C.c_longlong(0)  # ncclAllReduce calls this rank's context issued (start to here: W + repeats x K steps)
    hip.call("comm_get_allreduce_count", C.byref(collectives))
    steps_run = W + max(1, args.repeats) * K + K  # (+ the roofline leg's region)
    name = C.create_string_buffer(64)
    hip.call("get_step_kernel", name, 64)
    kernel = name.value.decode()
    shape = (C.c_int * 4)()
    hip.call("get_step_shape", shape)
    fused = kernel in ("tracking_step_tree_kernel", "tracking_step_tree_split_kernel")
    traffic, traffic_src = (measured_traffic("chain8", kernel, n_bodies, True) if (measured_traffic and fused) else (None, None))
    return {
        "metric": "pose-updates/sec (Mb-ICG kinematic chain, %d bodies, %d dof)" % (n_bodies, 6 + n_bodies - 1),
        "value": round(rate, 1), "unit": "pose-updates/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[4]: one kinematic chain of %d bodies (free root + %d revolute joints), one "
                               "RegionModality per body (RBOT parameters, 200 lines x 7 x 2); %s; body i on GPU i mod %d, "
                               "ONE ncclAllReduce of %d floats (the link sums) per Newton step%s" %
                               (n_bodies, n_bodies - 1,
                                "one launch per frame, %d workgroup(s) of %d threads per body, the structure solved by every "
                                "workgroup's first wave" % (shape[1], shape[2]) if fused else
                                ("one launch of tracking_step_tree_segment_kernel and one reduction of the link sums per Newton "
                                 "step" if "segment" in kernel else "one launch per sub-step, link kernels: one wave per structure"),
                                world, 42 * n_bodies, "" if world > 1 else " when N > 1"),
                   "bodies": n_bodies, "parallelism": "bodies sharded over %d GPU(s)" % world,
                   "rccl_ranks": rccl_ranks, "step_kernel": kernel,
                   "allreduce_calls_per_step": round(collectives.value / steps_run, 3),
                   "max_rotation_error_vs_ground_truth_rad": round(float(max(e[0] for e in gt_err)), 5),
                   "setup_s": round(setup_s, 1)},
        "roofline": {"bound": "hbm",
                     "kernel": kernel if fused else "whole step: 7 x (correspondence + 2 x (g/H, project, solve)) + results launches",
                     "achieved": round((n_bodies * B_ALG / (kernel_ms * 1e-3) / 1e9) if (fused and kernel_ms) else rate * B_ALG / 1e9, 3),
                     "peak": 8000.0, "unit": "GB/s",
                     "frac": round(((n_bodies * B_ALG / (kernel_ms * 1e-3)) if (fused and kernel_ms) else rate * B_ALG) / 8e12, 6),
                     "kernel_ms": round(kernel_ms, 4) if (fused and kernel_ms) else None,
                     "algorithmic_bytes_per_launch": n_bodies * B_ALG,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "note": ("latency-bound by construction: 14 dependent solves of one 13-dof structure per frame, each on "
                              "one wave" if fused else
                              "launch- and latency-bound by construction: 50 dependent launches per frame for one structure")},
        "cpu_baseline": cpu, "parity": parity,
        "repeats": {"n": len(times), "ms_per_step_min": round(min(times) / K * 1e3, 4),
                    "ms_per_step_median": round(elapsed / K * 1e3, 4)},
        "optimization_time_sweep": sweep,
        "distributed_path_world1": distributed,
        "projected_scaling": projected,
    }
