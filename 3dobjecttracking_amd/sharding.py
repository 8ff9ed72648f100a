"""Multi-GPU layout of the hot path: independent rigid objects are sharded over ranks (one
process per GPU); no collective is on the data path.  Only the final poses are
gathered, and bench.py reduces its wall-clock with MAX.  torch.distributed is plumbing here:
backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
import numpy as np


def shard_objects(n_objects, rank, world_size, mode="round_robin"):
    """Global object ids owned by `rank`.  round_robin: object i -> GPU i mod G (strong scaling,
    fixed total); block: contiguous ranges (bench.py weak scaling: rank r owns [r*n, (r+1)*n))."""
    ids = np.arange(n_objects)
    if mode == "round_robin":
        return ids[ids % world_size == rank]
    per = (n_objects + world_size - 1) // world_size
    return ids[rank * per:(rank + 1) * per]


def gather_poses(local_ids, local_poses, n_objects, dist=None):
    """All ranks end up with poses[n_objects, 4, 4] in global object order."""
    local_poses = np.asarray(local_poses, np.float32).reshape(-1, 4, 4)
    out = np.zeros((n_objects, 4, 4), np.float32)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        out[np.asarray(local_ids)] = local_poses
        return out
    import torch
    world = dist.get_world_size()
    payload = (np.asarray(local_ids, np.int64), local_poses)
    gathered = [None] * world
    dist.all_gather_object(gathered, payload)
    for ids, poses in gathered:
        out[ids] = poses
    return out


def max_over_ranks(value, dist=None, device=None):
    """bench.py timing contract: the slowest rank defines the step time."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def place_bodies(n_bodies, world_size, shared_histograms=()):
    """Rank of every body of ONE kinematic structure spread over `world_size` GPUs: bodies are placed
    round-robin, except that bodies whose RegionModalities share a ColorHistograms object
    (RegionModality::UseSharedColorHistograms, region_modality.h:200-201; evaluate_rtb_dataset.cpp:75) stay
    together on the rank of the group's first body -- the shared count table is then summed inside one context and
    the structure keeps its single all-reduce per Newton step (no second collective for the histograms).
    shared_histograms: iterable of body-index groups.  Returns a list: body -> rank."""
    rank_of = [None] * n_bodies
    group_of = {}
    for g, bodies in enumerate(shared_histograms):
        for b in bodies:
            if b in group_of:
                raise ValueError("body %d is in two shared-histogram groups" % b)
            group_of[b] = g
    group_rank, nxt = {}, 0
    for b in range(n_bodies):
        g = group_of.get(b)
        if g is not None and g in group_rank:
            rank_of[b] = group_rank[g]
            continue
        rank_of[b] = nxt % world_size
        nxt += 1
        if g is not None:
            group_rank[g] = rank_of[b]
    return rank_of


class ThreadRanks:
    """N contexts of ONE process playing the ranks of a kinematic structure spread over GPUs, one host
    thread per context, the collective supplied through m3t_hip_comm_set_reduce_callback: where a rank would call
    ncclAllReduce, its thread synchronises its stream, copies its link sums to the host, meets the others at a barrier,
    adds all of them in `order` (any order: every link's modalities live on one rank, the others add +0.0, the sum is
    exact) and writes the total back to its device buffer.  The contexts may sit on one GPU (tests, the `--rank-share`
    measurements of bench_chain.py -- the distributed step has no launch whose workgroups wait for each other) or on
    one GPU each (a one-process multi-GPU host without RCCL).  Plain HIP runtime calls through ctypes: no torch."""

    def __init__(self, contexts, order=None):
        import ctypes as C
        import threading
        from ._capi import REDUCE_FN
        self.C = C
        self.contexts = list(contexts)
        self.n = len(self.contexts)
        self.order = list(order) if order is not None else list(reversed(range(self.n)))
        assert sorted(self.order) == list(range(self.n))
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.barrier = threading.Barrier(self.n)
        self.parts = [None] * self.n
        self.calls = [0] * self.n
        self.errors = []
        self._keep = []
        for rank, ctx in enumerate(self.contexts):
            fn = REDUCE_FN(self._make(rank))
            self._keep.append(fn)
            ctx.call("comm_set_reduce_callback", C.cast(fn, C.c_void_p), None)

    def _make(self, rank):
        C, hip = self.C, self.hip

        def reduce(user, buffer, count, stream):
            try:
                if hip.hipStreamSynchronize(stream) != 0:
                    return 1
                part = np.empty(count, np.float32)
                if hip.hipMemcpy(part.ctypes.data_as(C.c_void_p), buffer, count * 4, 2) != 0:  # hipMemcpyDeviceToHost
                    return 2
                self.parts[rank] = part
                self.barrier.wait(timeout=60)
                total = self.parts[self.order[0]].copy()
                for r in self.order[1:]:
                    total += self.parts[r]
                self.barrier.wait(timeout=60)  # (everybody has read the parts before anybody's next call replaces one)
                if hip.hipMemcpy(buffer, total.ctypes.data_as(C.c_void_p), count * 4, 1) != 0:  # hipMemcpyHostToDevice
                    return 3
                self.calls[rank] += 1
                return 0
            except Exception as e:  # (a broken barrier, ...: the step fails with M3T_ERR_DEVICE)
                self.errors.append((rank, repr(e)))
                return 4
        return reduce

    def run(self, fn):
        """fn(rank, context) on one thread per rank, in lock-step wherever the contexts reduce; returns the results"""
        import threading
        out, failures = [None] * self.n, []

        def work(rank):
            try:
                out[rank] = fn(rank, self.contexts[rank])
            except BaseException as e:
                failures.append((rank, e))
                self.barrier.abort()
        threads = [threading.Thread(target=work, args=(r,)) for r in range(self.n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if failures:
            self.barrier.reset()
            raise failures[0][1]
        return out

    def close(self):
        for ctx in self.contexts:
            ctx.call("comm_set_reduce_callback", None, None)
        self._keep = []


class HostReduce:
    """m3t_hip_comm_set_reduce_callback with torch.distributed as the transport: the link sums of a rank go to the host,
    `dist.all_reduce(SUM)` adds them over the process group (gloo: works between processes that share one GPU, where
    RCCL refuses a second rank on the same device; MPI-style hosts do the same with their own call) and the total goes
    back to the device buffer.  What the library's own ncclAllReduce does in one call on the stream, done the slow way:
    for dry runs and for hosts without RCCL, not for speed."""

    def __init__(self, ctx, dist):
        import ctypes as C
        import torch
        from ._capi import REDUCE_FN
        self.ctx, self.calls = ctx, 0
        hip = C.CDLL("libamdhip64.so")
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.error = None

        def reduce(user, buffer, count, stream):
            try:
                if hip.hipStreamSynchronize(stream) != 0:
                    return 1
                part = torch.empty(count, dtype=torch.float32)
                if hip.hipMemcpy(part.data_ptr(), buffer, count * 4, 2) != 0:  # hipMemcpyDeviceToHost
                    return 2
                dist.all_reduce(part, op=dist.ReduceOp.SUM)
                if hip.hipMemcpy(buffer, part.data_ptr(), count * 4, 1) != 0:  # hipMemcpyHostToDevice
                    return 3
                self.calls += 1
                return 0
            except Exception as e:  # noqa: BLE001 (reported through the step's error code)
                self.error = repr(e)
                return 4
        self._fn = REDUCE_FN(reduce)
        ctx.call("comm_set_reduce_callback", C.cast(self._fn, C.c_void_p), None)

    def close(self):
        self.ctx.call("comm_set_reduce_callback", None, None)
