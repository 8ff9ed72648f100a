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
