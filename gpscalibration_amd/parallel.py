"""Sharding of the hot path across GPUs: one process per GPU, units = scan pairs or SLAM
segments (each segment's LOAM state is reset, so units are independent -- SURVEY.md 3.1/8e).

Contiguous blocks of ceil(S/G) units per rank keep the short-pass overlap merge local except
at the G-1 seams.  The only exchange is an all-gather of per-unit results (4x4 poses or pose
chains); torch.distributed is the transport (backend "nccl" = RCCL on ROCm, "gloo" on CPU).
"""
import numpy as np


def shard_range(n_units, rank, world):
    """[lo, hi) of the contiguous block owned by `rank`; blocks differ by at most one unit."""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_counts(n_units, world):
    return [shard_range(n_units, r, world)[1] - shard_range(n_units, r, world)[0] for r in range(world)]


def allgather_ragged(local, counts, dist, device=None):
    """All-gather of float64 rows whose count differs per rank (pose chains).

    local: torch tensor [counts[rank], ...]; returns [sum(counts), ...] in rank order.
    Pads to the maximum count so that ONE all_gather_into_tensor moves everything
    (payloads are KBs..MBs: latency-bound, a single collective is the right shape).
    """
    import torch
    world = dist.get_world_size()
    rank = dist.get_rank()
    assert len(counts) == world and local.shape[0] == counts[rank]
    cmax = max(counts)
    tail = tuple(local.shape[1:])
    pad = torch.zeros((cmax,) + tail, dtype=local.dtype, device=local.device)
    pad[:counts[rank]] = local
    out = torch.empty((world * cmax,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    parts = [out[r * cmax:r * cmax + counts[r]] for r in range(world)]
    return torch.cat(parts, dim=0)


def gather_segment_results(per_seg_arrays, seg_lengths_all, rank, world, dist):
    """Gathers a per-pose array (rows = poses of this rank's segments) from every rank.

    seg_lengths_all: lengths of ALL segments (global); this rank owns shard_range(len, rank, world).
    Returns the global per-pose array in segment order.
    """
    import torch
    nseg = len(seg_lengths_all)
    counts = []
    for r in range(world):
        lo, hi = shard_range(nseg, r, world)
        counts.append(int(np.sum(seg_lengths_all[lo:hi])))
    t = per_seg_arrays if isinstance(per_seg_arrays, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(per_seg_arrays))
    return allgather_ragged(t, counts, dist)


def loam_run_sharded(ctx, segments, stamps, dist=None):
    """The LOAM node chain over segments sharded in contiguous blocks across ranks (one process
    per GPU), followed by the one exchange of SURVEY 8(e): an all-gather of the per-segment
    /true_odometry_to_init tracks.  Every rank returns the tracks of ALL segments (list of [n,4]
    float64), ready for the global track alignment.  dist=None runs everything on this rank."""
    import torch
    nseg = len(segments)
    if dist is None or dist.get_world_size() == 1:
        return [r["track"] for r in ctx.loam_run(segments, stamps)]
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_range(nseg, rank, world)
    mine = ctx.loam_run(segments[lo:hi], stamps[lo:hi]) if hi > lo else []
    lens = np.array([len(s) for s in segments])
    local = np.concatenate([r["track"] for r in mine]) if mine else np.zeros((0, 4))
    t = torch.from_numpy(np.ascontiguousarray(local))
    if dist.get_backend() == "nccl":
        t = t.cuda()
    full = gather_segment_results(t, lens, rank, world, dist).cpu().numpy()
    starts = np.r_[0, np.cumsum(lens)]
    return [full[starts[s]:starts[s + 1]] for s in range(nseg)]
