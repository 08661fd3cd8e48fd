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


def gather_doubles_dist(dist):
    """gather(local, counts) over torch.distributed (gloo on CPU, nccl = RCCL on GPU): the transport of the
    tests; the product's transport is Context.allgather_chains (the exported RCCL path)."""
    import torch

    def gather(local, counts):
        t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64).reshape(-1))
        if dist.get_backend() == "nccl":
            t = t.cuda()
        return allgather_ragged(t, [int(c) for c in counts], dist).cpu().numpy()

    return gather


def bag_to_kml_sharded(bags, stamps, gps_log, slam, tracks, rank=0, world=1, gather=None, kml_original="",
                       kml_calibrated=""):
    """bag -> KML with the SLAM stage sharded over ranks (BASELINE configs[3] / [4]).

    Units = bags (a bag's segments are cut online from its own track, input_data.cpp:78-124, so a bag is the
    smallest independent piece; LOAM restarts per segment anyway).  Rank r runs `slam` on its contiguous block of
    bags; the pose chains of every long / short segment are exchanged with ONE ragged all-gather (`gather`, three
    calls: sizes, segment headers, rows); then the global track alignment -- long pass (GPS weights), short pass
    (fits), overlap merge (short_distance_track_process.cpp:73-158) and the KML writer -- runs on rank 0 through
    `tracks`.  Segments are put into (pass, bag, first sweep) order before the global stage, so the files do not
    depend on the number of ranks.

    slam(bags, stamps) -> list of dict(flag, bag, first, track[n,4])   (Context.input_data_run partially applied)
    tracks(gps_log, longs, shorts, kml_original, kml_calibrated) -> anything   (pipeline.run_tracks)
    gather(local_doubles, counts) -> all doubles in rank order              (Context.allgather_chains)
    Returns dict(seconds=[slam, exchange, global], segments=[long, short], result=tracks(...) on rank 0)."""
    import time
    t0 = time.perf_counter()
    lo, hi = shard_range(len(bags), rank, world)
    mine = slam(bags[lo:hi], stamps[lo:hi]) if hi > lo else []
    mine = [t for t in mine if len(t["track"])]
    t1 = time.perf_counter()
    head = np.array([[t["flag"], lo + t["bag"], t["first"], len(t["track"])] for t in mine], dtype=np.float64).reshape(-1)
    rows = (np.concatenate([t["track"] for t in mine]) if mine else np.zeros((0, 4))).reshape(-1)
    if world > 1:
        sizes = gather(np.array([len(mine), len(rows) // 4], dtype=np.float64), [2] * world).reshape(world, 2).astype(np.int64)
        head = gather(head, [4 * int(c) for c in sizes[:, 0]])
        rows = gather(rows, [4 * int(c) for c in sizes[:, 1]])
    head = head.reshape(-1, 4)
    rows = rows.reshape(-1, 4)
    t2 = time.perf_counter()
    starts = np.r_[0, np.cumsum(head[:, 3].astype(np.int64))]
    segs = [(int(h[0]), int(h[1]), int(h[2]), rows[starts[k]:starts[k + 1]]) for k, h in enumerate(head)]
    segs.sort(key=lambda s: s[:3])
    longs = [s[3] for s in segs if s[0] == 0]
    shorts = [s[3] for s in segs if s[0] == 1]
    result = tracks(gps_log, longs, shorts, kml_original, kml_calibrated) if rank == 0 else None
    t3 = time.perf_counter()
    return {"seconds": [t1 - t0, t2 - t1, t3 - t2], "segments": [len(longs), len(shorts)], "result": result}
