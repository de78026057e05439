"""Instance sharding over the GPUs of one node (one process per GPU, torch.distributed).

Problem instances are independent: rank r of W owns a contiguous range of target paths and all of their
placements, and every random draw is keyed by the *global* path / map id, so results do not depend on W.
There is no data-path collective; the only exchange is the end-of-batch all-gather of fixed-size
per-instance records (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist

RECORD_WIDTH = 26      # angle, flags, translation[2], segpoint[11,2] as float64


def shard_range(n_total, rank, world):
    """Contiguous, balanced split of range(n_total): the first n_total % world ranks get one extra."""
    q, r = divmod(n_total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def local_ids(n_paths_total, placements, rank, world, batch_index=0):
    """(first_path_id, n_local_paths, first_map_id) of this rank for batch `batch_index`."""
    lo, hi = shard_range(n_paths_total, rank, world)
    base = batch_index * n_paths_total
    return base + lo, hi - lo, (base + lo) * placements


def pack_records(angle, flags, translation, segpoint, out=None):
    n = angle.shape[0]
    rec = out if out is not None else torch.empty(n, RECORD_WIDTH, dtype=torch.float64, device=angle.device)
    rec[:, 0] = angle
    rec[:, 1] = flags.to(torch.float64)
    rec[:, 2:4] = translation.to(torch.float64)
    rec[:, 4:] = segpoint.reshape(n, 22)
    return rec


PLAN_WAYPOINTS = 32    # waypoints carried in a plan record (the reference's result lists are of this order)
PLAN_RECORD_WIDTH = 4 + 2 * PLAN_WAYPOINTS


def pack_plan_records(result, lengths, out=None):
    """Loop B's end-of-batch record, one row per problem (the batched analogue of the result list the reference collects
    on rank 0, SegNet/mmseg/apis/test.py:232-235): ok, collision, waypoint count, plan length, then the first
    PLAN_WAYPOINTS waypoints (row, col) — when a plan has more, every k-th one so that start and goal are kept."""
    ok, coll, cnt, wp = result["ok"], result["collision"], result["counts"], result["waypoints"]
    B = ok.shape[0]
    rec = out if out is not None else torch.empty(B, PLAN_RECORD_WIDTH, dtype=torch.float64, device=ok.device)
    rec[:, 0] = ok.to(torch.float64)
    rec[:, 1] = coll.to(torch.float64)
    rec[:, 2] = cnt.to(torch.float64)
    rec[:, 3] = lengths
    last = (cnt.to(torch.int64) - 1).clamp(min=0)
    t = torch.arange(PLAN_WAYPOINTS, device=ok.device, dtype=torch.float64) / (PLAN_WAYPOINTS - 1)
    idx = torch.where(last[:, None] < PLAN_WAYPOINTS, torch.arange(PLAN_WAYPOINTS, device=ok.device)[None, :].clamp(max=wp.shape[1] - 1).expand(B, -1),
                      (t[None, :] * last[:, None].to(torch.float64)).round().to(torch.int64))
    idx = torch.minimum(idx, last[:, None])
    rec[:, 4:] = torch.gather(wp, 1, idx[:, :, None].expand(-1, -1, 2)).reshape(B, 2 * PLAN_WAYPOINTS)
    return rec


def gather_records(rec, world, sizes=None, group=None, out=None, always_collective=False):
    """All-gather of per-instance records; `sizes` = rows per rank when shards are uneven; `out` = a preallocated
    [world * rows, width] buffer for the even case (a steady-state loop should not allocate on a side stream).
    always_collective: issue the collective even for a group of one (rehearses the RCCL path on a single GPU)."""
    if world == 1 and not always_collective:
        return rec
    if sizes is None or len(set(sizes)) == 1:
        if out is None:
            out = torch.empty(world * rec.shape[0], rec.shape[1], dtype=rec.dtype, device=rec.device)
        dist.all_gather_into_tensor(out, rec.contiguous(), group=group)
        return out
    m = max(sizes)
    pad = torch.zeros(m, rec.shape[1], dtype=rec.dtype, device=rec.device)
    pad[:rec.shape[0]] = rec
    out = torch.empty(world * m, rec.shape[1], dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * m:r * m + sizes[r]] for r in range(world)])


def agreed_max(value, world, device=None):
    """The largest `value` over the ranks, on every rank (one all-reduce; `value` itself when world == 1).  bench.py decides with it
    whether the untimed clock-warm phase goes on: every step carries a collective at N > 1, so all ranks must take the same number."""
    if world <= 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
