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


def all_gather_rows(out, src, group=None):
    """dist.all_gather_into_tensor, RCCL for CUDA tensors.  Under a gloo group CUDA rows travel through the host: that is the
    REHEARSAL transport of `bench.py` with BENCH_REHEARSE_SHARED_GPU (several ranks on one GPU cannot form an RCCL communicator);
    the control flow around the call — streams, events, slots, barriers — is the production one."""
    if src.is_cuda and dist.get_backend(group) == "gloo":
        torch.cuda.current_stream(src.device).synchronize()
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, src.cpu().contiguous(), group=group)
        out.copy_(host)
        return
    dist.all_gather_into_tensor(out, src, group=group)


def max_over_ranks(value, device, group=None):
    """The slowest rank's clock (bench.py's contract: MAX over ranks of the timed region)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if dist.get_backend(group) == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def gather_records(rec, world, sizes=None, group=None, out=None, always_collective=False):
    """All-gather of per-instance records; `sizes` = rows per rank when shards are uneven; `out` = a preallocated
    [world * rows, width] buffer for the even case (a steady-state loop should not allocate on a side stream).
    always_collective: issue the collective even for a group of one (rehearses the RCCL path on a single GPU)."""
    if world == 1 and not always_collective:
        return rec
    if sizes is None or len(set(sizes)) == 1:
        if out is None:
            out = torch.empty(world * rec.shape[0], rec.shape[1], dtype=rec.dtype, device=rec.device)
        all_gather_rows(out, rec.contiguous(), group)
        return out
    m = max(sizes)
    pad = torch.zeros(m, rec.shape[1], dtype=rec.dtype, device=rec.device)
    pad[:rec.shape[0]] = rec
    out = torch.empty(world * m, rec.shape[1], dtype=rec.dtype, device=rec.device)
    all_gather_rows(out, pad, group)
    return torch.cat([out[r * m:r * m + sizes[r]] for r in range(world)])


def gather_steps(world, rows_per_step, target_rows=1_000_000):
    """Steps between two end-of-batch gathers: enough that one collective carries the records of >= `target_rows` instances
    over all ranks (SURVEY 8e sizes the exchange as ONE gather per config-4 batch of 1 M instances, not one per 10 000-map
    kernel launch: a step is 0.19 ms, a collective launch costs tens of microseconds and CUs beside it)."""
    return max(1, -(-int(target_rows) // (max(int(world), 1) * int(rows_per_step))))


class RecordRing:
    """The records of G consecutive steps staged in device memory and shipped by ONE all-gather (the end-of-batch exchange),
    double-buffered: while the collective of group g reads slot g % 2 on the communication stream, the steps of group g + 1
    write the other slot.  Stage B writes a step's records straight into `slot_view(k)` (ppn_maps_t.records), so there is no
    pack pass and no copy in front of the collective.

        ring = RecordRing(world, rows_per_step, G, device, comm_stream)
        for every step:  out = ring.begin_step()      # [rows, width] slice to write; waits (on the current stream) for the
                         ...launch the producer...     #   slot's previous collective before its first step only
                         ring.end_step()               # ships the group after its G-th step
        ring.flush()                                   # ships a partial group (end of a timed region / of the job)

    On a CUDA device the hand-offs are events between the current (compute) stream and `comm_stream`; on the CPU (gloo tests)
    the same calls run synchronously.  world == 1 with collective=False copies instead of gathering (a one-GPU rehearsal of
    the stream logic)."""

    def __init__(self, world, rows_per_step, steps, device, comm_stream=None, width=RECORD_WIDTH, group=None, collective=True):
        self.world, self.rows, self.G, self.width = int(world), int(rows_per_step), int(steps), int(width)
        self.device, self.comm, self.group = torch.device(device), comm_stream, group
        self.collective = collective and (self.world > 1 or collective == "always")
        kw = dict(dtype=torch.float64, device=self.device)
        self.ring = [torch.empty(self.G * self.rows, self.width, **kw) for _ in range(2)]
        self.gathered = [torch.empty(max(self.world, 1) * self.G * self.rows, self.width, **kw) for _ in range(2)]
        self.sent = [None, None]            # event: the slot's records have been read by its collective
        self.slot, self.k = 0, 0
        self.n_gathers, self.last = 0, None  # `last` = (gathered rows view, steps in it) of the newest shipped group

    def slot_view(self, slot, k):
        return self.ring[slot][k * self.rows:(k + 1) * self.rows]

    def begin_step(self):
        if self.k == 0 and self.sent[self.slot] is not None:
            if self.device.type == "cuda":
                torch.cuda.current_stream(self.device).wait_event(self.sent[self.slot])
            self.sent[self.slot] = None
        return self.slot_view(self.slot, self.k)

    def end_step(self):
        self.k += 1
        if self.k == self.G:
            self.flush()

    def flush(self):
        k, slot = self.k, self.slot
        if k == 0:
            return None
        src = self.ring[slot][:k * self.rows]
        out = self.gathered[slot][:max(self.world, 1) * k * self.rows]

        def ship():
            if self.collective:
                all_gather_rows(out, src, self.group)
            else:
                out.copy_(src)
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record()                                     # the group's last producer, on the compute stream
            comm = self.comm if self.comm is not None else torch.cuda.current_stream(self.device)
            with torch.cuda.stream(comm):
                comm.wait_event(ev)
                ship()
                self.sent[slot] = torch.cuda.Event()
                self.sent[slot].record(comm)
        else:
            ship()
        self.slot, self.k = slot ^ 1, 0
        self.n_gathers += 1
        self.last = (out, k)
        return out

    def rank_rows(self, out, k, rank, step):
        """Rows of `rank`'s step `step` inside a gathered group of k steps (all_gather lays ranks out one after the other)."""
        base = (rank * k + step) * self.rows
        return out[base:base + self.rows]


def agreed_max(value, world, device=None):
    """The largest `value` over the ranks, on every rank (one all-reduce; `value` itself when world == 1).  bench.py decides with it
    whether the untimed clock-warm phase goes on: the steps feed collectives at N > 1, so all ranks must take the same number."""
    if world <= 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
