"""The N>1 path on CPU: two gloo ranks shard the instances by global id, generate their shard (with the CPU
oracle standing in for the kernels, which need a GPU) and all-gather the per-instance records; the result
must equal the single-process run — i.e. sharding changes neither ids nor random streams."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import edage_np as E
from ppnet_amd import shard

R, K, PATHS, PLACEMENTS, SEED = 64, 8, 5, 3, 21


def _records(first_path, n_paths, first_map):
    src = E.PhiloxSource(SEED)
    precs = E.generate_paths(src, n_paths, R, 50, 3, first_path_id=first_path)
    maps = E.generate_maps(src, precs, R, 50, 5, K, 3, PLACEMENTS, first_map_id=first_map, want_grid=False)
    return shard.pack_records(torch.tensor([m["angle"] for m in maps]),
                              torch.tensor([m["flags"] for m in maps]),
                              torch.tensor(np.array([m["translation"] for m in maps])),
                              torch.tensor(np.array([m["segpoint"] for m in maps])))


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first_path, n_local, first_map = shard.local_ids(PATHS, PLACEMENTS, rank, world)
    rec = _records(first_path, n_local, first_map)
    sizes = [(shard.shard_range(PATHS, r, world)[1] - shard.shard_range(PATHS, r, world)[0]) * PLACEMENTS for r in range(world)]
    full = shard.gather_records(rec, world, sizes=sizes)
    torch.save(full, os.path.join(out_dir, f"rank{rank}.pt"))
    # the even-shard form the bench uses: preallocated record and gather buffers (each rank contributes its first 6 rows)
    buf = torch.empty(6, shard.RECORD_WIDTH, dtype=torch.float64)
    packed = shard.pack_records(rec[:6, 0], rec[:6, 1].to(torch.int32), rec[:6, 2:4].to(torch.int32), rec[:6, 4:].reshape(6, 11, 2), out=buf)
    assert packed is buf and torch.equal(buf, rec[:6])
    out = torch.full((world * 6, shard.RECORD_WIDTH), -1.0, dtype=torch.float64)
    got = shard.gather_records(buf, world, out=out)
    assert got is out
    torch.save(out, os.path.join(out_dir, f"even{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _warm_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # bench.py's clock-warm loop with each rank on its own (skewed) clock: the chunks every rank runs must be the same number
    clock, chunks = 0.0, 0
    while True:
        if shard.agreed_max(clock, world) >= 1.0:
            break
        chunks += 1
        clock += 0.21 if rank == 0 else 0.34          # rank 1's steps are slower
        dist.all_reduce(torch.zeros(1))               # the collective a step carries: a rank that ran one more would hang here
    with open(os.path.join(out_dir, f"warm{rank}.txt"), "w") as f:
        f.write(str(chunks))
    dist.barrier()
    dist.destroy_process_group()


def test_clock_warm_phase_takes_the_same_steps_on_every_rank(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_warm_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = (int(open(tmp_path / f"warm{r}.txt").read()) for r in range(2))
    assert a == b == 3                                # the slowest clock decides: 0.34, 0.68, 1.02
    assert shard.agreed_max(0.5, 1) == 0.5


def test_shard_ranges_cover_exactly():
    for n, w in [(100, 1), (100, 8), (5, 2), (7, 4), (3, 8)]:
        seen = []
        for r in range(w):
            lo, hi = shard.shard_range(n, r, w)
            seen += list(range(lo, hi))
        assert seen == list(range(n))
    assert shard.local_ids(100, 100, 3, 8, batch_index=2) == (200 + 39, 13, (200 + 39) * 100)   # 100 = 4*13 + 4*12


def test_two_ranks_gloo_equal_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    whole = _records(0, PATHS, 0)
    for r in range(2):
        got = torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=True)
        assert got.shape == whole.shape
        assert torch.equal(got, whole)          # bit-exact: same ids, same streams, any world size
    n0 = (shard.shard_range(PATHS, 0, 2)[1]) * PLACEMENTS
    even = torch.cat([whole[:6], whole[n0:n0 + 6]])
    for r in range(2):
        assert torch.equal(torch.load(os.path.join(tmp_path, f"even{r}.pt"), weights_only=True), even)


def test_plan_records_layout():
    """Loop B's fixed-size result record: flags, count, length, then <= 32 waypoints with start and goal kept."""
    wp = torch.zeros(3, 50, 2, dtype=torch.float64)
    wp[0, :5] = torch.arange(10, dtype=torch.float64).reshape(5, 2)
    wp[1, :40] = torch.arange(80, dtype=torch.float64).reshape(40, 2)
    res = dict(ok=torch.tensor([True, True, False]), collision=torch.tensor([False, True, False]),
               counts=torch.tensor([5, 40, 0], dtype=torch.int32), waypoints=wp)
    rec = shard.pack_plan_records(res, torch.tensor([12.5, 99.0, 0.0], dtype=torch.float64))
    assert rec.shape == (3, shard.PLAN_RECORD_WIDTH)
    assert rec[:, :4].tolist() == [[1, 0, 5, 12.5], [1, 1, 40, 99.0], [0, 0, 0, 0.0]]
    w0 = rec[0, 4:].reshape(-1, 2)
    assert torch.equal(w0[:5], wp[0, :5]) and torch.equal(w0[5:], wp[0, 4].expand(27, 2))     # short plan: padded with the goal
    w1 = rec[1, 4:].reshape(-1, 2)
    assert torch.equal(w1[0], wp[1, 0]) and torch.equal(w1[-1], wp[1, 39])                      # long plan: subsampled, ends kept
    assert bool((w1[1:, 0] >= w1[:-1, 0]).all())


def _plan_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    B = 4
    cnt = torch.randint(0, 60, (B,), generator=g, dtype=torch.int32)
    res = dict(ok=cnt > 0, collision=torch.rand(B, generator=g) > 0.5, counts=cnt,
               waypoints=torch.rand(B, 64, 2, generator=g, dtype=torch.float64))
    rec = shard.pack_plan_records(res, torch.rand(B, generator=g, dtype=torch.float64))
    out = torch.empty(world * B, shard.PLAN_RECORD_WIDTH, dtype=torch.float64)
    shard.gather_records(rec, world, out=out)
    torch.save((rec, out), os.path.join(out_dir, f"plan{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_plan_records_gather_two_ranks_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_plan_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, o0 = torch.load(os.path.join(tmp_path, "plan0.pt"), weights_only=True)
    r1, o1 = torch.load(os.path.join(tmp_path, "plan1.pt"), weights_only=True)
    assert torch.equal(o0, torch.cat([r0, r1])) and torch.equal(o1, o0)


def _ring_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, G, steps = 4, 3, 8                          # 8 steps = two full groups + a partial one of 2 (flushed at the end)
    ring = shard.RecordRing(world, rows, G, "cpu")
    groups = []
    for it in range(steps):
        out = ring.begin_step()
        # what stage B writes for step `it` on this rank: a value that names (rank, step, row, column)
        out.copy_(torch.arange(rows * shard.RECORD_WIDTH, dtype=torch.float64).reshape(rows, -1) + 1000.0 * it + 100000.0 * rank)
        before = ring.n_gathers
        ring.end_step()
        if ring.n_gathers != before:
            groups.append((ring.last[0].clone(), ring.last[1]))
    tail = ring.flush()
    groups.append((tail.clone(), ring.last[1]))
    assert ring.flush() is None                        # nothing staged: no collective
    torch.save(groups, os.path.join(out_dir, f"ring{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_record_ring_ships_groups_of_steps_in_one_collective(tmp_path):
    """bench.py's end-of-batch exchange at N > 1: the records of G steps staged in a ring, ONE all-gather per group, a partial
    group flushed at the end; every rank receives every rank's rows of every step, in (rank, step) order."""
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_ring_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(tmp_path / f"ring{r}.pt") for r in range(world)]
    assert [k for _, k in got[0]] == [3, 3, 2]
    base = torch.arange(4 * shard.RECORD_WIDTH, dtype=torch.float64).reshape(4, -1)
    step0 = 0
    for gi, (out, k) in enumerate(got[0]):
        assert torch.equal(out, got[1][gi][0])        # both ranks hold the same gathered group
        assert out.shape == (world * k * 4, shard.RECORD_WIDTH)
        for r in range(world):
            for j in range(k):
                want = base + 1000.0 * (step0 + j) + 100000.0 * r
                assert torch.equal(out[(r * k + j) * 4:(r * k + j + 1) * 4], want)
        step0 += k
    assert shard.gather_steps(8, 10000) == 13 and shard.gather_steps(1, 10000) == 100 and shard.gather_steps(8, 10 ** 7) == 1
