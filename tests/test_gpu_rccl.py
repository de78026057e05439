"""The RCCL path on one GPU: a process group of one on backend "nccl" (= RCCL on ROCm), the end-of-batch all-gather of
stage B's device-resident records and of loop B's plan records issued as real collectives on a side stream — so that RCCL
initialisation and `all_gather_into_tensor` on HBM buffers are not first exercised on the 8-GPU node."""
import datetime
import os
import socket

import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


def test_rccl_group_of_one_gathers_records():
    import torch.distributed as dist
    from ppnet_amd import edage, evaluate, shard
    from ppnet_amd.ppnet import PPNet
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev,
                            timeout=datetime.timedelta(seconds=180))
    try:
        R = 64
        pb = edage.generate_paths(3, R, 50, 3, seed=9, device=dev)
        mb = edage.generate_maps(pb, 4, 5, 20, seed=9)
        ev = torch.cuda.Event(); ev.record()
        s_comm = torch.cuda.Stream(dev)
        out = torch.full((mb.n, shard.RECORD_WIDTH), -1.0, dtype=torch.float64, device=dev)
        with torch.cuda.stream(s_comm):                                       # bench.py's exchange, as a real collective
            s_comm.wait_event(ev)
            got = shard.gather_records(mb.records, 1, out=out, always_collective=True)
        s_comm.synchronize()
        assert got is out and torch.equal(out, mb.records)
        want = shard.pack_records(mb.angle, mb.flags, mb.translation, mb.segpoint)
        assert torch.equal(out, want)                                         # the kernel-written record == the packed one
        # loop B's record through the same collective
        heat = evaluate.label_heatmaps(pb, mb, 4)
        tiny = torch.nn.Module(); tiny.prepare_inference = lambda: tiny
        p = PPNet(R, segnet=tiny, gennet=tiny)
        res = p.plan_tail(heat, mb.segpoint[:, 0].contiguous(), mb.segpoint[:, 10].contiguous(), mb.obstacles,
                          mb.n_obstacles[:, 0].contiguous())
        rec = shard.pack_plan_records(res, evaluate.plan_lengths(res["waypoints"], res["counts"]))
        gathered = shard.gather_records(rec, 1, always_collective=True)
        t = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                              # bench.py's max-over-ranks timing
        dist.barrier()
        torch.cuda.synchronize()
        assert torch.equal(gathered, rec) and float(t) == 1.5
        assert int(rec[:, 0].sum()) >= 1                                      # at least one ridge walk reached its goal
        # bench.py's record ring as real collectives: stage B writes three steps straight into the ring's slices, one
        # all_gather_into_tensor per group of two steps on the communication stream, the third step flushed alone
        ring = shard.RecordRing(1, mb.n, 2, dev, s_comm, collective="always")
        views = [[mb.with_records(ring.slot_view(sl, k)) for k in range(2)] for sl in range(2)]
        shipped = []
        for it in range(3):
            ring.begin_step()
            edage.generate_maps(pb, 4, 5, 20, seed=9, first_map_id=it * mb.n, out=views[ring.slot][ring.k])
            before = ring.n_gathers
            ring.end_step()
            if ring.n_gathers != before:
                shipped.append(ring.last)
        tail = ring.flush()
        s_comm.synchronize(); torch.cuda.synchronize()
        assert len(shipped) == 1 and shipped[0][1] == 2 and ring.last[1] == 1 and ring.n_gathers == 2
        for it in range(3):
            want = edage.generate_maps(pb, 4, 5, 20, seed=9, first_map_id=it * mb.n).records
            got = ring.rank_rows(shipped[0][0], 2, 0, it) if it < 2 else ring.rank_rows(tail, 1, 0, 0)
            assert torch.equal(got, want)
    finally:
        dist.destroy_process_group()
