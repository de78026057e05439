#!/usr/bin/env python3
"""bench.py — EDaGe-PP map+path instances/s on MI355X (BASELINE.json metric, config 2).

A "step" is one pass of the hot path over one batch of synthetic input: stage A for 100 target
paths (Philox seed) + stage B for 100 x 100 = 10 000 maps at R=256, K=20, clearance 3 — per GPU.
Problem instances shard embarrassingly (rank r owns path ids [r*100, (r+1)*100) and their maps, no
data-path collective); the only collective is the end-of-batch all-gather of the fixed-size
per-instance records over RCCL.  One JSON line is printed by rank 0.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

R, K, PATHS, PLACEMENTS = 256, 20, 100, 100
MAP_SIZE, CLEARANCE, OBST_SIZE, SEED = 50, 3, 5, 0
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_per_map(k_tot):
    """SURVEY.md §8(d) A_B(R) = 2*R^2 + 16384 + 12*K_tot (grid write + corridor mask read + labels + obstacles)."""
    return 2 * R * R + 16384 + 12.0 * k_tot


def cpu_baseline():
    """The CPU oracle (a NumPy port of the reference's algorithm) on a bounded sample of the same
    workload, one host core: 20 target paths x 100 placements at R=256, K=20 (~15 s).  Reported, not a target."""
    from oracle import edage_np as E
    n_paths, placements = 20, 100
    src = E.PhiloxSource(SEED)
    t0 = time.perf_counter()
    precs = E.generate_paths(src, n_paths, R, MAP_SIZE, CLEARANCE)
    t1 = time.perf_counter()
    maps = E.generate_maps(src, precs, R, MAP_SIZE, OBST_SIZE, K, CLEARANCE, placements)
    t2 = time.perf_counter()
    # same amortisation as the GPU step: stage A once per PLACEMENTS maps
    per_inst = (t1 - t0) / n_paths / PLACEMENTS + (t2 - t1) / len(maps)
    return {"value": round(1.0 / per_inst, 3), "unit": "instances/s", "cores": 1, "kind": "port",
            "sample": f"oracle/edage_np.py: {n_paths} paths ({(t1 - t0) / n_paths * 1e3:.0f} ms each) + {len(maps)} maps "
                      f"({(t2 - t1) / len(maps) * 1e3:.1f} ms each) at R={R}, K={K}; stage A amortised over {PLACEMENTS} placements",
            "host_cores": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ppnet_amd import edage

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs a torch.distributed.run launch with WORLD_SIZE={args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # buffers are allocated once and reused every step (resident in HBM)
    pb = edage.PathsBatch(PATHS, R, MAP_SIZE, CLEARANCE, dev)
    mb = edage.MapsBatch(PATHS * PLACEMENTS, R, K, dev)
    n_local = PATHS * PLACEMENTS
    rec_w = 2 + 2 + 22                       # angle, flags/n_obs, translation, segpoint[11,2] -> float64 record
    rec = torch.empty(n_local, rec_w, dtype=torch.float64, device=dev)
    gathered = torch.empty(world * n_local, rec_w, dtype=torch.float64, device=dev) if world > 1 else None

    def step(it):
        # a fresh batch every step: path / map ids advance so no two steps generate the same instances
        first_path = (it * world + rank) * PATHS
        edage.generate_paths(PATHS, R, MAP_SIZE, CLEARANCE, seed=SEED, first_path_id=first_path, device=dev, out=pb)
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record()
        edage.generate_maps(pb, PLACEMENTS, OBST_SIZE, K, seed=SEED, first_map_id=first_path * PLACEMENTS, out=mb)
        ev1.record()
        if world > 1:                         # end-of-batch gather of the fixed-size records (RCCL over xGMI)
            rec[:, 0] = mb.angle
            rec[:, 1] = mb.flags.to(torch.float64)
            rec[:, 2:4] = mb.translation.to(torch.float64)
            rec[:, 4:] = mb.segpoint.reshape(n_local, 22)
            dist.all_gather_into_tensor(gathered, rec)
        return ev0, ev1

    for it in range(args.warmup):
        step(it)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = [step(args.warmup + it) for it in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    maps_kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
    k_tot = float(mb.n_obstacles[:, 0].double().mean().item())
    placed = float(((mb.flags & 2) == 0).double().mean().item())
    total_instances = world * n_local * args.steps
    value = total_instances / elapsed

    if rank == 0:
        bytes_per_launch = algorithmic_bytes_per_map(k_tot) * n_local
        achieved = bytes_per_launch / (maps_kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "edage_pp_map_path_instances_per_sec",
            "value": round(value, 1),
            "unit": "instances/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "EDaGe-PP config 2: 100 target paths x 100 placements = 10000 maps+paths per GPU per step",
                       "resolution": R, "obstacles_num": K, "clearance": CLEARANCE, "map_size": MAP_SIZE,
                       "rng": "philox4x32-10", "parallelism": f"instances sharded over {world} GPU(s), end-of-batch all-gather"},
            "roofline": {"bound": "hbm", "kernel": "edage_maps_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "kernel_ms": round(maps_kernel_ms, 4),
                         "algorithmic_bytes_per_map": round(algorithmic_bytes_per_map(k_tot), 1)},
            "placement_success": round(placed, 4),
            "mean_obstacles_per_map": round(k_tot, 2),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
