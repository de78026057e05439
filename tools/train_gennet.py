#!/usr/bin/env python3
"""Train GenNet (AE-ViT, 53 713 parameters at R = 256) with the build's own training step on pairs from the build's own generator,
and report the planner's success rate on what the TRAINED network predicts.

The reference's loop: GenNet/train.py:93-147 (AdamW(1e-3, betas (0, 0.99)), MSE against mask_path, PolyLR per iteration, batch
size from the command line, 320 000 (mask_space, mask_path) pairs read back from PNG files, my_dataset.py:44-50) and then
EDaGe-PP/process_map.py:452-506 (extract_path on the network's normalised output + collision check) with the OMPL harness's
(1 + eps) x target-length criterion (updated_geometric_planner.py:260-277).  Here every step draws a FRESH batch from the stage-A/B
kernels on the device (ppnet_amd.train.generator_pairs: `paths` target paths x `placements` placements), so nothing is read from
disk and no sample repeats; the step itself is ppnet_amd.train.gennet_train_step (autograd through the ROCm libraries — the fused
inference kernels are forward-only and step aside while autograd records).

    python tools/train_gennet.py --R 256 --steps 6000 --out ppnet_amd/weights/gennet_r256.pth

The checkpoint is the reference's layout, {'model': state_dict} with float32 tensors (predict.py:51-52 loads exactly that key), about
215 KB; it is what bench.py's `ppnet` / `end_to_end_r512` legs load for `tail_network_output`.  Evaluation (every --eval-every steps
and at the end, on a held-out seed): the eval-mode network in the PREPARED bf16 inference form bench.py times (PPNet.heatmap on the
label mask_space), the planner tail on that heat map, evaluate.evaluate_plans -> success / length_ratio / within_eps."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def make_batch(edage, train, dev, R, paths_n, placements, seed, first_path, first_map, K=20, clearance=3.0, map_size=50.0, obstacles_size=5.0):
    pb = edage.generate_paths(paths_n, R, map_size, clearance, seed=seed, first_path_id=first_path, device=dev)
    mb = edage.generate_maps(pb, placements, obstacles_size, K, seed=seed, first_map_id=first_map)
    grid, mask_space, mask_path = train.generator_pairs(pb, mb, placements)
    return pb, mb, mask_space, mask_path


def evaluate(torch, model, dev, R, args, seed=987654321):
    """Held-out problems through the PREPARED inference path (what bench.py runs): bf16 AE-ViT on the label masks -> u8 heat map ->
    extract_path + collision check -> the harness's criterion."""
    import copy
    from ppnet_amd import _lib as L, edage, evaluate as EV, fused, plan, train
    pb, mb, mask_space, mask_path = make_batch(edage, train, dev, R, args.eval_paths, args.eval_placements, seed, 0, 0)
    gen = copy.deepcopy(model).eval()
    gen.prepare_inference()                                       # PPNet.__init__: BatchNorm folded, fused stages, bf16 weights
    gen.to(torch.bfloat16)
    with torch.no_grad():
        heat = fused.heatmap_u8(gen(mask_space.to(torch.bfloat16).unsqueeze(1)))          # PPNet.heatmap
        init, end = mb.segpoint[:, 0].contiguous(), mb.segpoint[:, 10].contiguous()
        ok, wp, cnt = plan.extract_paths(heat, init, end, 2, L.MAX_WAYPOINTS)           # PPNet.plan_tail
        coll = plan.plan_collision(wp, cnt, mb.obstacles, mb.n_obstacles[:, 0].contiguous(), 1 / 50 * R, bound=R)
        res = dict(ok=ok, waypoints=wp, counts=cnt, collision=coll, success=ok & ~coll)
        target_px = pb.length.repeat_interleave(args.eval_placements) * (R / 50.0)
        ev = EV.evaluate_plans(res, target_px)
        mse = float(train.evaluate_gennet(copy.deepcopy(model), mask_space, mask_path))
    ev["val_mse"] = mse
    return ev


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--R", type=int, default=256)
    ap.add_argument("--steps", type=int, default=6000)
    ap.add_argument("--paths", type=int, default=8, help="target paths per step")
    ap.add_argument("--placements", type=int, default=8, help="placements per target path (batch = paths x placements)")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--amp", action="store_true", help="bf16 autocast for the forward / backward")
    ap.add_argument("--eval-every", type=int, default=1000)
    ap.add_argument("--eval-paths", type=int, default=32)
    ap.add_argument("--eval-placements", type=int, default=8)
    ap.add_argument("--seed", type=int, default=20260105)
    ap.add_argument("--minutes", type=float, default=0.0, help="stop after this much wall time (0 = run all steps)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--log", default=None)
    args = ap.parse_args()

    from ppnet_amd import edage, train
    from ppnet_amd.gennet import AEViT
    dev = torch.device("cuda:0")
    torch.manual_seed(args.seed)
    R = args.R
    model = AEViT(1, 1, R, 24).to(dev)
    opt = train.gennet_optimizer(model, args.lr)
    sched = train.PolyLR(opt, args.steps, power=0.9)
    batch = args.paths * args.placements
    log = {"R": R, "steps": args.steps, "batch": batch, "lr": args.lr, "amp": bool(args.amp), "params": sum(p.numel() for p in model.parameters()),
           "history": []}
    t0 = time.time()
    loss_acc, n_acc = 0.0, 0
    step = 0
    for step in range(1, args.steps + 1):
        pb, mb, mask_space, mask_path = make_batch(edage, train, dev, R, args.paths, args.placements, args.seed,
                                                   step * args.paths, step * batch)
        loss = train.gennet_train_step(model, opt, sched, mask_space, mask_path, amp_dtype=torch.bfloat16 if args.amp else None)
        loss_acc += float(loss) if step % 50 == 0 else 0.0
        n_acc += 1 if step % 50 == 0 else 0
        if step % args.eval_every == 0 or step == args.steps:
            ev = evaluate(torch, model, dev, R, args)
            rec = {"step": step, "train_mse": round(loss_acc / max(n_acc, 1), 6), "wall_s": round(time.time() - t0, 1),
                   **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in ev.items()}}
            loss_acc, n_acc = 0.0, 0
            log["history"].append(rec)
            print(json.dumps(rec), flush=True)
            if args.log:                                              # progress on disk: a run behind a pipe looks hung to the GPU box's watchdog
                os.makedirs(os.path.dirname(os.path.abspath(args.log)), exist_ok=True)
                with open(args.log + ".progress", "a") as f:
                    f.write(json.dumps(rec) + "\n")
        if args.minutes and time.time() - t0 > args.minutes * 60:
            break
    if step != args.steps and (not log["history"] or log["history"][-1]["step"] != step):
        ev = evaluate(torch, model, dev, R, args)
        log["history"].append({"step": step, "wall_s": round(time.time() - t0, 1), **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in ev.items()}})
        print(json.dumps(log["history"][-1]), flush=True)
    log["samples_seen"] = step * batch
    log["samples_per_s"] = round(step * batch / (time.time() - t0), 1)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        sd = {k: v.detach().float().cpu().contiguous() for k, v in model.state_dict().items()}
        torch.save({"model": sd}, args.out)                          # GenNet/train.py:133-141's 'model' entry, predict.py:51-52 reads it
        log["checkpoint"] = {"path": os.path.relpath(args.out, ROOT), "bytes": os.path.getsize(args.out)}
    print(json.dumps({k: v for k, v in log.items() if k != "history"}), flush=True)
    if args.log:
        os.makedirs(os.path.dirname(os.path.abspath(args.log)), exist_ok=True)
        with open(args.log, "w") as f:
            json.dump(log, f, indent=1)


if __name__ == "__main__":
    main()
