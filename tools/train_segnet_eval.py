#!/usr/bin/env python3
"""SegNet after N training steps with the build's own training step, and the planner's success through the FULL chain (VERDICT r04 item 5:
"a SegNet-after-N-steps figure — mIoU, success through the full chain — goes into profiles/r05_train_eval.txt").

Reference: SegNet/mmseg/apis/train.py:67-167 with configs/nat/setr_up_nat_base.py:46-56 (SGD 0.08, momentum 0.9, head x 10, poly with
linear warm-up; the reference starts from an ImageNet checkpoint, dinat_base.py:16-17, and runs 160 000 iterations at batch 16 — there is
no network here and no such checkpoint, so this is a FROM-SCRATCH run of a bounded number of minutes: what the figure says is "the
training step trains", not what a converged SegNet reaches).  Every step draws a fresh batch from the generator kernels
(ppnet_amd.train.generator_pairs: occupancy codes -> normalised image, labels = mask_space).  Evaluation on a held-out seed: mean IoU of
the two classes (mmseg's mIoU), pixel accuracy, and the chain SegNet labels -> the build's TRAINED GenNet (ppnet_amd/weights) -> 8-bit
heat map -> extract_path + collision check -> the OMPL harness's success / length criterion (process_map.py:452-506,
updated_geometric_planner.py:260-277), beside the same chain fed with the label masks.  The 400 MB checkpoint is not kept.

    python tools/train_segnet_eval.py --minutes 8 --out gpurun_out/r05/train_eval.txt"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def chain_eval(torch, mask_u8, pb, mb, placements, R, gen):
    from ppnet_amd import _lib as L, evaluate as EV, fused, plan
    with torch.no_grad():
        heat = fused.heatmap_u8(gen(mask_u8.to(torch.bfloat16).unsqueeze(1)))
        init, end = mb.segpoint[:, 0].contiguous(), mb.segpoint[:, 10].contiguous()
        ok, wp, cnt = plan.extract_paths(heat, init, end, 2, L.MAX_WAYPOINTS)
        coll = plan.plan_collision(wp, cnt, mb.obstacles, mb.n_obstacles[:, 0].contiguous(), 1 / 50 * R, bound=R)
        return EV.evaluate_plans(dict(ok=ok, waypoints=wp, counts=cnt, collision=coll, success=ok & ~coll),
                                 pb.length.repeat_interleave(placements) * (R / 50.0))


def evaluate(torch, net, dev, R, gen, paths_n=16, placements=8, seed=987654321):
    from ppnet_amd import edage, fused, train
    from ppnet_amd.segnet import IMG_MEAN, IMG_STD
    pb = edage.generate_paths(paths_n, R, 50.0, 3.0, seed=seed, device=dev)
    mb = edage.generate_maps(pb, placements, 5.0, 20, seed=seed)
    grid, space, _ = train.generator_pairs(pb, mb, placements)
    net.eval()
    labels = []
    with torch.no_grad():
        for i in range(0, grid.shape[0], 16):
            img = fused.grid_to_image(grid[i:i + 16], IMG_MEAN, IMG_STD, torch.float32)
            labels.append(net.encode_decode(img).argmax(1).to(torch.uint8))
    lab = torch.cat(labels)
    net.train()
    gt = space.to(torch.uint8)
    ious = []
    for c in (0, 1):
        inter = ((lab == c) & (gt == c)).sum().item()
        union = ((lab == c) | (gt == c)).sum().item()
        ious.append(inter / max(union, 1))
    out = {"mIoU": sum(ious) / 2, "IoU_free_corridor": ious[1], "IoU_background": ious[0], "pixel_acc": float((lab == gt).float().mean()),
           "corridor_fraction_gt": float((gt == 1).float().mean()), "corridor_fraction_pred": float((lab == 1).float().mean())}
    out["chain_segnet_labels"] = chain_eval(torch, lab, pb, mb, placements, R, gen)
    out["chain_label_masks"] = chain_eval(torch, gt, pb, mb, placements, R, gen)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--R", type=int, default=256)
    ap.add_argument("--minutes", type=float, default=8.0)
    ap.add_argument("--max-iters", type=int, default=20000, help="the schedule's horizon (poly decay runs to here)")
    ap.add_argument("--batch-paths", type=int, default=4)
    ap.add_argument("--batch-placements", type=int, default=4)
    ap.add_argument("--lr", type=float, default=0.08)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--eval-every", type=int, default=1000)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from ppnet_amd import edage, train
    from ppnet_amd.gennet import AEViT, load_trained
    from ppnet_amd.segnet import SegNet
    dev = torch.device("cuda:0")
    R = args.R
    torch.manual_seed(5)
    net = SegNet().to(dev)                                            # DiNAT-B + SETR-UP, the configuration bench.py times
    trainer = train.segnet_trainer(net, dev)
    opt = train.segnet_optimizer(trainer, lr=args.lr)
    gen = AEViT(1, 1, R, 24).eval()
    assert load_trained(gen, R), "train GenNet first (tools/train_gennet.py)"
    gen.prepare_inference()
    gen.to(dev).to(torch.bfloat16)
    batch = args.batch_paths * args.batch_placements
    lines = []

    def emit(rec):
        line = json.dumps(rec)
        print(line, flush=True)
        lines.append(line)
        if args.out:
            os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
            with open(args.out, "w") as f:
                f.write(__doc__.split("\n\n")[0] + "\n\n" + "\n".join(lines) + "\n")

    emit({"config": "DiNAT-B + SETR-UP from scratch, float32", "R": R, "batch": batch, "lr": args.lr, "warmup_iters": args.warmup,
          "schedule_horizon": args.max_iters, "parameters": sum(p.numel() for p in net.parameters()), "minutes": args.minutes})
    r0 = evaluate(torch, net, dev, R, gen)
    emit({"step": 0, **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r0.items()}})
    t0 = time.time()
    it, loss_acc, n_acc = 0, 0.0, 0
    while time.time() - t0 < args.minutes * 60 and it < args.max_iters:
        pb = edage.generate_paths(args.batch_paths, R, 50.0, 3.0, seed=777, first_path_id=it * args.batch_paths, device=dev)
        mb = edage.generate_maps(pb, args.batch_placements, 5.0, 20, seed=777, first_map_id=it * batch)
        grid, space, _ = train.generator_pairs(pb, mb, args.batch_placements)
        loss = train.segnet_train_step(trainer, opt, it, args.max_iters, grid, space, schedule=dict(warmup_iters=args.warmup))
        it += 1
        if it % 50 == 0:
            loss_acc += float(loss); n_acc += 1
        if it % args.eval_every == 0:
            r = evaluate(torch, net, dev, R, gen)
            emit({"step": it, "wall_s": round(time.time() - t0, 1), "train_loss": round(loss_acc / max(n_acc, 1), 4),
                  **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}})
            loss_acc, n_acc = 0.0, 0
    r = evaluate(torch, net, dev, R, gen)
    emit({"step": it, "final": True, "wall_s": round(time.time() - t0, 1), "images_seen": it * batch, "images_per_s": round(it * batch / (time.time() - t0), 1),
          **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}})


if __name__ == "__main__":
    main()
