"""Training-step plumbing on the CPU: the two learning-rate schedules against their closed forms, the optimisers' settings, and
the data-parallel gradient exchange on two gloo ranks (the N > 1 path of ppnet_amd/train.py; RCCL replaces gloo on the GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ppnet_amd import train
from ppnet_amd.gennet import AEViT


def test_poly_lr_matches_reference_formula():
    """GenNet/utils/scheduler.py:3-12: lr_i = max(base * (1 - i / max_iters) ** power, min_lr), stepped per iteration."""
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-3)
    sch = train.PolyLR(opt, max_iters=50, power=0.9, min_lr=1e-6)
    seen = []
    for i in range(50):
        seen.append(opt.param_groups[0]["lr"])
        opt.step(); sch.step()
    want = [max(1e-3 * (1 - i / 50) ** 0.9, 1e-6) for i in range(50)]
    assert seen == pytest.approx(want, rel=1e-12)
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-6)                # (1 - 50/50) ** 0.9 = 0 -> the floor


def test_mmseg_poly_warmup_schedule():
    """setr_up_nat_base.py:50-56: poly power 1, min 0, linear warm-up over 1500 iterations from ratio 1e-6."""
    base, T = 0.08, 160000
    assert train.mmseg_poly_lr(base, 0, T) == pytest.approx(base * 1e-6)
    assert train.mmseg_poly_lr(base, 750, T) == pytest.approx(base * (1 - 750 / T) * (1 - 0.5 * (1 - 1e-6)))
    assert train.mmseg_poly_lr(base, 1500, T) == pytest.approx(base * (1 - 1500 / T))
    assert train.mmseg_poly_lr(base, T, T) == 0.0
    lrs = [train.mmseg_poly_lr(base, i, T) for i in range(0, T, 997)]
    assert max(lrs) < base and all(a >= b for a, b in zip(lrs[2:], lrs[3:]))   # monotone after the warm-up


def test_optimizer_settings():
    net = AEViT(1, 1, img_resolution=56, dim=24)
    opt = train.gennet_optimizer(net)
    g = opt.param_groups[0]
    assert isinstance(opt, torch.optim.AdamW) and g["lr"] == 1e-3 and g["betas"] == (0.0, 0.99) and g["eps"] == 1e-8 and g["weight_decay"] == 0

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = torch.nn.Linear(2, 2)
            self.decode_head = torch.nn.Linear(2, 2)
    opt = train.segnet_optimizer(Toy())
    assert isinstance(opt, torch.optim.SGD) and [g["lr"] for g in opt.param_groups] == [0.08, 0.8]
    assert all(g["momentum"] == 0.9 and g["weight_decay"] == 0.0 for g in opt.param_groups)
    train.segnet_set_lr(opt, 1500, 160000)
    assert [g["lr"] for g in opt.param_groups] == pytest.approx([0.08 * (1 - 1500 / 160000), 0.8 * (1 - 1500 / 160000)])


def _tiny_gennet():
    torch.manual_seed(0)
    net = AEViT(1, 1, img_resolution=56, dim=24)
    for blk in net.vit_blocks:
        blk.drop_path_rate = 0.0                      # deterministic: the comparison below is exact arithmetic, not statistics
    return net


def _batch():
    g = torch.Generator().manual_seed(3)
    space = (torch.rand(4, 56, 56, generator=g) < 0.4).to(torch.uint8)
    path = ((torch.rand(4, 56, 56, generator=g) < 0.05).to(torch.uint8) * 255)
    return space, path


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = train.data_parallel(_tiny_gennet(), bucket_cap_mb=1)
    assert isinstance(net, torch.nn.parallel.DistributedDataParallel)
    opt = train.gennet_optimizer(net)
    sch = train.PolyLR(opt, max_iters=10)
    space, path = _batch()
    lo, hi = rank * 2, rank * 2 + 2
    loss = train.gennet_train_step(net, opt, sch, space[lo:hi], path[lo:hi])
    torch.save({"loss": loss, "params": [p.detach().clone() for p in net.module.parameters()]}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_step_equals_averaged_gradients(tmp_path):
    """Two data-parallel ranks, two samples each, one AdamW step: the parameters afterwards are identical on both ranks and equal
    a single process stepping on the MEAN of the two half-batch gradients (what MMDistributedDataParallel computes; BatchNorm
    statistics stay per rank, as with the reference's broadcast_buffers=False)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in (0, 1))
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    net = _tiny_gennet()
    opt = train.gennet_optimizer(net)
    space, path = _batch()
    grads, losses = None, []
    for lo in (0, 2):
        net.train()
        out = net(space[lo:lo + 2].float().unsqueeze(1))
        loss = torch.nn.functional.mse_loss(out.squeeze(1), path[lo:lo + 2].float() / 255.0)
        gs = torch.autograd.grad(loss, list(net.parameters()))
        grads = gs if grads is None else [g0 + g1 for g0, g1 in zip(grads, gs)]
        losses.append(loss.detach())
    for p, g in zip(net.parameters(), grads):
        p.grad = g / 2
    opt.step()
    assert float(r0["loss"]) == pytest.approx(float(losses[0]), rel=1e-6) and float(r1["loss"]) == pytest.approx(float(losses[1]), rel=1e-6)
    for p, q in zip(net.parameters(), r0["params"]):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), float((p - q).abs().max())


def test_assert_trainable_names_prepared_modules():
    """train.assert_trainable: fresh modules pass, prepared / folded ones raise (they run forward-only kernels on folded weights)."""
    from ppnet_amd import train
    from ppnet_amd.gennet import AEViT
    from ppnet_amd.segnet import SegNet
    tiny = dict(backbone=dict(embed_dim=16, mlp_ratio=2.0, depths=[1, 1, 1, 1], num_heads=[1, 1, 2, 4], kernel_size=7, layer_scale=1e-5),
                decode_head=dict(in_channels=128, channels=16, num_convs=2, up_scale=2, num_classes=2))
    train.assert_trainable(AEViT(1, 1, img_resolution=64, dim=24))
    train.assert_trainable(SegNet(**tiny))
    with pytest.raises(RuntimeError):
        train.assert_trainable(AEViT(1, 1, img_resolution=64, dim=24).eval().prepare_inference())
    with pytest.raises(RuntimeError):
        train.assert_trainable(SegNet(**tiny).eval().prepare_inference())


def test_decode_losses_follow_mmseg_with_ignored_pixels():
    """BaseDecodeHead.losses with ignore (255) labels present: the cross entropy is averaged over ALL pixels (ignored ones add 0
    to the sum), accuracy divides by target.numel() and has no ignore index (decode_head.py:231-265,
    losses/cross_entropy_loss.py:20-31, losses/utils.py:66-68, losses/accuracy.py:39-49) — restated here op by op."""
    from ppnet_amd.segnet import decode_losses
    g = torch.Generator().manual_seed(3)
    logit = torch.randn(2, 2, 9, 11, generator=g, dtype=torch.float64)
    gt = torch.randint(0, 2, (2, 9, 11), generator=g)
    gt[0, :3] = 255
    gt[1, 4:, 5:] = 255
    # the reference's formulae, element by element
    logp = torch.log_softmax(logit, dim=1)
    per_px = torch.zeros(2, 9, 11, dtype=torch.float64)
    correct = 0
    for b in range(2):
        for i in range(9):
            for j in range(11):
                t = int(gt[b, i, j])
                if t != 255:
                    per_px[b, i, j] = -logp[b, t, i, j]
                correct += int(int(logit[b, :, i, j].argmax()) == t)
    want_loss = 0.4 * per_px.sum() / per_px.numel()
    want_acc = 100.0 * correct / gt.numel()
    loss, acc = decode_losses(logit, gt, 0.4)
    assert float(loss) == pytest.approx(float(want_loss), rel=1e-12)
    assert float(acc) == pytest.approx(want_acc, rel=1e-6)
    # without ignored pixels both reduce to the plain mean / plain accuracy
    gt2 = gt.clamp(max=1)
    loss2, _ = decode_losses(logit, gt2)
    assert float(loss2) == pytest.approx(float(torch.nn.functional.cross_entropy(logit, gt2)), rel=1e-12)


def test_weight_cache_invalidation():
    """WeightCache follows optimizer-style in-place updates by itself; a write through `.data` needs invalidate_caches()."""
    from ppnet_amd import fused
    p = torch.nn.Parameter(torch.ones(4))
    cache, builds = fused.WeightCache(), []

    def build():
        builds.append(1)
        return p.detach() * 2

    assert torch.equal(cache.get((p,), build), torch.full((4,), 2.0)) and len(builds) == 1
    cache.get((p,), build)
    assert len(builds) == 1                                         # unchanged sources: no rebuild
    with torch.no_grad():
        p.mul_(3)
    assert torch.equal(cache.get((p,), build), torch.full((4,), 6.0)) and len(builds) == 2
    p.data.mul_(2)                                                  # invisible to the key
    stale = cache.get((p,), build)
    if len(builds) == 2:
        assert torch.equal(stale, torch.full((4,), 6.0))
        fused.invalidate_caches()
    assert torch.equal(cache.get((p,), build), torch.full((4,), 12.0))
    cache.clear()
    n = len(builds)
    cache.get((p,), build)
    assert len(builds) == n + 1


def test_head_in_index_selects_backbone_output_slot():
    """A head's in_index is a position in the backbone's OUTPUT LIST (nat.py:326-332 returns one entry per out_index), so
    with out_indices=(1, 2, 3) and in_index=0 the level to compute is 1."""
    from ppnet_amd.segnet import SegNet
    cfg = dict(backbone=dict(type="NAT", embed_dim=32, mlp_ratio=2.0, depths=[1, 1, 1, 1], num_heads=[1, 2, 4, 8], kernel_size=7,
                             out_indices=(1, 2, 3)),
               decode_head=dict(type="SETRUPHead", in_channels=64, channels=16, in_index=0, num_classes=2, num_convs=1, up_scale=2,
                                kernel_size=3, norm_cfg=dict(type="BN")),
               test_cfg=dict(mode="whole"))
    net = SegNet.from_config(cfg)
    assert net.backbone.compute_indices == (1,)
    cfg["decode_head"]["in_index"] = 5
    with pytest.raises(ValueError):
        SegNet.from_config(cfg)
