"""Training steps on the GPU: GenNet and SegNet fed by the generator kernels, the neighbourhood attention's
HIP backward inside a real optimisation step, and DistributedDataParallel over RCCL on a group of one.
Reference: GenNet/train.py:93-147, GenNet/utils/train_and_eval.py:24-46, SegNet/mmseg/apis/train.py:67-167,
SegNet/configs/nat/setr_up_nat_base.py:46-56."""
import datetime
import os
import socket

import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

TINY_SEG = dict(
    backbone=dict(embed_dim=32, mlp_ratio=2.0, depths=[1, 1, 2, 1], num_heads=[1, 2, 4, 8], kernel_size=7, layer_scale=1e-1,
                  dilations=[[1], [2], [1, 2], [1]], drop_path_rate=0.1),
    decode_head=dict(in_channels=256, channels=32, num_convs=4, up_scale=2, num_classes=2, kernel_size=3))


def _pairs(R, n_paths, placements, seed):
    from ppnet_amd import edage, train
    dev = torch.device("cuda:0")
    pb = edage.generate_paths(n_paths, R, 50, 3, seed=seed, device=dev)
    mb = edage.generate_maps(pb, placements, 5, 20, seed=seed)
    return train.generator_pairs(pb, mb, placements)


def test_gennet_training_steps_on_generator_pairs():
    """AdamW + MSE + PolyLR on (mask_space, mask_path) pairs that never leave the device: the loss on the training batch falls
    (a fixed batch of 8 maps, 25 steps: below a quarter of the starting loss), all gradients are finite, and the scheduler follows
    the reference's per-iteration poly curve."""
    from ppnet_amd import train
    from ppnet_amd.gennet import AEViT
    grid, space, path = _pairs(64, 2, 4, seed=2)
    assert space.dtype == torch.uint8 and set(space.unique().tolist()) <= {0, 1} and set(path.unique().tolist()) <= {0, 255}
    torch.manual_seed(0)
    net = AEViT(1, 1, img_resolution=64, dim=24).cuda()
    opt = train.gennet_optimizer(net)
    sch = train.PolyLR(opt, max_iters=25)
    losses = [float(train.gennet_train_step(net, opt, sch, space, path)) for _ in range(25)]
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    assert sum(p.grad is not None for p in net.parameters()) == sum(1 for _ in net.parameters())
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-6)                 # the schedule's floor after max_iters steps
    # train-mode losses (BatchNorm on batch statistics, as train_one_epoch reports them); the eval-mode loss uses the running
    # statistics, which 25 steps at momentum 0.1 have not converged — it only has to be finite
    assert losses[-1] < 0.25 * losses[0], (losses[:3], losses[-3:])
    assert torch.isfinite(train.evaluate_gennet(net, space, path))


def test_segnet_training_steps_use_the_na_backward_kernel():
    """SGD + cross-entropy on (rendered map, mask_space) pairs for a small DiNAT + SETR-UP (dilated and padded levels
    included): the gradient reaches every parameter — in particular each level's relative position bias, which only the
    hand-written backward (ppn_na2d_bwd) produces — and 12 steps lower the loss on the training batch."""
    from ppnet_amd import train
    from ppnet_amd.segnet import SegNet
    grid, space, path = _pairs(128, 2, 3, seed=4)
    torch.manual_seed(1)
    net = SegNet(**TINY_SEG).cuda()
    trainer = train.segnet_trainer(net)
    opt = train.segnet_optimizer(trainer, lr=0.02)
    sched = dict(warmup_iters=3, warmup_ratio=0.1)
    losses = []
    for it in range(12):
        losses.append(float(train.segnet_train_step(trainer, opt, it, 40, grid, space, schedule=sched)))
        if it == 0:
            missing = [n for n, p in net.named_parameters() if p.requires_grad and p.grad is None]
            assert not missing, missing
            frozen = sorted(n for n, p in net.named_parameters() if not p.requires_grad)      # norms of levels SETR-UP does not read
            assert frozen == sorted(f"backbone.norm{i}.{w}" for i in (0, 1, 2) for w in ("weight", "bias")), frozen
            rpb = [(n, float(p.grad.abs().sum())) for n, p in net.named_parameters() if n.endswith("rpb")]
            assert len(rpb) == 5 and all(v > 0 for _, v in rpb), rpb
            assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    assert losses[-1] < 0.8 * losses[0], losses
    assert [g["lr"] for g in opt.param_groups] == pytest.approx([train.mmseg_poly_lr(0.02, 11, 40, **sched), train.mmseg_poly_lr(0.2, 11, 40, **sched)])


def test_rccl_group_of_one_data_parallel_step():
    """DistributedDataParallel on backend "nccl" (= RCCL) with one rank: bucket construction, the gradient all-reduce on HBM
    buffers and the optimiser step run as they will on the 8-GPU node; the result equals the unwrapped model's step."""
    import copy
    import torch.distributed as dist
    from ppnet_amd import train
    from ppnet_amd.gennet import AEViT
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    grid, space, path = _pairs(64, 1, 4, seed=6)
    torch.manual_seed(0)
    base = AEViT(1, 1, img_resolution=64, dim=24).cuda()
    for blk in base.vit_blocks:
        blk.drop_path_rate = 0.0
    plain = copy.deepcopy(base)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev,
                                timeout=datetime.timedelta(seconds=180))
    try:
        ddp = train.data_parallel(base, dev, bucket_cap_mb=1, force=True)
        assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
        la = train.gennet_train_step(ddp, train.gennet_optimizer(ddp), None, space, path)
        lb = train.gennet_train_step(plain, train.gennet_optimizer(plain), None, space, path)
        torch.cuda.synchronize()
        assert float(la) == pytest.approx(float(lb), rel=1e-5)
        for p, q in zip(base.parameters(), plain.parameters()):
            assert torch.allclose(p, q, rtol=1e-4, atol=1e-6)
    finally:
        if created:
            dist.destroy_process_group()
