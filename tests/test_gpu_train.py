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


def test_segnet_training_with_the_auxiliary_head():
    """The reference's NAT training config (configs/nat/setr_up_nat_base.py:39-42 over _base_/models/nat.py:22-35) adds an FCN
    auxiliary head on level 2 with loss weight 0.4: the step's loss is decode + 0.4 x aux, level 2's output norm stays live, the
    auxiliary head sits in the 10x 'head' parameter group, and every parameter receives a gradient."""
    from ppnet_amd import train
    from ppnet_amd.segnet import SegNet
    import torch.nn.functional as F
    grid, space, path = _pairs(128, 2, 2, seed=7)
    torch.manual_seed(2)
    aux = dict(type="FCNHead", in_channels=128, in_index=2, channels=32, num_convs=1, concat_input=False, dropout_ratio=0.1,
               num_classes=2, align_corners=False, loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=0.4))
    net = SegNet(**TINY_SEG, auxiliary_head=aux).cuda()
    assert net.backbone.compute_indices == (2, 3)
    trainer = train.segnet_trainer(net)
    opt = train.segnet_optimizer(trainer, lr=0.02)
    head_group = {id(p) for p in opt.param_groups[1]["params"]}
    assert all(id(p) in head_group for p in net.auxiliary_head.parameters()) and opt.param_groups[1]["lr"] == pytest.approx(0.2)
    l0 = float(train.segnet_train_step(trainer, opt, 0, 40, grid, space, schedule=dict(warmup_iters=0)))
    missing = [n for n, p in net.named_parameters() if p.requires_grad and p.grad is None]
    assert not missing, missing
    frozen = sorted(n for n, p in net.named_parameters() if not p.requires_grad)
    assert frozen == sorted(f"backbone.norm{i}.{w}" for i in (0, 1) for w in ("weight", "bias")), frozen
    # the loss is the weighted sum of the two heads' cross-entropies (same dropout masks: eval mode for the check)
    net.eval()
    with torch.no_grad():
        from ppnet_amd import fused
        from ppnet_amd.segnet import IMG_MEAN, IMG_STD
        img = fused.grid_to_image(grid, IMG_MEAN, IMG_STD, torch.float32)
        d = net.forward_train(img, None, space)
        feats = net.backbone(img)
        ce = lambda head: F.cross_entropy(F.interpolate(head(feats).float(), space.shape[-2:], mode="bilinear", align_corners=False), space.long())
        assert float(d["decode.loss_ce"]) == pytest.approx(float(ce(net.decode_head)), rel=1e-4)
        assert float(d["aux.loss_ce"]) == pytest.approx(0.4 * float(ce(net.auxiliary_head)), rel=1e-4)
    assert l0 > 0
    losses = [float(train.segnet_train_step(trainer, opt, it, 40, grid, space, schedule=dict(warmup_iters=0))) for it in range(1, 10)]
    assert losses[-1] < l0, (l0, losses)


def test_prepared_networks_refuse_to_train_and_caches_follow_the_weights():
    """(1) PPNet's prepared networks run forward-only kernels on folded weights: a training step on them raises a clear error.
    (2) Packed-weight caches are keyed on the parameters' version / storage: a bf16 eval forward, an in-place weight update (what
    optimizer.step / load_state_dict do), a second eval forward — the MFMA paths (tokenizer codes, downsampler, head) agree with
    the library paths (PPNET_LIBRARY_CONV / PPNET_LIBRARY_TOKENIZER) on the NEW weights."""
    from ppnet_amd import train
    from ppnet_amd.gennet import AEViT
    from ppnet_amd.segnet import DINAT_BASE, SegNet
    gen = AEViT(1, 1, img_resolution=64, dim=24).cuda().eval().prepare_inference()
    with pytest.raises(RuntimeError, match="prepared inference form"):
        train.gennet_train_step(gen, None, None, torch.zeros(2, 64, 64, device="cuda"), torch.zeros(2, 64, 64, device="cuda"))
    seg = SegNet(**TINY_SEG).cuda().eval().prepare_inference()
    with pytest.raises(RuntimeError, match="prepared inference form"):
        train.segnet_trainer(seg)
    # caches: DiNAT-B's own widths (the MFMA kernels serve 64 -> 128 tokenizer, C % 64 downsamplers, 512-channel head) at a small size
    torch.manual_seed(5)
    cfg = dict(backbone=dict(DINAT_BASE["backbone"], depths=[1, 1, 1, 1], dilations=[[1], [1], [1], [1]]), decode_head=DINAT_BASE["decode_head"])
    net = SegNet(**cfg).cuda().eval().prepare_inference().to(torch.bfloat16)
    g = (torch.rand(2, 64, 64, device="cuda") > 0.3).to(torch.uint8) * 255

    def lowres(library):
        for k in ("PPNET_LIBRARY_CONV", "PPNET_LIBRARY_TOKENIZER"):
            os.environ.pop(k, None)
            if library:
                os.environ[k] = "1"
        try:
            with torch.no_grad():
                from ppnet_amd import fused
                from ppnet_amd.segnet import IMG_MEAN, IMG_STD
                x = fused.grid_to_image(g, IMG_MEAN, IMG_STD, torch.bfloat16) if library else g
                return net.decode_head(net.backbone(x), lowres=True).float()
        finally:
            for k in ("PPNET_LIBRARY_CONV", "PPNET_LIBRARY_TOKENIZER"):
                os.environ.pop(k, None)
    a0, b0 = lowres(False), lowres(True)
    scale = b0.abs().mean().item()
    assert (a0 - b0).abs().mean().item() < 0.05 * scale
    with torch.no_grad():                                                      # an optimizer step's worth of in-place change
        for p in net.parameters():
            p.mul_(1.0 + 0.25 * torch.randn_like(p.float()).to(p.dtype))
    a1, b1 = lowres(False), lowres(True)
    assert (b1 - b0).abs().mean().item() > 0.2 * scale                         # the weights did change the output
    assert (a1 - b1).abs().mean().item() < 0.05 * b1.abs().mean().item()       # and the MFMA paths followed them
