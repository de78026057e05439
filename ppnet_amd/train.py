"""Training step of the two networks, data-parallel over RCCL (the rows after the inference path).

Reference: GenNet/train.py:93-147 + GenNet/utils/train_and_eval.py:24-46 (AdamW(lr 1e-3, betas (0, 0.99), eps 1e-8, no weight
decay), MSE between the network output and mask_path, PolyLR stepped every iteration, GenNet/utils/scheduler.py:3-12);
SegNet/mmseg/apis/train.py:67-167 + SegNet/configs/nat/setr_up_nat_base.py:46-56 (SGD lr 0.08, momentum 0.9, no weight decay,
decode-head parameters at 10x the rate, poly schedule with power 1 and a 1500-iteration linear warm-up from ratio 1e-6,
cross-entropy of the decode head's logits resized to the label map plus 0.4 x the auxiliary FCN head's on level 2,
mmseg/models/segmentors/encoder_decoder.py:81-152, decode_heads/decode_head.py:209-237, fcn_head.py; SyncBN in the heads;
MMDistributedDataParallel = gradient all-reduce averaged over the ranks).

What is native here: the training PAIRS come straight from the generator kernels on the device (stage A/B + ppn_label_masks:
mask_space / mask_path / the rendered map — the reference re-reads them from image files, my_dataset.py:30-76), the
neighbourhood attention's backward is the HIP kernel (ppn_na2d_bwd through na.na2d_autograd), and the gradient exchange is
torch's DistributedDataParallel over RCCL with buckets sized for xGMI rings (few large all-reduces).  Every other op of the
backward pass is a ROCm library call through autograd: the fused inference kernels are forward-only and step aside while
autograd is recording (fused.recording).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class PolyLR(torch.optim.lr_scheduler.LRScheduler):
    """lr = max(base_lr * (1 - it / max_iters) ** power, min_lr), stepped per iteration (GenNet/utils/scheduler.py:3-12)."""

    def __init__(self, optimizer, max_iters, power=0.9, last_epoch=-1, min_lr=1e-6):
        self.power, self.max_iters, self.min_lr = power, max_iters, min_lr
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        return [max(base * (1 - self.last_epoch / self.max_iters) ** self.power, self.min_lr) for base in self.base_lrs]


def mmseg_poly_lr(base_lr, it, max_iters, power=1.0, min_lr=0.0, warmup_iters=1500, warmup_ratio=1e-6):
    """mmcv's PolyLrUpdaterHook with linear warm-up (by iteration): regular = (base - min) * (1 - it / max) ** power + min;
    during warm-up lr = regular * (1 - (1 - it / warmup_iters) * (1 - warmup_ratio))
    (SegNet/configs/nat/setr_up_nat_base.py:50-56)."""
    regular = (base_lr - min_lr) * (1 - it / max_iters) ** power + min_lr
    if it < warmup_iters:
        return regular * (1 - (1 - it / warmup_iters) * (1 - warmup_ratio))
    return regular


def gennet_optimizer(model, lr=1e-3):
    """GenNet/train.py:93."""
    return torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=lr, weight_decay=0, betas=(0.0, 0.99), eps=1e-8)   # torch >= 2.x wants both betas as floats


def segnet_optimizer(model, lr=0.08, momentum=0.9, weight_decay=0.0, head_lr_mult=10.0):
    """SGD with the decode head at 10x (paramwise_cfg custom_keys {'head': lr_mult 10}, setr_up_nat_base.py:46-48).  The groups
    carry `base_lr` for segnet_set_lr."""
    head, body = [], []
    for name, p in model.named_parameters():
        if p.requires_grad:
            (head if "head" in name else body).append(p)
    groups = [{"params": body, "lr": lr, "base_lr": lr}, {"params": head, "lr": lr * head_lr_mult, "base_lr": lr * head_lr_mult}]
    return torch.optim.SGD([g for g in groups if g["params"]], lr=lr, momentum=momentum, weight_decay=weight_decay)


def segnet_set_lr(optimizer, it, max_iters, **schedule):
    for g in optimizer.param_groups:
        g["lr"] = mmseg_poly_lr(g.get("base_lr", g["lr"]), it, max_iters, **schedule)


def data_parallel(model, device=None, bucket_cap_mb=128, force=False):
    """DistributedDataParallel over the initialised process group (RCCL on the GPU, gloo in the CPU tests).  xGMI is point to
    point, a ring all-reduce is bound by one link: few large buckets (128 MB) instead of torch's 25 MB default; gradients are
    views of the buckets (no copy before the all-reduce); buffers are not broadcast (the reference passes broadcast_buffers=False,
    mmseg/apis/train.py:97-101)."""
    import torch.distributed as dist
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):     # force: wrap a group of one (tests)
        return model
    ids = [device] if device is not None and torch.device(device).type == "cuda" else None
    return nn.parallel.DistributedDataParallel(model, device_ids=ids, broadcast_buffers=False, bucket_cap_mb=bucket_cap_mb,
                                               gradient_as_bucket_view=True)


def generator_pairs(paths, maps, placements, bound=None):
    """Training tensors of one generated batch, all on the device: (grid u8 [n,R,R] occupancy codes, mask_space u8 {0,1},
    mask_path u8 {0,255}) — what process_map.py:148-191 writes to mask_space/ and mask_path/ and MapGenerate.py:111-126 to map/."""
    from . import edage
    mask_path, mask_space = edage.label_masks(paths, maps, placements, bound=bound, want_path=True, want_space=True)
    return maps.grid, mask_space, mask_path


def assert_trainable(model):
    """The prepared inference forms (AEViT.prepare_inference: _FusedStage; SegNet.prepare_inference / NATBlock.fold: BatchNorm and
    LayerScale folded into the weights, forward-only fused kernels) cannot be trained: say so instead of failing inside autograd
    or dropping gradients silently.  PPNet() always prepares its networks — train fresh AEViT / SegNet modules."""
    from .gennet import _FusedStage
    from .segnet import NATLayer
    m = model.module if isinstance(model, nn.parallel.DistributedDataParallel) else model
    for sub in m.modules():
        if isinstance(sub, _FusedStage) or (isinstance(sub, NATLayer) and sub.folded) or getattr(sub, "prepared", False):
            raise RuntimeError(f"{type(sub).__name__} is in its prepared inference form (forward-only kernels, folded weights): "
                               "build a fresh module (and load the checkpoint) to train")


def gennet_train_step(model, optimizer, scheduler, mask_space, mask_path, amp_dtype=None):
    """One iteration of train_one_epoch (train_and_eval.py:24-46): input = mask_space as {0,1} floats (ToTensor * 255 of the
    palette image, my_dataset.py:8-16), target = mask_path / 255.  Returns the loss (a 0-d tensor, not synchronised)."""
    assert_trainable(model)
    model.train()
    x = mask_space.to(torch.float32).unsqueeze(1)
    target = (mask_path.to(torch.float32) / 255.0)
    with torch.autocast(x.device.type, dtype=amp_dtype, enabled=amp_dtype is not None):
        out = model(x)
        loss = F.mse_loss(out.squeeze(1).float(), target)
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return loss.detach()


class _SegTrain(nn.Module):
    """forward = the training loss, so DistributedDataParallel sees one forward per step: the sum of the `loss_ce` entries of
    SegNet.forward_train (encoder_decoder.py:122-152; mmseg's _parse_losses sums every key containing 'loss', base.py:160-190) —
    the decode head's cross-entropy plus, where the config has one (configs/nat/setr_up_nat_base.py:39-42 over
    _base_/models/nat.py:22-35), 0.4 x the auxiliary FCN head's on level 2."""

    def __init__(self, segnet):
        super().__init__()
        self.net = segnet
        # output norms of levels no head reads are never evaluated (SegNet narrows compute_indices): freeze them, or
        # DistributedDataParallel would wait for gradients that never come
        bb = segnet.backbone
        for i in getattr(bb, "out_indices", ()):
            if i not in bb.compute_indices:
                getattr(bb, f"norm{i}").requires_grad_(False)

    def forward(self, img, labels):
        losses = self.net.forward_train(img, None, labels)
        return sum(v for k, v in losses.items() if "loss" in k)


def segnet_trainer(segnet, device=None, bucket_cap_mb=128, sync_bn=True):
    """The module to call as loss = trainer(img, labels): SegNet wrapped for data-parallel training.  Under an initialised
    process group of more than one rank the heads' BatchNorm layers become SyncBatchNorm first, as the reference's norm_cfg
    `SyncBN` does under MMDistributedDataParallel (configs/_base_/models/nat.py:2, SegNet/train.py:179-185 reverts it to BN
    only for non-distributed runs): batch statistics are all-reduced, so every rank holds the same running statistics."""
    import torch.distributed as dist
    assert_trainable(segnet)
    if sync_bn and dist.is_initialized() and dist.get_world_size() > 1 and (device is None or torch.device(device).type == "cuda"):
        segnet = nn.SyncBatchNorm.convert_sync_batchnorm(segnet)               # (SyncBatchNorm has no CPU / gloo implementation)
    return data_parallel(_SegTrain(segnet), device, bucket_cap_mb)


def segnet_train_step(trainer, optimizer, it, max_iters, grid_u8, mask_space, schedule=None):
    """One iteration: the normalised image from the occupancy codes (ppn_grid_to_image, planning_seg.py:12-41), labels =
    mask_space; lr by the warm-up poly schedule; SGD step.  Returns the loss (0-d tensor)."""
    from . import fused
    from .segnet import IMG_MEAN, IMG_STD
    trainer.train()
    dtype = next(trainer.parameters()).dtype
    img = fused.grid_to_image(grid_u8, IMG_MEAN, IMG_STD, dtype) if grid_u8.dtype == torch.uint8 else grid_u8
    segnet_set_lr(optimizer, it, max_iters, **(schedule or {}))
    loss = trainer(img, mask_space)
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    optimizer.step()
    return loss.detach()


def evaluate_gennet(model, mask_space, mask_path):
    """Mean MSE of the eval-mode network over a batch (train_and_eval.py:8-21)."""
    model.eval()
    with torch.no_grad():
        out = model(mask_space.to(torch.float32).unsqueeze(1))
        return F.mse_loss(out.squeeze(1).float(), mask_path.to(torch.float32) / 255.0)
