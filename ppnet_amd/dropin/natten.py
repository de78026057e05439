"""`natten` as SegNet/nat.py:14 and SegNet/dinats.py:17 import it (`from natten import NeighborhoodAttention2D`).
With `ppnet_amd/dropin` first on sys.path the reference's backbone files construct the HIP-backed module: the
constructor arguments of nat.py:111-120 (dim, kernel_size, dilation, num_heads, qkv_bias, qk_scale, attn_drop,
proj_drop), forward([B, H, W, C]) -> [B, H, W, C], state-dict keys qkv.* / rpb / proj.*."""
from ppnet_amd.na import NeighborhoodAttention2D  # noqa: F401

NeighborhoodAttention = NeighborhoodAttention2D   # the alias nat.py:14 gives it

__all__ = ["NeighborhoodAttention2D", "NeighborhoodAttention"]
