"""TEST INFRASTRUCTURE — definition oracle for 2-D neighbourhood attention (loop B row B3), not product code.

PARITY UNPINNED: the reference calls `natten.NeighborhoodAttention2D` (SegNet/nat.py:14,111-120,144); NATTEN
is a third-party CUDA package (unpinned in requirements.txt:2, 0.14.x era) whose source is not under
/root/reference, is not installed here, and for which the reference holds no fixtures. This file states the
semantics the build implements (as published in the NAT / DiNAT papers and NATTEN 0.14's kernels):

  * tokens x [B,H,W,C]; if H or W < k*d the input is zero-padded bottom/right to k*d BEFORE the qkv
    projection (so padded tokens carry the qkv bias) and the output is cropped back;
  * q *= head_dim**-0.5;
  * along each axis a query at position i belongs to dilation group g = i mod d, whose members are
    g, g+d, ... (n of them); its window is the k consecutive members starting at member index
    clamp(i//d - k//2, 0, n-k)  (a centred window, shifted inward at the borders);
  * logit(query, member t) = q.k_member + rpb[head, (start_i + t_i - i//d) + k-1, (start_j + t_j - j//d) + k-1];
  * softmax over the k*k members, output = sum p * v_member.

Brute force, per query; used only by tests (and cross-checked for d=1 interior queries against an
unfold-based dense formulation in tests/test_oracle_na.py).
"""
import numpy as np


def window(i, L, k, d):
    """(member positions [k], bias indices [k]) of query i on an axis of length L."""
    g, u = i % d, i // d
    n = (L - g + d - 1) // d
    assert n >= k, "axis shorter than kernel*dilation: pad first"
    start = min(max(u - k // 2, 0), n - k)
    t = np.arange(k)
    return g + (start + t) * d, (start + t - u) + (k - 1)


def na2d_core(q, k_, v, rpb, kernel_size, dilation):
    """q, k_, v: [B, heads, H, W, hd] (q already scaled); rpb [heads, 2k-1, 2k-1]. Returns [B, heads, H, W, hd]."""
    B, nh, H, W, hd = q.shape
    K = kernel_size
    out = np.zeros_like(q, dtype=np.float64)
    q = q.astype(np.float64); k_ = k_.astype(np.float64); v = v.astype(np.float64)
    for i in range(H):
        pi, bi = window(i, H, K, dilation)
        for j in range(W):
            pj, bj = window(j, W, K, dilation)
            kk = k_[:, :, pi][:, :, :, pj]                          # [B, nh, K, K, hd]
            vv = v[:, :, pi][:, :, :, pj]
            logit = np.einsum("bhc,bhijc->bhij", q[:, :, i, j], kk) + rpb[None][:, :, bi][:, :, :, bj]
            logit = logit.reshape(B, nh, K * K)
            p = np.exp(logit - logit.max(-1, keepdims=True))
            p /= p.sum(-1, keepdims=True)
            out[:, :, i, j] = np.einsum("bhn,bhnc->bhc", p, vv.reshape(B, nh, K * K, hd))
    return out


def na2d_from_qkv(qkv, rpb, heads, kernel_size, dilation, scale):
    """qkv: [B,H,W,3*C] (the qkv Linear's output). Returns [B,H,W,C] (input of the output projection)."""
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    hd = C // heads
    t = qkv.reshape(B, H, W, 3, heads, hd).transpose(3, 0, 4, 1, 2, 5).astype(np.float64)
    o = na2d_core(t[0] * scale, t[1], t[2], rpb.astype(np.float64), kernel_size, dilation)
    return o.transpose(0, 2, 3, 1, 4).reshape(B, H, W, C)


def neighborhood_attention_2d(x, w_qkv, b_qkv, rpb, w_proj, b_proj, heads, kernel_size=7, dilation=1):
    """Full module forward (pad -> qkv -> NA -> crop -> proj) on x [B,H,W,C], weights in torch Linear layout."""
    B, H, W, C = x.shape
    win = kernel_size * dilation
    pr, pb = max(0, win - W), max(0, win - H)
    xp = np.pad(x, ((0, 0), (0, pb), (0, pr), (0, 0))) if (pr or pb) else x
    qkv = xp @ w_qkv.T + b_qkv
    o = na2d_from_qkv(qkv, rpb, heads, kernel_size, dilation, (C // heads) ** -0.5)
    o = o[:, :H, :W]
    return o @ w_proj.T + b_proj
