"""ppn_nat_gemm_bf16 (csrc/nat_gemm.hip): the NAT projections of DiNAT levels 1-3 with LayerNorm / residual / row statistics in the
GEMM epilogues, against float64 compositions of the reference's ops on the same bfloat16 operands (SegNet/nat.py:62-85,140-153:
norm -> qkv / fc1 (+ GELU); x + proj(...) / x + fc2(...)).  Tolerance: one bfloat16 rounding of the result (2^-8 relative) plus the
float32 accumulation — stated per assertion."""
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

SHAPES = [  # (M, N, K): one tile per workgroup ... several tiles per workgroup (the persistent tile boundary), K from 3 k-tiles up
    (512, 768, 256), (512, 256, 256), (768, 512, 192), (512, 1536, 512), (1024, 1024, 1024), (256, 3072, 1024), (16384, 1536, 512),
    (36864, 768, 256)]


def _ops(M, N, K, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = (torch.randn(M, K, device="cuda", generator=g) * 1.3 + 0.4).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g) * 0.5
    return a, w, b


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("gelu", [False, True])
def test_ln_folded_projection_vs_float64(M, N, K, gelu):
    """mode 0 / 1: out = [gelu](LN(a) W0^T + b0) from the raw rows of a.  The float64 reference normalises the bfloat16 rows and
    multiplies by the SAME folded bfloat16 weight the kernel reads (W0 diag(gamma) rounded once), so the comparison isolates the
    kernel: rstd (a w^T - mean colsum) + bias in float32 against the definition."""
    from ppnet_amd import fused
    a, w0, b0 = _ops(M, N, K, 1)
    g = torch.Generator(device="cuda").manual_seed(2)
    gamma = 1.0 + 0.2 * torch.randn(K, device="cuda", generator=g)
    beta = 0.1 * torch.randn(K, device="cuda", generator=g)
    w = (w0.float() * gamma).to(torch.bfloat16).contiguous()
    bias = (b0 + w0.float() @ beta).contiguous()
    colsum = w.float().sum(1).contiguous()
    stats = fused.row_stats(a)
    # row_stats itself: float32 sums of the bfloat16 values
    ad = a.double()
    assert torch.allclose(stats[0, :, 0].double(), ad.sum(1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(stats[0, :, 1].double(), (ad * ad).sum(1), rtol=1e-5, atol=1e-3)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    fused.nat_gemm(a, w, bias, "ln_gelu" if gelu else "ln", out, colsum=colsum, stats_in=stats, eps=1e-5)
    mean = ad.mean(1, keepdim=True)
    var = ad.var(1, unbiased=False, keepdim=True)
    ref = ((ad - mean) / torch.sqrt(var + 1e-5)) @ w.double().t() + bias.double()
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    err = (out.double() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 4e-3                      # one bf16 rounding + accumulation / the logistic erf fit (3e-5)
    assert bool((err <= tol).all()), float((err - tol).max())
    assert float(err.mean()) < 2.5e-3 * float(ref.abs().mean() + 0.05)
    # bit-reproducible
    out2 = torch.empty_like(out)
    fused.nat_gemm(a, w, bias, "ln_gelu" if gelu else "ln", out2, colsum=colsum, stats_in=stats, eps=1e-5)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("M,N,K", [(512, 256, 256), (512, 256, 512), (512, 512, 128), (1024, 1024, 2048), (16384, 512, 512), (70400 // 256 * 256, 256, 256),
                                   # round 5 (the old s read in the epilogue of the 256 x 256 core): DiNAT-B's own level-2 / level-3 shapes at a
                                   # quarter batch, and more tiles than compute units with an odd tile count per workgroup
                                   (16384, 512, 1024), (4096, 1024, 1024), (4096, 1024, 2048), (256 * 259, 512, 256), (768, 768, 192)])
def test_accumulating_projection_and_row_partials(M, N, K):
    """mode 2: s += a W^T + b in place; stats_out[t] = (sum, sum of squares) of the new bfloat16 rows over tile column t (128 columns
    for streams of width <= 512, else 256).  The old s is read in the epilogue and added in float32 (round 3-4: through the matrix
    pipe, times an identity — equally exact), so the only roundings are the float32 accumulation and the final bfloat16 one."""
    from ppnet_amd import fused
    a, w, b = _ops(M, N, K, 3)
    g = torch.Generator(device="cuda").manual_seed(4)
    s0 = (torch.randn(M, N, device="cuda", generator=g) * 2.0).to(torch.bfloat16)
    s = s0.clone()
    P = fused.nat_partials(N)                                  # one partial per 128 columns (N <= 512) or per 256
    st = torch.full((P, M, 2), float("nan"), dtype=torch.float32, device="cuda")
    fused.nat_gemm(a, w, b, "acc", s, stats_out=st)
    ref = s0.double() + a.double() @ w.double().t() + b.double()
    err = (s.double() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 2e-3
    assert bool((err <= tol).all()), float((err - tol).max())
    sd = s.double().view(M, P, N // P)
    assert torch.allclose(st[:, :, 0].double().t(), sd.sum(2), rtol=1e-5, atol=2e-3)
    assert torch.allclose(st[:, :, 1].double().t(), (sd * sd).sum(2), rtol=1e-5, atol=2e-3)
    s2 = s0.clone()
    st2 = torch.empty_like(st)
    fused.nat_gemm(a, w, b, "acc", s2, stats_out=st2)
    assert torch.equal(s, s2) and torch.equal(st, st2)        # no atomics: bit-reproducible


def test_chain_equals_layernorm_then_projection():
    """acc -> ln: the row partials the accumulating GEMM leaves are what the next LN-folded GEMM needs (P = N / 256 partials):
    the chain equals LayerNorm of the updated stream followed by the projection."""
    from ppnet_amd import fused
    M, C = 1024, 512
    a, wp, bp = _ops(M, C, C, 5)
    _, w1, b1 = _ops(M, 2 * C, C, 6)
    g = torch.Generator(device="cuda").manual_seed(7)
    s = torch.randn(M, C, device="cuda", generator=g).to(torch.bfloat16)
    st = torch.empty(fused.nat_partials(C), M, 2, dtype=torch.float32, device="cuda")
    fused.nat_gemm(a, wp, bp, "acc", s, stats_out=st)
    out = torch.empty(M, 2 * C, dtype=torch.bfloat16, device="cuda")
    fused.nat_gemm(s, w1, b1, "ln", out, colsum=w1.float().sum(1).contiguous(), stats_in=st, eps=1e-5)
    ref = torch.nn.functional.layer_norm(s.double(), (C,), None, None, 1e-5) @ w1.double().t() + b1.double()
    err = (out.double() - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 4e-3).all())


def test_gelu_logistic_fit_of_erf():
    """The GELU of mode 1 is erf-GELU evaluated as x * sigmoid(x (p0 + p1 x^2 + p2 x^4)): the fit's error over the reals, in float64
    with the kernel's coefficients (csrc/nat_gemm.hip gelu_logistic), stays below 3.1e-5 — 1 / 60 of a bfloat16 ulp at 1."""
    import math
    x = torch.linspace(-12, 12, 480001, dtype=torch.float64)
    x2 = torch.clamp(x * x, max=64.0)
    t = x * (2.3009787 + x2 * (0.10690469 - 1.0350827e-3 * x2))
    fit = x / (1.0 + torch.exp2(-t))
    assert float((fit - torch.nn.functional.gelu(x)).abs().max()) < 3.1e-5
    assert math.isclose(2.3009787 / math.log2(math.e), 1.59491694, rel_tol=1e-6)


def test_argument_checks():
    from ppnet_amd import _lib
    L = _lib.lib
    assert L.ppn_nat_gemm_bf16(None, None, None, None, None, 0, None, None, 256, 256, 256, 0, 1e-5, None) == -1
    assert L.ppn_row_stats_bf16(None, 4, 256, None, None) == -1


@pytest.mark.parametrize("M,hidden", [(128, 512), (1280, 512), (256, 64), (36864, 512), (128 * 257, 768)])
def test_fused_mlp_vs_float64(M, hidden):
    """ppn_nat_mlp_bf16 (csrc/nat_mlp.hip): s += gelu(LN(s) W1^T + b1) W2^T + b2 in one kernel, C = 256, against the float64
    definition on the same bfloat16 operands (LayerNorm of the bfloat16 rows, the folded bfloat16 W1 the kernel reads, erf-GELU,
    the hidden activation NOT rounded: the kernel rounds it to bfloat16 between the products, which the tolerance carries).
    One workgroup pass (M = 128), several row blocks per workgroup and a partial last round (257 blocks on 256 CUs), a hidden width
    of two chunks; row partials per 128 columns; bit-reproducible."""
    from ppnet_amd import fused
    C = 256
    g = torch.Generator(device="cuda").manual_seed(11)
    s0 = (torch.randn(M, C, device="cuda", generator=g) * 1.3 + 0.4).to(torch.bfloat16)
    w1 = (torch.randn(hidden, C, device="cuda", generator=g) * 0.06).to(torch.bfloat16)
    b1 = torch.randn(hidden, device="cuda", generator=g) * 0.3
    gamma = 1.0 + 0.2 * torch.randn(C, device="cuda", generator=g)
    beta = 0.1 * torch.randn(C, device="cuda", generator=g)
    w2 = (torch.randn(C, hidden, device="cuda", generator=g) * 0.06).to(torch.bfloat16)
    b2 = (torch.randn(C, device="cuda", generator=g) * 0.3).contiguous()
    w1f = (w1.float() * gamma).to(torch.bfloat16).contiguous()
    b1f = b1 + w1.float() @ beta
    hb = torch.stack([w1f.float().sum(1), b1f], dim=1).contiguous()
    assert fused.nat_mlp_ok(M, C, hidden) and not fused.nat_mlp_ok(M + 64, C, hidden) and not fused.nat_mlp_ok(M, 512, hidden)
    wpk = fused.nat_mlp_pack(w1f, w2)
    assert torch.equal(wpk.float().sort().values, torch.cat([w1f.flatten(), w2.flatten()]).float().sort().values)   # a permutation of the weights
    s = s0.clone()
    st = torch.full((2, M, 2), float("nan"), dtype=torch.float32, device="cuda")
    fused.nat_mlp_(s, wpk, hb, b2, hidden, stats_out=st, eps=1e-5)
    sd = s0.double()
    xn = torch.nn.functional.layer_norm(sd, (C,), None, None, 1e-5)
    h = torch.nn.functional.gelu(xn @ w1f.double().t() + b1f.double())
    ref = sd + h @ w2.double().t() + b2.double()
    err = (s.double() - ref).abs()
    # one bf16 rounding of the result, one of every hidden value (2^-9 relative each, sqrt(hidden) of them at |w2| ~ 0.06), float32 sums
    tol = 2.0 ** -8 * ref.abs() + 2.0 ** -8 * 0.06 * (h.abs().pow(2).sum(1, keepdim=True).sqrt()) + 6e-3
    assert bool((err <= tol).all()), float((err - tol).max())
    assert float(err.mean()) < 3e-3 * float(ref.abs().mean() + 0.05)
    # the second output: (sum, sum of squares) of the stored rows per 128 columns
    so = s.double().view(M, 2, 128)
    assert torch.allclose(st[:, :, 0].double().t(), so.sum(2), rtol=1e-5, atol=2e-3)
    assert torch.allclose(st[:, :, 1].double().t(), (so * so).sum(2), rtol=1e-5, atol=2e-3)
    # against the two-kernel form it replaces (same folded operands, hidden activation through HBM)
    s_two = s0.clone()
    hbuf = torch.empty(M, hidden, dtype=torch.bfloat16, device="cuda")
    if M % 256 == 0 and hidden % 256 == 0:
        fused.nat_gemm(s_two, w1f, b1f.contiguous(), "ln_gelu", hbuf, colsum=hb[:, 0].contiguous(), stats_in=fused.row_stats(s_two), eps=1e-5)
        st2 = torch.empty(fused.nat_partials(C), M, 2, dtype=torch.float32, device="cuda")
        fused.nat_gemm(hbuf, w2, b2, "acc", s_two, stats_out=st2)
        d = (s.double() - s_two.double()).abs()
        assert bool((d <= 2.0 ** -7 * ref.abs() + 8e-3).all())
    s3 = s0.clone()
    st3 = torch.empty_like(st)
    fused.nat_mlp_(s3, wpk, hb, b2, hidden, stats_out=st3, eps=1e-5)
    assert torch.equal(s, s3) and torch.equal(st, st3)          # no atomics: bit-reproducible
    fused.nat_mlp_(s3, wpk, hb, b2, hidden, stats_out=None)      # the statistics are optional
