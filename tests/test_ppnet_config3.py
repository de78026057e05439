"""BASELINE config 3 — the path bench.py times: PPNet(256) = SegNet.prepare_inference() (BN fold, NATBlock.fold(), head bias
inside the upsample kernel, commuted classifier) in bf16 -> segment_u8 -> AE-ViT -> per-sample min-max -> waypoint walk ->
collision check.  Every stage of THAT object is compared with something independent of it:

  SegNet   prepared-fp32 logits vs a float64 op-by-op composition of the reference's sources (oracle/segnet_ref.py; the
           attention is the definition oracle's window rule — NATTEN itself is absent: parity unpinned);
           prepared-bf16 segment_u8 labels vs the unprepared fp32 module;
  GenNet   heatmap() vs the reference module's golden output at R = 256 (tests/golden/g13_aevit.npz), fp32 and bf16;
  tail     plan_tail() vs oracle/plan_np.py on the same heat maps; evaluate_plans vs a NumPy restatement;
  batch    plan() once at batch 256 with size-independent properties.
"""
import copy
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import na_np as NA
from oracle import plan_np as PN
from oracle import segnet_ref as SR

R = 256


def test_gather_attention_equals_definition_oracle():
    """The float64 gather used by the full-model composition IS the definition oracle (brute force per query)."""
    rng = np.random.RandomState(0)
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    for (H, W, C, heads, d) in [(9, 11, 64, 2, 1), (16, 16, 64, 2, 2), (8, 8, 64, 2, 2), (10, 6, 32, 1, 3)]:
        x = rng.standard_normal((2, H, W, C)); wq = rng.standard_normal((3 * C, C)) * 0.2; bq = rng.standard_normal(3 * C) * 0.3
        rpb = rng.standard_normal((heads, 13, 13)); wp = rng.standard_normal((C, C)) * 0.2; bp = rng.standard_normal(C)
        want = NA.neighborhood_attention_2d(x, wq, bq, rpb, wp, bp, heads, 7, d)
        got = SR.na_fp64(t(x), t(wq), t(bq), t(rpb), t(wp), t(bp), heads, 7, d).numpy()
        assert np.abs(got - want).max() < 1e-12


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def stage_b(dev):
    """8 stage-B grids at 256 x 256 (2 target paths x 4 placements, Philox seed 11) with their labels."""
    from ppnet_amd import edage
    pb = edage.generate_paths(2, R, 50, 3, seed=11, device=dev)
    mb = edage.generate_maps(pb, 4, 5, 20, seed=11)
    torch.cuda.synchronize()
    return pb, mb


@pytest.fixture(scope="module")
def segnet_models(dev):
    """(unprepared fp32 SegNet, PPNet fp32-prepared, PPNet bf16-prepared = what bench.py builds) on the same weights."""
    from ppnet_amd.ppnet import PPNet
    from ppnet_amd.segnet import SegNet
    torch.manual_seed(0)
    m = SR.randomize(SegNet().eval(), seed=1)
    p32 = PPNet(R, segnet=copy.deepcopy(m), weights_dtype=None).to(dev).eval()
    p16 = PPNet(R, segnet=copy.deepcopy(m)).to(dev).eval()                    # bench.py:75
    return m.to(dev), p32, p16


@pytest.mark.gpu
def test_prepared_fp32_logits_vs_fp64_composition(dev, stage_b, segnet_models):
    from ppnet_amd import edage, fused
    from ppnet_amd.segnet import DINAT_BASE, IMG_MEAN, IMG_STD, normalize_images
    m, p32, _ = segnet_models
    grid = stage_b[1].grid
    img = normalize_images(edage.grid_to_rgb(grid) * 255.0)                   # render-then-normalise (planning_seg.py:12-41)
    with torch.no_grad():
        want = SR.segnet_logits_fp64(m, DINAT_BASE, img)
        base = m.encode_decode(img).double()                                  # unprepared product module
        x = fused.grid_to_image(grid, IMG_MEAN, IMG_STD, torch.float32)
        got = p32.segnet.encode_decode(x).double()                            # prepared: fold + BN fold + fused head
    scale = max(1.0, float(want.abs().max()))
    assert want.shape == (8, 2, R, R) and float(want.std()) > 1e-3            # the composition is not degenerate
    assert float((base - want).abs().max()) < 2e-3 * scale                    # float32 GPU libraries vs float64
    assert float((got - want).abs().max()) < 2e-3 * scale
    # and the labels: disagreement only where the float64 margin is inside the float32 error
    margin = (want[:, 1] - want[:, 0]).abs()
    lab64 = want.argmax(dim=1)
    lab = p32.segment_u8(grid).long()
    sure = margin > 4e-3 * scale
    assert bool((lab[sure] == lab64[sure]).all()) and float(sure.float().mean()) > 0.9


@pytest.mark.gpu
def test_bf16_segment_u8_vs_unprepared_fp32(dev, stage_b, segnet_models):
    """The timed object's labels against the unprepared fp32 module.  bf16 (8 significant bits, fp32 accumulation) through
    30 layers moves a logit by a few percent of its scale, so a pixel may flip only where the fp32 margin is that small:
    bar = every pixel whose fp32 margin exceeds 6 x the rms logit error agrees, and >= 97 % of all pixels agree."""
    from ppnet_amd import edage, fused
    from ppnet_amd.segnet import IMG_MEAN, IMG_STD, normalize_images
    m, p32, p16 = segnet_models
    grid = stage_b[1].grid
    with torch.no_grad():
        logits = m.encode_decode(normalize_images(edage.grid_to_rgb(grid) * 255.0))
        lab32 = logits.argmax(dim=1)
        x16 = fused.grid_to_image(grid, IMG_MEAN, IMG_STD, torch.bfloat16)
        logits16 = p16.segnet.encode_decode(x16).float()
        lab16 = p16.segment_u8(grid).long()
    err = (logits16 - logits)
    rms = float(err.pow(2).mean().sqrt())
    rel = rms / float(logits.pow(2).mean().sqrt())
    print(f"bf16 prepared vs fp32 unprepared: rms logit error {rms:.4g} ({rel:.2%} of the logit rms)")
    assert rel < 0.08
    agree = (lab16 == lab32)
    margin = (logits[:, 1] - logits[:, 0]).abs()
    assert float(agree.float().mean()) > 0.97
    assert bool(agree[margin > 6 * rms].all())
    # the fused tail (grid_to_image + labels_u8) gives the labels of segment(), in both precisions
    for p in (p32, p16):
        a, b = p.segment_u8(grid).long(), p.segment(grid)
        assert float((a == b).float().mean()) > 0.999


@pytest.mark.gpu
def test_bf16_parity_criteria_with_balanced_classifier(dev, stage_b):
    """bench.py's `ppnet.parity` on the objects bench.py builds (bench_ppnet: neutral parameters randomised, the untrained
    classifier's bias balanced so that both classes occur): the criteria its `within_tolerance` is made of —
    rms logit error < 1 % of the logit rms, every pixel whose float32 class margin exceeds 6 x that error agrees, GenNet's 8-bit
    heat map on the same mask within 6 codes / 2 codes rms (<= 4 x the measured 0.0025 / 2 / 0.56: bench.PARITY_TOLERANCE).  Overall label agreement is NOT a criterion there: with a balanced
    bias the margin is a small difference of two near-equal logits and pixels inside the band may flip."""
    import bench
    grid = stage_b[1].grid
    p16 = bench.bench_ppnet(torch, dev, R, calibrate=grid[:4])
    p32 = bench.bench_ppnet(torch, dev, R, weights_dtype="f32", calibrate=grid[:4])
    par = bench.ppnet_parity(torch, p16, p32, grid)
    print(par)
    tol = bench.PARITY_TOLERANCE
    assert (tol["rms_logit_rel_max"], tol["heat_u8_max_code_diff_max"], tol["heat_u8_rms_code_diff_max"]) == (0.01, 6, 2.0)
    assert par["within_tolerance"] and par["rms_logit_rel"] < 0.01 and par["labels_agree_where_margin_exceeds_6rms"]
    assert par["heat_u8_max_code_diff"] <= 6 and par["heat_u8_rms_code_diff"] <= 2.0
    assert 0.1 < par["free_fraction_fp32"] < 0.9 and par["pixels_with_such_margin"] > 0.5 and par["label_agreement_vs_fp32"] > 0.9


def _golden_gennet(golden_dir, R_):
    from ppnet_amd.gennet import AEViT
    g = np.load(os.path.join(golden_dir, "g13_aevit.npz"))
    m = AEViT(1, 1, R_, 24).eval()
    m.load_state_dict({k[len(f"R{R_}/w/"):]: torch.tensor(g[k]) for k in g.files if k.startswith(f"R{R_}/w/")}, strict=True)
    return m, torch.tensor(g[f"R{R_}/x"]), torch.tensor(g[f"R{R_}/y"])


@pytest.mark.gpu
@pytest.mark.parametrize("R_", [256, 512])
def test_heatmap_vs_reference_golden(dev, golden_dir, R_):
    """PPNet.heatmap (prepared AE-ViT + per-sample min-max -> u8) on the reference module's golden weights / inputs:
    fp32 within one 8-bit code of the golden output's own normalisation; bf16 (what bench.py runs): <= 3 % of full scale
    rms, every pixel within 16 codes (the heat map is an 8-bit image whose ridge the walk follows)."""
    from ppnet_amd.gennet import normalize_heatmap_u8
    from ppnet_amd.ppnet import PPNet
    from ppnet_amd.segnet import NAT
    gm, x, y = _golden_gennet(golden_dir, R_)
    want = normalize_heatmap_u8(y).to(torch.int32)
    tiny = torch.nn.Module()                                                  # PPNet wants a SegNet: not used here
    tiny.prepare_inference = lambda: tiny
    for dt, tol_max, tol_rms in ((None, 1, 0.5), (torch.bfloat16, 16, 0.03 * 255)):
        p = PPNet(R_, segnet=tiny, gennet=copy.deepcopy(gm), weights_dtype=dt).to(dev).eval()
        got = p.heatmap(x[:, 0].to(dev)).to(torch.int32).cpu()
        d = (got - want).abs()
        rms = float(d.float().pow(2).mean().sqrt())
        print(f"R={R_} dtype={dt}: max code diff {int(d.max())}, rms {rms:.3f}")
        assert int(d.max()) <= tol_max and rms <= tol_rms


def _oracle_tail(heat, init, end, obs, n_obs, clearance, max_wp):
    ok, path = PN.extract_path(heat, init, end, down_sample_rate=2, max_wp=max_wp)
    if not ok:
        return False, None, False
    p32 = path.astype(np.float32)
    hit = any(PN.collision_check_circle_edge(p32[i], p32[i + 1], obs[:n_obs].astype(np.float32), clearance, bound=heat.shape[0])
              for i in range(len(path) - 1))
    return True, path, hit


@pytest.mark.gpu
def test_plan_tail_on_a_side_stream_equals_the_serial_tail(dev, stage_b, segnet_models):
    """PPNet.plan_tail(side_stream=...) (round 5: the walk kernel is as long as its longest walk, so bench.py runs it beside the next
    batch's networks): the same kernels behind an event on another stream — identical results once that stream is synchronised, also
    when the heat map is produced on the current stream right in front of the call."""
    from ppnet_amd import evaluate
    pb, mb = stage_b
    _, _, p16 = segnet_models
    init, end = mb.segpoint[:, 0].contiguous(), mb.segpoint[:, 10].contiguous()
    obs, n_obs = mb.obstacles, mb.n_obstacles[:, 0].contiguous()
    want = p16.plan_tail(evaluate.label_heatmaps(pb, mb, 4), init, end, obs, n_obs)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev)
    for _ in range(3):
        heat = evaluate.label_heatmaps(pb, mb, 4)              # enqueued on the current stream; the tail must wait for it
        got = p16.plan_tail(heat, init, end, obs, n_obs, side_stream=side)
        del heat                                               # (record_stream keeps its memory until the side stream is done)
        side.synchronize()
        for k in ("ok", "counts", "collision", "success", "waypoints"):
            assert torch.equal(got[k], want[k]), k


@pytest.mark.gpu
def test_plan_tail_vs_oracle_on_network_and_ridge_heatmaps(dev, stage_b, segnet_models):
    """plan() = segment_u8 -> heatmap -> plan_tail.  The tail is checked against oracle/plan_np.py fed the SAME heat maps:
    (i) the maps the (random-weight) networks produce — noise, the walk fails or wanders: both sides must agree on that —
    and (ii) ridge maps along the label path (evaluate.label_heatmaps), where plans exist."""
    from ppnet_amd import evaluate
    pb, mb = stage_b
    _, _, p16 = segnet_models
    init, end = mb.segpoint[:, 0].contiguous(), mb.segpoint[:, 10].contiguous()
    obs, n_obs = mb.obstacles, mb.n_obstacles[:, 0].contiguous()
    clearance = R / 50
    heat_net = p16.heatmap(p16.segment_u8(mb.grid))
    heat_ridge = evaluate.label_heatmaps(pb, mb, 4)
    for heat, cap, expect_plans in ((heat_net, 300, False), (heat_ridge, 2048, True)):
        res = p16.plan_tail(heat, init, end, obs, n_obs, max_wp=cap)
        hh, ih, eh = heat.cpu().numpy(), init.cpu().numpy(), end.cpu().numpy()
        oh, nh = obs.cpu().numpy(), n_obs.cpu().numpy()
        n_ok = 0
        for i in range(heat.shape[0]):
            ok, path, hit = _oracle_tail(hh[i], ih[i], eh[i], oh[i], int(nh[i]), clearance, cap)
            assert bool(res["ok"][i]) == ok
            if ok:
                n_ok += 1
                assert int(res["counts"][i]) == len(path)
                assert np.abs(res["waypoints"][i, :len(path)].cpu().numpy() - path).max() < 1e-9
                assert bool(res["collision"][i]) == hit and bool(res["success"][i]) == (not hit)
            else:
                assert not bool(res["success"][i])
        if expect_plans:
            assert n_ok >= 7
            # evaluate_plans against a NumPy restatement of process_map.py:496-503 + the harness criterion
            target = pb.length.repeat_interleave(4) * R / 50
            ev = evaluate.evaluate_plans(res, target, epsilon=0.1)
            wp, cnt = res["waypoints"].cpu().numpy(), res["counts"].cpu().numpy()
            succ = (res["ok"] & ~res["collision"]).cpu().numpy()
            lens = np.array([np.sqrt(((wp[i, 1:cnt[i]] - wp[i, :cnt[i] - 1]) ** 2).sum(1)).sum() if cnt[i] else 0.0 for i in range(len(cnt))])
            ratio = lens / target.cpu().numpy()
            assert abs(ev["success"] - succ.mean()) < 1e-12
            assert abs(ev["length_ratio"] - ratio[succ].mean()) < 1e-9
            assert abs(ev["within_eps"] - (succ & (ratio <= 1.1)).mean()) < 1e-12
            assert 0.8 < ev["length_ratio"] < 1.3                               # a walk along the ridge is about as long as the path


@pytest.mark.gpu
def test_plan_batch_256_properties(dev, segnet_models):
    """Config 3 at its stated size: one plan() over 256 problems at 256 x 256 on the bench's object, checked through
    properties that do not need an oracle run: shapes, per-sample normalisation, chain structure of every returned plan."""
    from ppnet_amd import edage, evaluate
    _, _, p16 = segnet_models
    pb = edage.generate_paths(3, R, 50, 3, seed=5, device=dev)
    mb = edage.generate_maps(pb, 100, 5, 20, seed=5)
    B = 256
    g = mb.grid[:B]
    init, end = mb.segpoint[:B, 0].contiguous(), mb.segpoint[:B, 10].contiguous()
    obs, n_obs = mb.obstacles[:B], mb.n_obstacles[:B, 0].contiguous()
    res = p16.plan(g, init, end, obs, n_obs)
    heat = p16.heatmap(p16.segment_u8(g))
    torch.cuda.synchronize()
    assert heat.shape == (B, R, R) and heat.dtype == torch.uint8
    assert int(heat.reshape(B, -1).max(dim=1).values.min()) == 255 and int(heat.reshape(B, -1).min(dim=1).values.max()) == 0
    assert res["ok"].shape == (B,) and res["collision"].shape == (B,) and res["waypoints"].shape[0] == B
    assert bool((res["success"] == (res["ok"] & ~res["collision"])).all())
    assert bool((res["counts"][~res["ok"]] == 0).all())

    def chain_ok(r):
        wp, cnt, ok = r["waypoints"].cpu().numpy(), r["counts"].cpu().numpy(), r["ok"].cpu().numpy()
        ih, eh = init.cpu().numpy(), end.cpu().numpy()
        for i in np.nonzero(ok)[0]:
            w = wp[i, :cnt[i]]
            assert np.array_equal(w[0], ih[i]) and np.array_equal(w[-1], eh[i])          # [init] + walk + [end]
            steps = np.sqrt(((w[2:-1] - w[1:-2]) ** 2).sum(1))
            assert steps.size == 0 or (steps.min() >= 2.0 - 1e-9 and steps.max() <= 2 * np.sqrt(2) + 1e-9)   # 8-neighbour moves x rate
            assert np.sqrt(((w[-2] - eh[i]) ** 2).sum()) <= 2 * 2.5 + 1e-9            # stop rule, process_map.py:346
    chain_ok(res)
    # the same batch with plans that exist: ridge heat maps along the labels
    ridge = evaluate.label_heatmaps(pb, mb, 100)[:B]
    res2 = p16.plan_tail(ridge, init, end, obs, n_obs)
    chain_ok(res2)
    ev = evaluate.evaluate_plans(res2, (pb.length.repeat_interleave(100) * R / 50)[:B])
    print("ridge heat maps, batch 256:", ev)
    assert ev["extract_ok"] > 0.9 and ev["success"] > 0.5


@pytest.mark.gpu
def test_captured_graph_equals_eager(dev, stage_b, segnet_models):
    """PPNet.capture(): the batch recorded as one HIP graph writes what the eager calls write — same kernels, same buffers —
    bit for bit, and a replay after new inputs were copied into the static buffers follows the new inputs."""
    from ppnet_amd import evaluate
    _, _, p16 = segnet_models
    pb, mb = stage_b
    g = mb.grid.clone()
    init, end = mb.segpoint[:, 0].contiguous(), mb.segpoint[:, 10].contiguous()
    obs, n_obs = mb.obstacles.clone(), mb.n_obstacles[:, 0].contiguous()
    ridge = evaluate.label_heatmaps(pb, mb, 4).contiguous()

    def eager(grid, heat_for_tail):
        mask = p16.segment_u8(grid)
        heat = p16.heatmap(mask)
        return mask, heat, p16.plan_tail(heat_for_tail if heat_for_tail is not None else heat, init, end, obs, n_obs)
    for tail_heat in (None, ridge):
        cp = p16.capture(g, init, end, obs, n_obs, tail_heat=tail_heat)
        for grid in (mb.grid, mb.grid.flip(0).contiguous()):                    # second pass: a different batch through the same graph
            cp.grid.copy_(grid)
            res = cp.replay()
            torch.cuda.synchronize()
            mask, heat, want = eager(grid, tail_heat)
            assert torch.equal(cp.mask, mask) and torch.equal(cp.heat, heat)
            for k in ("ok", "counts", "collision", "success", "waypoints"):
                assert torch.equal(res[k], want[k]), k


@pytest.mark.gpu
def test_two_captures_of_different_batch_sizes_in_one_process(dev, stage_b, segnet_models):
    """Regression cover for round 3's capture fault (HSA memory aperture violation in na2d_halo16_kernel when a second graph, of
    another batch size, was captured in the same process: tile descriptors used to be graph allocation nodes).  Now the
    launchers allocate nothing per launch: descriptor tables are cached per geometry, and a geometry FIRST MET WHILE CAPTURING
    is declined (na2d_halo16_launch returns -1, the per-tile kernel runs — same arithmetic).  Here: a warm capture at batch 4
    (PPNet.capture), then a COLD capture at batch 6 — a batch size this process has never run, recorded without a warm-up pass,
    so its geometries are first met inside the capture —, both graphs replayed alternately, compared with eager runs bit for bit."""
    if any(os.environ.get(k) for k in ("PPNET_LIBRARY_TOKENIZER", "PPNET_TOKENIZER_TWO_KERNELS", "PPNET_LIBRARY_CONV")):
        pytest.skip("A/B knob: a COLD capture cannot hold a library convolution (MIOpen searches for a solution at a shape's first call)")
    _, _, p16 = segnet_models
    pb, mb = stage_b
    init, end = mb.segpoint[:, 0].contiguous(), mb.segpoint[:, 10].contiguous()
    obs, n_obs = mb.obstacles, mb.n_obstacles[:, 0].contiguous()
    g4 = mb.grid[:4].clone()
    cp4 = p16.capture(g4, init[:4].contiguous(), end[:4].contiguous(), obs[:4].contiguous(), n_obs[:4].contiguous())
    g6 = mb.grid[2:8].clone()
    cold = {}
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    torch.cuda.synchronize()
    graph6 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph6):                                             # no warm-up at this batch size
        cold["mask"] = p16.segment_u8(g6)
        cold["heat"] = p16.heatmap(cold["mask"])
    for rep in range(2):
        for grid4, grid6 in ((mb.grid[:4], mb.grid[2:8]), (mb.grid[4:8], mb.grid[:6])):
            cp4.grid.copy_(grid4)
            g6.copy_(grid6)
            cp4.replay()
            graph6.replay()
            torch.cuda.synchronize()
            m4, m6 = cp4.mask.clone(), cold["mask"].clone()
            h4, h6 = cp4.heat.clone(), cold["heat"].clone()
            want4 = p16.segment_u8(grid4.contiguous())                         # eager (from the second pass on: cached descriptors)
            want6 = p16.segment_u8(grid6.contiguous())
            assert torch.equal(m4, want4) and torch.equal(m6, want6)
            assert torch.equal(h4, p16.heatmap(want4)) and torch.equal(h6, p16.heatmap(want6))


@pytest.mark.gpu
def test_batches_on_two_streams_equal_one_stream(dev, stage_b, segnet_models):
    """bench.py alternates consecutive batches over two HIP streams.  Every buffer of the path belongs to its call (or is keyed
    by the stream), so two batches in flight must produce exactly what they produce one after the other."""
    _, _, p16 = segnet_models
    pb, mb = stage_b
    init, end = mb.segpoint[:, 0].contiguous(), mb.segpoint[:, 10].contiguous()
    obs, n_obs = mb.obstacles, mb.n_obstacles[:, 0].contiguous()
    grids = [mb.grid, mb.grid.flip(0).contiguous(), mb.grid.flip(1).contiguous(), mb.grid.flip(2).contiguous()]

    def one(g):
        mask = p16.segment_u8(g)
        heat = p16.heatmap(mask)
        return mask, heat, p16.plan_tail(heat, init, end, obs, n_obs)
    want = [one(g) for g in grids]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    got = []
    for rep in range(2):                                                       # second round: the streams' allocator pools are warm
        got = []
        for i, g in enumerate(grids):
            with torch.cuda.stream(streams[i % 2]):
                got.append(one(g))
        torch.cuda.synchronize()
        for (m0, h0, r0), (m1, h1, r1) in zip(want, got):
            assert torch.equal(m0, m1) and torch.equal(h0, h1)
            for k in ("ok", "counts", "collision", "waypoints"):
                assert torch.equal(r0[k], r1[k]), k
