"""The GenNet checkpoints this build trained itself (tools/train_gennet.py: the build's own training step — ppnet_amd.train.gennet_train_step,
GenNet/train.py:93-147's optimiser and loss — on pairs from the build's own generator; ppnet_amd/weights/gennet_r{256,512}.pth).

CPU: the files have the reference's checkpoint layout ({'model': state_dict}, what GenNet/predict.py:51-52 reads), load with
weights_only=True, and fit the reference architecture strictly (87 entries).  GPU: the harness's criterion on what the trained network
PREDICTS — label mask_space -> AE-ViT (prepared bf16 inference form) -> 8-bit heat map -> extract_path + collision check
(EDaGe-PP/process_map.py:452-506) -> success rate and plan length / target length (updated_geometric_planner.py:260-277)."""
import os

import pytest

torch = pytest.importorskip("torch")


@pytest.mark.parametrize("R", [256, 512])
def test_checkpoint_layout_is_the_references(R):
    from ppnet_amd.gennet import AEViT, trained_checkpoint, load_trained
    path = trained_checkpoint(R)
    assert path is not None and os.path.getsize(path) < 400_000
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert list(ck.keys()) == ["model"]
    sd = ck["model"]
    m = AEViT(1, 1, R, 24)
    assert list(sd.keys()) == list(m.state_dict().keys())
    assert all(v.dtype == torch.float32 for k, v in sd.items() if not k.endswith("num_batches_tracked"))
    assert all(torch.isfinite(v.float()).all() for v in sd.values())
    assert load_trained(m, R)
    assert sum(p.numel() for p in m.parameters()) == (53713 if R == 256 else 64225)       # 53 713 at down_time 3


@pytest.mark.gpu
@pytest.mark.parametrize("R,floor", [(256, 0.85), (512, 0.8)])
def test_trained_gennet_plans_on_its_own_predictions(R, floor):
    from ppnet_amd import _lib as L, edage, evaluate, fused, plan, train
    from ppnet_amd.gennet import AEViT, load_trained
    dev = torch.device("cuda:0")
    paths_n, placements = 16, 8
    pb = edage.generate_paths(paths_n, R, 50.0, 3.0, seed=4242, device=dev)
    mb = edage.generate_maps(pb, placements, 5.0, 20, seed=4242)
    _, mask_space, mask_path = train.generator_pairs(pb, mb, placements)
    gen = AEViT(1, 1, R, 24).eval()
    assert load_trained(gen, R)
    gen.prepare_inference()
    gen.to(dev).to(torch.bfloat16)
    with torch.no_grad():
        heat = fused.heatmap_u8(gen(mask_space.to(torch.bfloat16).unsqueeze(1)))
        init, end = mb.segpoint[:, 0].contiguous(), mb.segpoint[:, 10].contiguous()
        ok, wp, cnt = plan.extract_paths(heat, init, end, 2, L.MAX_WAYPOINTS)
        coll = plan.plan_collision(wp, cnt, mb.obstacles, mb.n_obstacles[:, 0].contiguous(), 1 / 50 * R, bound=R)
        ev = evaluate.evaluate_plans(dict(ok=ok, waypoints=wp, counts=cnt, collision=coll, success=ok & ~coll),
                                     pb.length.repeat_interleave(placements) * (R / 50.0))
    print(ev)
    assert ev["success"] >= floor and 1.0 <= ev["length_ratio"] < 1.2
