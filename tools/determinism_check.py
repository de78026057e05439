"""Is PPNet.segment_u8 bit-reproducible run to run (one stream), and across two streams?  Prints mismatching label / logit counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppnet_amd import edage, fused
from ppnet_amd.gennet import AEViT
from ppnet_amd.ppnet import PPNet
from ppnet_amd.segnet import IMG_MEAN, IMG_STD, SegNet, balance_classifier_bias, normalize_images, randomize_neutral_parameters

dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(0)
seg = randomize_neutral_parameters(SegNet().eval(), seed=1).to(dev)
pb = edage.generate_paths(2, R, 50, 3, seed=5, device=dev)
mb = edage.generate_maps(pb, 4, 5, 20, seed=5)
balance_classifier_bias(seg, normalize_images(edage.grid_to_rgb(mb.grid[:2]) * 255.0))
model = PPNet(R, segnet=seg, gennet=AEViT(1, 1, R, 24).eval()).to(dev).eval()
g = mb.grid
x = fused.grid_to_image(g, IMG_MEAN, IMG_STD, torch.bfloat16)
with torch.no_grad():
    l = [model.segnet.encode_decode(x).float() for _ in range(4)]
    m = [model.segment_u8(g) for _ in range(4)]
torch.cuda.synchronize()
for i in range(1, 4):
    print(f"run {i} vs 0: logits differ at {(l[i] != l[0]).sum().item()} of {l[0].numel()} (max {float((l[i] - l[0]).abs().max()):.3g}), labels at {(m[i] != m[0]).sum().item()}")
st = [torch.cuda.Stream(dev) for _ in range(2)]
outs = []
for i in range(4):
    with torch.cuda.stream(st[i % 2]):
        outs.append(model.segment_u8(g))
torch.cuda.synchronize()
for i in range(4):
    print(f"two-stream run {i} vs sequential: labels differ at {(outs[i] != m[0]).sum().item()}")
# backbone features only
with torch.no_grad():
    f = [model.segnet.backbone(model.segnet.backbone.patch_embed.codes_or_image(g) if hasattr(model.segnet.backbone.patch_embed, 'codes_or_image') else x) for _ in range(2)]
try:
    for a, b in zip(f[0], f[1]):
        if a is not None:
            print("backbone level equal:", torch.equal(a, b))
except Exception as e:
    print("backbone compare skipped", e)
