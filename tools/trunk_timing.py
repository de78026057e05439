"""Times the fused GenNet trunk kernel alone (batch 256, 32 x 32 tokens, bf16).  PPNET_TRUNK_VALU=1 selects the v_dot2 kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppnet_amd import fused
from ppnet_amd.gennet import _Block, pack_trunk_params
torch.manual_seed(0)
blocks = torch.nn.Sequential(*[_Block(24, 3, 4) for _ in range(3)]).cuda().eval()
params = pack_trunk_params(blocks).cuda()
x = torch.randn(256, 24, 32, 32, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
for nb in (3, 1):
    for _ in range(3):
        fused.gennet_trunk(x, params[:nb * 7224].contiguous(), nb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fused.gennet_trunk(x, params[:nb * 7224].contiguous(), nb)
    e1.record(); torch.cuda.synchronize()
    print(f"{'valu' if os.environ.get('PPNET_TRUNK_VALU') else 'mfma'} trunk, {nb} block(s): {e0.elapsed_time(e1) / 10:.3f} ms per batch of 256")
