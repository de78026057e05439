import sys; sys.path.insert(0, '/root/repo')
import torch
from ppnet_amd import edage
pb = edage.generate_paths(1000, 256, 50, 3, seed=0, device=torch.device('cuda:0'))
hn = pb.hull_n.float()
print('hull_n mean', hn.mean().item(), 'max', hn.max().item(), 'min', hn.min().item(), 'straight frac', pb.straight.float().mean().item())
