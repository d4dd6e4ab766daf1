import os, sys, time, torch
ROOT='/root/repo'; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
from coupe.dvsg_amd.model import StabNet
from coupe.dvsg_amd.weights import make_synthetic_weights
H, W = int(sys.argv[1]), int(sys.argv[2])
net = StabNet(H, W).load_weights(make_synthetic_weights(0))
x = torch.rand((1, H, W, 21), device="cuda"); u = x[..., 18:].contiguous()
o = torch.empty((1, H, W, 3), device="cuda"); F = torch.empty((1, 25, 2), device="cuda")
for _ in range(3): net.locnet.stabilize(x, u, o, F)
torch.cuda.synchronize()
for _ in range(30): net.locnet.stabilize(x, u, o, F)
torch.cuda.synchronize()
