import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import inputs
from coupe.dvsg_amd.networks import LocNet
from coupe.dvsg_amd.weights import make_synthetic_weights
from oracle.cnn_torch import TorchLocNet
w = make_synthetic_weights(0)
net = LocNet(w)
ref64 = TorchLocNet(w, dtype=torch.float64)
ref32 = TorchLocNet(w)
for seed, B, H, W in [(4242, 2, 288, 512), (4243, 2, 288, 512), (4244, 1, 720, 1280), (4245, 4, 64, 96)]:
    x = inputs.window_frames(seed, B, H, W)
    rF, rpool = ref64.forward(x), ref64.features(x).numpy()
    xt = torch.from_numpy(x).cuda()
    o = ref32.features(x).numpy()
    line = "seed %d %dx%d: oracle32 pool max %.3g rms %.3g |" % (seed, W, H, np.abs(o - rpool).max() / np.abs(rpool).max(), np.sqrt(((o - rpool) ** 2).mean()) / np.abs(rpool).max())
    for p in ("f32", "f32x3", "f32s"):
        F = net.forward(xt, precision=p).cpu().numpy()
        pool = net.tap(xt, 18, precision=p).cpu().numpy().reshape(rpool.shape)
        e = pool - rpool
        line += " %s: F %.3g pool max %.3g rms %.3g mean %.3g |" % (p, np.abs(F - rF).max(), np.abs(e).max() / np.abs(rpool).max(), np.sqrt((e ** 2).mean()) / np.abs(rpool).max(), e.mean() / np.abs(rpool).max())
    print(line, flush=True)
