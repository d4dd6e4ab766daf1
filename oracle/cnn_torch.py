"""Oracle (test infrastructure): second, independently written CPU implementation of
networks.py `localizationNet` on torch-CPU (F.conv2d, NCHW, explicit F.pad for the TF
padding rules).  Cross-checks oracle/networks.py and is the multi-threaded CPU baseline
timed by bench.py (`cpu_baseline.kind = "port"`).  parity unpinned (see oracle/__init__.py).

Semantics restated from networks.py:6-16,30-46 and TF 1.11 tf.contrib.slim resnet_v1
(third-party; see oracle/networks.py header).
"""
import numpy as np
import torch
import torch.nn.functional as F

BLOCKS = [("block1", 64, 3, 2), ("block2", 128, 4, 2), ("block3", 256, 6, 2), ("block4", 512, 3, 1)]
PREFIX = "stabNet/localizationNet/"


def _w(weights, name, dtype=torch.float32):
    for k in (name, name + ":0"):
        if k in weights:
            return torch.from_numpy(np.ascontiguousarray(weights[k], dtype=np.float32)).to(dtype)
    raise KeyError(name)


class TorchLocNet:
    """Holds weights converted once (OIHW, BN as scale/shift applied AFTER the conv).  dtype=torch.float64: the same
    graph on the same float32 weights and frames with every operation in float64 -- an ARBITER between two float32
    evaluations (this one's and the GPU's), not the reference's arithmetic."""

    def __init__(self, weights, prefix=PREFIX, dtype=torch.float32):
        self.p = prefix
        self.dtype = dtype

        def _w(weights, name, _load=globals()["_w"]):   # the module-level loader, in this instance's dtype
            return _load(weights, name, dtype)
        self.convs = {}
        rn = prefix + "resnet_v1_50"
        scopes = [rn + "/conv1"]
        for bname, base, units, _ in BLOCKS:
            for u in range(1, units + 1):
                s = "%s/%s/unit_%d/bottleneck_v1" % (rn, bname, u)
                if u == 1:
                    scopes.append(s + "/shortcut")
                scopes += [s + "/conv1", s + "/conv2", s + "/conv3"]
        for s in scopes:
            w = _w(weights, s + "/weights").permute(3, 2, 0, 1).contiguous()
            g = _w(weights, s + "/BatchNorm/gamma")
            b = _w(weights, s + "/BatchNorm/beta")
            m = _w(weights, s + "/BatchNorm/moving_mean")
            v = _w(weights, s + "/BatchNorm/moving_variance")
            inv = g / torch.sqrt(v + 1e-5)
            self.convs[s] = (w, inv.view(1, -1, 1, 1), (b - m * inv).view(1, -1, 1, 1))
        self.dense = [(_w(weights, prefix + "df/dense%d/W" % i), _w(weights, prefix + "df/dense%d/b" % i))
                      for i in range(1, 5)]

    def _cb(self, x, scope, stride, relu, k):
        w, sc, sh = self.convs[scope]
        if k > 1:
            p = (k - 1) // 2
            x = F.pad(x, (p, k - 1 - p, p, k - 1 - p))
        elif stride > 1:
            x = x[:, :, ::stride, ::stride]
            stride = 1
        y = F.conv2d(x, w, stride=stride) * sc + sh
        return F.relu(y) if relu else y

    @torch.no_grad()
    def features(self, patches_nhwc):
        x = torch.as_tensor(patches_nhwc, dtype=torch.float32).to(self.dtype)
        # scale_RGB (networks.py:6-16): groups of 7 channels reversed, per-group mean
        x = x * 255.0
        g1, g2, g3 = torch.split(x, x.shape[3] // 3, dim=3)
        x = torch.cat([g3 - 103.939, g2 - 116.779, g1 - 123.68], dim=3)
        x = x.permute(0, 3, 1, 2).contiguous()
        rn = self.p + "resnet_v1_50"
        x = self._cb(x, rn + "/conv1", 2, True, 7)
        H, W = x.shape[2], x.shape[3]
        Ho, Wo = -(-H // 2), -(-W // 2)
        ph = max((Ho - 1) * 2 + 3 - H, 0)
        pw = max((Wo - 1) * 2 + 3 - W, 0)
        x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=float("-inf"))
        x = F.max_pool2d(x, 3, 2)
        for bname, base, units, last_stride in BLOCKS:
            for u in range(1, units + 1):
                stride = last_stride if u == units else 1
                s = "%s/%s/unit_%d/bottleneck_v1" % (rn, bname, u)
                if x.shape[1] == base * 4:
                    sc = x[:, :, ::stride, ::stride]
                else:
                    sc = self._cb(x, s + "/shortcut", stride, False, 1)
                r = self._cb(x, s + "/conv1", 1, True, 1)
                r = self._cb(r, s + "/conv2", stride, True, 3)
                r = self._cb(r, s + "/conv3", 1, False, 1)
                x = F.relu(sc + r)
        return x.mean(dim=(2, 3))

    @torch.no_grad()
    def forward(self, patches_nhwc, param_dim=25):
        h = self.features(patches_nhwc)
        for i, (W, b) in enumerate(self.dense):
            h = h @ W + b
            if i < 3:
                h = F.leaky_relu(h, 0.2)
        return h.reshape(-1, param_dim, 2).numpy()   # (float64 for the arbiter)
