"""Oracle (test infrastructure): networks.py `scale_RGB` + `localizationNet` restated in
NumPy float32.  parity unpinned (see oracle/__init__.py).

Reference call sites: networks.py:6-16 (scale_RGB), networks.py:30-46 (localizationNet).

THIRD-PARTY ARITHMETIC.  `resnet_v1_50`, `resnet_arg_scope`, `bottleneck`, `conv2d_same`,
`subsample` are tensorflow.contrib.slim (TF 1.11; pinned only by the reference README.md:5;
source not in the reference tree nor in this container).  They are restated here from the
published TF r1.11 definition (contrib/slim/python/slim/nets/resnet_v1.py, resnet_utils.py):

* resnet_arg_scope: conv2d has no bias, is followed by batch_norm(decay .997, epsilon 1e-5,
  scale=True) and ReLU; max_pool2d uses padding='SAME'.
* root: conv2d_same(64, 7, stride 2) = explicit zero pad 3/3 then VALID conv; then
  max_pool2d(3x3, stride 2, SAME).
* blocks (base depth, units, stride of the LAST unit): (64,3,2) (128,4,2) (256,6,2) (512,3,1).
* bottleneck(depth=4*base, depth_bottleneck=base, stride): shortcut = subsample(x, stride)
  (1x1 max-pool, i.e. x[:, ::s, ::s]) when depth_in == depth else conv 1x1 stride + BN (no
  ReLU); residual = conv1 1x1 (BN, ReLU) -> conv2 3x3 conv2d_same(stride) (BN, ReLU) ->
  conv3 1x1 (BN, no ReLU); out = relu(shortcut + residual).
* global_pool: reduce_mean over H, W.
`DenseLayer` is tensorlayer 1.x (version unpinned): y = act(x @ W + b), W [in, out].
"""
import numpy as np

F32 = np.float32
BN_EPS = 1e-5
RESNET_MEAN = [103.939, 116.779, 123.68]            # networks.py:7
BLOCKS = [("block1", 64, 3, 2), ("block2", 128, 4, 2), ("block3", 256, 6, 2), ("block4", 512, 3, 1)]
PREFIX = "stabNet/localizationNet/"                 # model.py:114-117


def scale_RGB(rgb):
    """networks.py:6-16.  Written for 3 channels; on the 21-channel window tf.split cuts
    the channel axis into 3 groups of 7 and the concat reverses the GROUPS."""
    rgb = np.asarray(rgb, dtype=F32)
    rgb_scaled = (rgb * F32(255.0)).astype(F32)                       # :8
    red, green, blue = np.split(rgb_scaled, 3, axis=3)                # :9
    return np.concatenate([                                           # :10-14
        (blue - F32(RESNET_MEAN[0])).astype(F32),
        (green - F32(RESNET_MEAN[1])).astype(F32),
        (red - F32(RESNET_MEAN[2])).astype(F32)], axis=3)


def _get(weights, name):
    for k in (name, name + ":0"):
        if k in weights:
            return np.asarray(weights[k], dtype=F32)
    raise KeyError(name)


def conv2d_valid(x, w, stride):
    """VALID NHWC conv, HWIO weights, as an im2col GEMM in float32."""
    B, H, W, C = x.shape
    kh, kw, ci, co = w.shape
    assert ci == C
    Ho = (H - kh) // stride + 1
    Wo = (W - kw) // stride + 1
    s0, s1, s2, s3 = x.strides
    cols = np.lib.stride_tricks.as_strided(
        x, shape=(B, Ho, Wo, kh, kw, C),
        strides=(s0, s1 * stride, s2 * stride, s1, s2, s3), writeable=False)
    a = np.ascontiguousarray(cols).reshape(B * Ho * Wo, kh * kw * C)
    out = a @ w.reshape(kh * kw * C, co)
    return out.reshape(B, Ho, Wo, co).astype(F32)


def conv2d_same_slim(x, w, stride):
    """slim resnet_utils.conv2d_same: stride 1 -> SAME; else pad (k-1)//2 / rest, VALID."""
    k = w.shape[0]
    pad_total = k - 1
    pb = pad_total // 2
    pe = pad_total - pb
    xp = np.pad(x, [[0, 0], [pb, pe], [pb, pe], [0, 0]])
    return conv2d_valid(xp, w, stride)


def batch_norm_inference(x, weights, scope):
    gamma = _get(weights, scope + "/BatchNorm/gamma")
    beta = _get(weights, scope + "/BatchNorm/beta")
    mean = _get(weights, scope + "/BatchNorm/moving_mean")
    var = _get(weights, scope + "/BatchNorm/moving_variance")
    inv = (gamma / np.sqrt(var + F32(BN_EPS))).astype(F32)
    return (x * inv + (beta - mean * inv).astype(F32)).astype(F32)


def max_pool_3x3_s2_same(x):
    """TF SAME: out = ceil(n/2), pad_total = max((out-1)*2+3-n, 0), pad_before = total//2."""
    B, H, W, C = x.shape
    Ho, Wo = -(-H // 2), -(-W // 2)
    pth = max((Ho - 1) * 2 + 3 - H, 0)
    ptw = max((Wo - 1) * 2 + 3 - W, 0)
    xp = np.pad(x, [[0, 0], [pth // 2, pth - pth // 2], [ptw // 2, ptw - ptw // 2], [0, 0]],
                constant_values=-np.inf)
    out = np.full((B, Ho, Wo, C), -np.inf, dtype=F32)
    for i in range(3):
        for j in range(3):
            out = np.maximum(out, xp[:, i:i + 2 * Ho - 1:2, j:j + 2 * Wo - 1:2, :])
    return out


def conv_bn(x, weights, scope, stride, relu, same):
    w = _get(weights, scope + "/weights")
    y = conv2d_same_slim(x, w, stride) if same else conv2d_valid(x[:, ::stride, ::stride, :], w, 1)
    y = batch_norm_inference(y, weights, scope)
    return np.maximum(y, F32(0)) if relu else y


def bottleneck(x, weights, scope, depth, depth_bottleneck, stride):
    if x.shape[3] == depth:
        shortcut = x[:, ::stride, ::stride, :]                        # subsample
    else:
        shortcut = conv_bn(x, weights, scope + "/shortcut", stride, relu=False, same=False)
    r = conv_bn(x, weights, scope + "/conv1", 1, relu=True, same=False)
    r = conv_bn(r, weights, scope + "/conv2", stride, relu=True, same=True)
    r = conv_bn(r, weights, scope + "/conv3", 1, relu=False, same=False)
    return np.maximum((shortcut + r).astype(F32), F32(0))


def resnet_v1_50_features(x, weights, prefix=PREFIX, taps=None):
    """[B,H,W,21] scaled input -> [B,2048] (global_pool=True, num_classes=None)."""
    rn = prefix + "resnet_v1_50"
    net = conv_bn(x, weights, rn + "/conv1", 2, relu=True, same=True)
    if taps is not None:
        taps["conv1"] = net
    net = max_pool_3x3_s2_same(net)
    if taps is not None:
        taps["pool1"] = net
    for bname, base, units, last_stride in BLOCKS:
        for u in range(1, units + 1):
            stride = last_stride if u == units else 1
            scope = "%s/%s/unit_%d/bottleneck_v1" % (rn, bname, u)
            net = bottleneck(net, weights, scope, base * 4, base, stride)
            if taps is not None:
                taps["%s/unit_%d" % (bname, u)] = net
    return np.mean(net, axis=(1, 2), dtype=F32).astype(F32)


def dense(x, weights, scope, act):
    W = _get(weights, scope + "/W")
    b = _get(weights, scope + "/b")
    y = (x @ W + b).astype(F32)
    return act(y)


def lrelu02(x):
    return np.where(x >= 0, x, (F32(0.2) * x).astype(F32)).astype(F32)   # networks.py:31


def localizationNet(inp, param_dim, weights, prefix=PREFIX, taps=None):
    """networks.py:30-46.  inp [B,H,W,21] in [0,1] -> [B,param_dim,2]."""
    feat = resnet_v1_50_features(scale_RGB(inp), weights, prefix, taps)           # :34
    if taps is not None:
        taps["pool5"] = feat
    net = dense(feat, weights, prefix + "df/dense1", lrelu02)                     # :38
    net = dense(net, weights, prefix + "df/dense2", lrelu02)                      # :39
    net = dense(net, weights, prefix + "df/dense3", lrelu02)                      # :40
    net = dense(net, weights, prefix + "df/dense4", lambda v: v)                  # :41
    return net.reshape(-1, param_dim, 2)                                          # :44
