"""Weight plumbing for `localizationNet`: reference `.npz` reader and a seeded synthetic
checkpoint in the reference's variable naming.

Reference: ckpt_manager.py:15-33,42 (tensorlayer `save_npz_dict` / `load_and_assign_npz_dict`:
an `.npz` whose keys are TF variable names, normally with a `:0` suffix), model.py:114-117,
networks.py:32-41 (scope prefix `stabNet/localizationNet/`), SURVEY.md section 8b (273
arrays: 53 conv `weights` HWIO + 53x4 BatchNorm arrays + 4 dense `W`[in,out], `b`).

No pretrained checkpoint ships with the reference (`pretrained/*` is git-ignored), so the
benchmarks and tests use `make_synthetic_weights`, which produces the same 273 arrays with
seeded values chosen so that activations stay O(1) and |F_t| stays ~0.05 (a realistic
control-point displacement).
"""
import collections
import os

import numpy as np

PREFIX = "stabNet/localizationNet/"
BLOCKS = (("block1", 64, 3, 2), ("block2", 128, 4, 2), ("block3", 256, 6, 2), ("block4", 512, 3, 1))
BN_KEYS = ("gamma", "beta", "moving_mean", "moving_variance")
DENSE_DIMS = ((2048, 2048), (2048, 1024), (1024, 512), (512, 50))


def conv_specs(c_in=21):
    """[(scope, k, cin, cout, stride, relu)] in execution order; shortcut first in a unit."""
    rn = PREFIX + "resnet_v1_50"
    specs = [(rn + "/conv1", 7, c_in, 64, 2, True)]
    depth_in = 64
    for bname, base, units, last_stride in BLOCKS:
        for u in range(1, units + 1):
            stride = last_stride if u == units else 1
            s = "%s/%s/unit_%d/bottleneck_v1" % (rn, bname, u)
            if depth_in != base * 4:
                specs.append((s + "/shortcut", 1, depth_in, base * 4, stride, False))
            specs.append((s + "/conv1", 1, depth_in, base, 1, True))
            specs.append((s + "/conv2", 3, base, base, stride, True))
            specs.append((s + "/conv3", 1, base, base * 4, 1, False))
            depth_in = base * 4
    return specs


def expected_names(c_in=21):
    names = []
    for scope, *_ in conv_specs(c_in):
        names.append(scope + "/weights")
        names += [scope + "/BatchNorm/" + k for k in BN_KEYS]
    for i in range(1, 5):
        names += [PREFIX + "df/dense%d/W" % i, PREFIX + "df/dense%d/b" % i]
    return names


def make_synthetic_weights(seed=0, c_in=21, f_scale=0.2):
    """Seeded stand-in checkpoint: 273 float32 arrays keyed like the reference's `.npz`
    (with the `:0` suffix tensorlayer writes)."""
    rng = np.random.default_rng(seed)
    w = collections.OrderedDict()

    def bn(scope, gamma_scale=1.0, var_scale=1.0, mean_std=0.1):
        c = w[scope + "/weights:0"].shape[3]
        w[scope + "/BatchNorm/gamma:0"] = (gamma_scale * rng.uniform(0.9, 1.1, c)).astype(np.float32)
        w[scope + "/BatchNorm/beta:0"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        w[scope + "/BatchNorm/moving_mean:0"] = (mean_std * rng.standard_normal(c)).astype(np.float32)
        w[scope + "/BatchNorm/moving_variance:0"] = (var_scale * rng.uniform(0.9, 1.1, c)).astype(np.float32)

    for scope, k, cin, cout, stride, relu in conv_specs(c_in):
        fan_in = k * k * cin
        w[scope + "/weights:0"] = (rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        if scope.endswith("resnet_v1_50/conv1"):
            bn(scope, var_scale=8000.0, mean_std=10.0)        # input is 255*x - mean: O(100)
        elif scope.endswith("/conv3"):
            bn(scope, gamma_scale=0.35)
        elif scope.endswith("/shortcut"):
            bn(scope, gamma_scale=0.7)
        else:
            bn(scope)
    for i, (din, dout) in enumerate(DENSE_DIMS, 1):
        std = 0.02 if i < 4 else 0.02 * f_scale
        w[PREFIX + "df/dense%d/W:0" % i] = (std * rng.standard_normal((din, dout))).astype(np.float32)
        w[PREFIX + "df/dense%d/b:0" % i] = (0.01 * rng.standard_normal(dout)).astype(np.float32)
    return w


def normalize_names(weights):
    """Accept keys with or without `:0`; returns {name_without_suffix: float32 ndarray}."""
    out = {}
    for k in weights:
        name = k[:-2] if k.endswith(":0") else k
        out[name] = np.ascontiguousarray(weights[k], dtype=np.float32)
    return out


def validate(weights, c_in=21):
    """Raise (never silently continue: the reference's `load_ckpt` swallows a missing
    checkpoint, ckpt_manager.py:17-22) unless every expected array is present with the
    expected shape.  Unknown keys are ignored."""
    w = normalize_names(weights)
    specs = {s: (k, cin, cout) for s, k, cin, cout, _, _ in conv_specs(c_in)}
    problems = []
    for name in expected_names(c_in):
        if name not in w:
            problems.append("missing " + name)
            continue
        shape = tuple(w[name].shape)
        scope, leaf = name.rsplit("/", 1)
        if leaf == "weights":
            k, cin, cout = specs[scope]
            want = (k, k, cin, cout)
        elif scope.endswith("/BatchNorm"):
            want = (specs[scope[:-len("/BatchNorm")]][2],)
        else:
            i = int(scope[-1]) - 1
            want = DENSE_DIMS[i] if leaf == "W" else (DENSE_DIMS[i][1],)
        if shape != want:
            problems.append("%s has shape %s, expected %s" % (name, shape, want))
    if problems:
        raise ValueError("checkpoint does not match localizationNet: " + "; ".join(problems[:8])
                         + (" ... (%d problems)" % len(problems) if len(problems) > 8 else ""))
    return w


def load_npz(path):
    """Read a reference checkpoint written by `tl.files.save_npz_dict` (ckpt_manager.py:42)."""
    if not os.path.isfile(path):
        raise FileNotFoundError("checkpoint not found: %s" % path)
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def load_ckpt_dir(ckpt_dir, by_score=True):
    """ckpt_manager.py:15-33: the text index `checkpoints` holds `"<file> <score>"` lines
    sorted by score with the most recent file on the last line; `by_score` picks line 0,
    otherwise the last line."""
    index = os.path.join(ckpt_dir, "checkpoints")
    if not os.path.isfile(index):
        raise FileNotFoundError("checkpoint index not found: %s" % index)
    with open(index) as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip()]
    if not lines:
        raise ValueError("checkpoint index is empty: %s" % index)
    name = (lines[0] if by_score else lines[-1]).split(" ")[0]
    return load_npz(os.path.join(ckpt_dir, name))
