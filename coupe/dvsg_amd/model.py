"""Drop-in for the evaluation path of the reference's model.py on MI355X.

Reference surface kept (model.py:14-25, 98-123, eval.py:46-56,106-110):

    net = StabNet(h, w)
    inputs, outputs = net.get_evaluation_model(sample_num)
    sess = Session()                                   # stands in for tf.Session
    net.load_ckpt(ckpt_dir) | net.load_weights(dict)   # stands in for CKPT_Manager.load_ckpt
    s_t_pred = sess.run(outputs['s_t_pred'],
                        {inputs['patches_t']: frames21, inputs['u_t']: frames21[..., 18:]})

`inputs` / `outputs` are OrderedDicts with the reference's keys; their values are symbolic
handles (Placeholder / Fetch) because nothing is computed until `run`.  As in TF, only what
the requested fetches need is executed: fetching `s_t_pred` alone never touches the
`s_t_pred_mask` branch (model.py:121).  When `s_t_pred` (and optionally `F_t`,
`x_offset_t`, `y_offset_t`) is requested the whole graph runs as ONE C-ABI call,
`dvsg_stabilize_f32`.  Training graphs (`get_train_model`, `init_vars`, `random_mask`'s
random H) are out of scope.
"""
import collections

import numpy as np
import torch

from . import _lib, weights as _weights
from .ThinPlateSpline import ThinPlateSpline as stn
from .networks import LocNet
from ._tensor import as_dev, empty, is_host, ptr, stream


class Placeholder(object):
    def __init__(self, name, channels):
        self.name = name
        self.channels = channels

    def __repr__(self):
        return "<Placeholder %s [None,None,None,%d]>" % (self.name, self.channels)


class Fetch(object):
    def __init__(self, model, key):
        self.model = model
        self.key = key

    def __repr__(self):
        return "<Fetch %s>" % self.key


V_SRC = np.array([  # source position (model.py:105-110)
    [-1, -1], [-0.5, -1], [0, -1], [0.5, -1], [1, -1],
    [-1, -0.5], [-0.5, -0.5], [0, -0.5], [0.5, -0.5], [1, -0.5],
    [-1, 0], [-0.5, 0], [0, 0], [0.5, 0], [1, 0],
    [-1, 0.5], [-0.5, 0.5], [0, 0.5], [0.5, 0.5], [1, 0.5],
    [-1, 1], [-0.5, 1], [0, 1], [0.5, 1], [1, 1]], dtype=np.float32)


class StabNet:
    def __init__(self, h, w):
        self.h = h
        self.w = w
        self.c = 3
        self.num_control_points = 5
        self.param_dim = self.num_control_points ** 2
        self.stabNet_model = 'resnet_v1_50'
        self.n_streams = 1   # 2 = batch halves on two HIP streams (LocNet.stabilize): <1 % at B=16 720p
        # "f32": exact float32 matrix cores (the reference's arithmetic, the path of record); "f32s": float32 width and
        # accumulation, products from two float16 pieces per operand (2x faster; float32-GEMM-level differences while operand
        # magnitudes stay above ~2^-3 .. 0.03, fewer bits below: include/dvsg_amd.h);
        # "f16": float16 activations, hi / lo float16 conv weights
        self.precision = "f32"
        self.locnet = None
        self.inputs = None
        self.outputs = None

    # -- weights (eval.py:56 / ckpt_manager.py:15-33) ---------------------------------------
    def load_weights(self, weights):
        self.locnet = LocNet(weights)
        return self

    def load_ckpt(self, ckpt_dir, by_score=True):
        return self.load_weights(_weights.load_ckpt_dir(ckpt_dir, by_score))

    def init_vars(self, weights, ckpt_path='./pretrained/resnet_v1_50.ckpt', sess=None):
        """model.py:125-154: the trainer's ImageNet initialisation, for evaluation-only users.  `weights` (a dict in the
        reference's variable naming, e.g. a `.npz` of ckpt_manager.py or `make_synthetic_weights`) supplies what the
        slim checkpoint does not hold or must not overwrite -- the 21-channel root `conv1` (excluded, :126) and the
        tensorlayer dense head -- and every other `resnet_v1_50/...` array is read from the TensorFlow checkpoint at
        `ckpt_path` (V1 single file or V2 bundle prefix; read without TensorFlow by `tf_checkpoint`, parity unpinned).
        `sess` is accepted for call-site symmetry and not used."""
        from .tf_checkpoint import init_from_slim_checkpoint
        return self.load_weights(init_from_slim_checkpoint(weights, ckpt_path, model=self.stabNet_model))

    # -- graph (model.py:98-123) ------------------------------------------------------------
    def get_evaluation_model(self, sample_num):
        self.sample_num = sample_num
        inputs = collections.OrderedDict()
        inputs['patches_t'] = Placeholder('input_frames_t', 3 * sample_num)
        inputs['u_t'] = Placeholder('unstable_frame_t', 3)
        outputs = collections.OrderedDict()
        for key in ('V_src', 'num_control_points', 'F_t', 's_t_pred', 'x_offset_t', 'y_offset_t',
                    's_t_pred_mask'):
            outputs[key] = Fetch(self, key)
        outputs['num_control_points'] = self.num_control_points
        self.inputs, self.outputs = inputs, outputs
        return inputs, outputs

    # -- eval_train.py's occlusion mask (model.py:156-167) ----------------------------------
    def random_mask(self, patches, out_size, sample_num, H=None, generator=None):
        """model.py:156-167: a ones mask over the history frames (first 3*(sample_num-1) channels)
        is warped by a random near-identity homography (zeros enter from outside the frame) and
        multiplies the window; the current frame's 3 channels are never masked.  The reference
        draws H with tf.random_uniform inside the graph; pass `H` [B,8] to make it reproducible
        (or a torch `generator`).  Returns (patches * mask, mask)."""
        from .spatial_transformer import ProjectiveTransformer
        p = as_dev(patches)
        B = p.shape[0]
        c_hist = 3 * (sample_num - 1)
        if H is None:
            gdev = p.device if generator is None else generator.device
            u = torch.rand((B, 8), generator=generator, device=gdev).to(p.device) * 2.0 - 1.0       # :161
            Ht = u * torch.tensor([0.1, 0.1, 0.5, 0.1, 0.1, 0.5, 0.1, 0.1], device=p.device)       # :162
            Ht = Ht + torch.tensor([1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], device=p.device)      # :163
        else:
            Ht = as_dev(H).reshape(B, 8)
        mask = ProjectiveTransformer(out_size).transform(torch.ones_like(p[..., :c_hist]).contiguous(), Ht)  # :164
        mask = torch.cat([mask, torch.ones_like(p[..., :3])], dim=3)                                # :165
        out = p * mask
        if is_host(patches):
            return out.cpu().numpy(), mask.cpu().numpy()
        return out, mask

    # -- execution --------------------------------------------------------------------------
    def _execute(self, keys, feed):
        if self.locnet is None:
            raise _lib.DvsgError("StabNet has no weights: call load_weights()/load_ckpt() first (the "
                                 "reference would silently evaluate random weights, ckpt_manager.py:21-22)")
        vals = {}
        need_cnn = any(k in keys for k in ('F_t', 's_t_pred', 'x_offset_t', 'y_offset_t', 's_t_pred_mask'))
        patches = u_t = None
        if need_cnn:
            if self.inputs['patches_t'] not in feed:
                raise KeyError("feed_dict lacks inputs['patches_t']")
            patches = as_dev(feed[self.inputs['patches_t']])
            if patches.dim() != 4 or patches.shape[3] != 3 * self.sample_num:
                raise ValueError("patches_t must be [B,H,W,%d]" % (3 * self.sample_num))
            if 3 * self.sample_num != self.locnet.in_channels:
                raise ValueError("get_evaluation_model(%d) feeds %d channels but conv1 of the loaded checkpoint "
                                 "has %d" % (self.sample_num, 3 * self.sample_num, self.locnet.in_channels))
        need_warp = any(k in keys for k in ('s_t_pred', 'x_offset_t', 'y_offset_t', 's_t_pred_mask'))
        if need_warp or 'V_src' in keys:
            if self.inputs['u_t'] not in feed:
                raise KeyError("feed_dict lacks inputs['u_t']")
            u_t = as_dev(feed[self.inputs['u_t']])
            if u_t.dim() != 4 or u_t.shape[3] != 3:
                raise ValueError("u_t must be [B,H,W,3]")
        if 'V_src' in keys:  # tiled over the batch of u_t (model.py:111)
            vals['V_src'] = torch.from_numpy(V_SRC).to(u_t.device).unsqueeze(0).repeat(u_t.shape[0], 1, 1)
        if need_warp:
            B, H, W, _ = u_t.shape
            if (H, W) != (self.h, self.w) or tuple(patches.shape[:3]) != (B, H, W):
                raise ValueError("fed frames must be [B,%d,%d,*] (StabNet(h, w) fixes the STN out_size)"
                                 % (self.h, self.w))
            out = empty((B, H, W, 3), u_t)
            F = empty((B, self.param_dim, 2), u_t)
            want_xy = 'x_offset_t' in keys or 'y_offset_t' in keys
            xs = empty((B * H * W,), u_t) if want_xy else None
            ys = empty((B * H * W,), u_t) if want_xy else None
            self.locnet.stabilize(patches, u_t, out, F, xs, ys, n_streams=self.n_streams, precision=self.precision)
            vals.update(F_t=F, s_t_pred=out, x_offset_t=xs, y_offset_t=ys)
            if 's_t_pred_mask' in keys:  # model.py:121
                V = torch.from_numpy(V_SRC).to(u_t.device).unsqueeze(0).repeat(B, 1, 1)
                vals['s_t_pred_mask'], _, _ = stn(torch.ones_like(u_t), V, F, [self.h, self.w])
        elif 'F_t' in keys:
            vals['F_t'] = self.locnet.forward(patches, self.param_dim, precision=self.precision)
        return vals


class Session(object):
    """Minimal stand-in for the `tf.Session` of eval.py:46: `run(fetches, feed_dict)`.
    NumPy feeds give NumPy results (as sess.run does); torch feeds stay on the device."""

    def run(self, fetches, feed_dict=None):
        feed = feed_dict or {}
        single = not isinstance(fetches, (list, tuple))
        flist = [fetches] if single else list(fetches)
        models = {f.model for f in flist if isinstance(f, Fetch)}
        if len(models) > 1:
            raise ValueError("fetches belong to different StabNet instances")
        vals = {}
        if models:
            model = models.pop()
            vals = model._execute({f.key for f in flist if isinstance(f, Fetch)}, feed)
        host = any(is_host(v) for v in feed.values())
        res = []
        for f in flist:
            if isinstance(f, Fetch):
                v = vals[f.key]
                res.append(v.cpu().numpy() if host and isinstance(v, torch.Tensor) else v)
            else:
                res.append(f)  # plain Python value, e.g. outputs['num_control_points']
        return res[0] if single else res
