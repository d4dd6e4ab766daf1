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
`dvsg_stabilize_f32`.  Training graphs (`get_train_model`) are out of scope.  eval_train.py builds
its OWN evaluation graph, whose CNN input is masked (eval_train.py:25-51): that one is
`coupe.dvsg_amd.eval_train.get_evaluation_model`, a `StabNet` with `masked = True`.
"""
import collections

import numpy as np
import torch

from . import _lib, weights as _weights
from .ThinPlateSpline import ThinPlateSpline as stn
from .networks import LocNet, random_mask_plane
from ._tensor import as_dev, empty, is_host, ptr, stream

RANDOM_MASK_SCALE = (0.1, 0.1, 0.5, 0.1, 0.1, 0.5, 0.1, 0.1)    # model.py:162
RANDOM_MASK_OFFSET = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0)   # model.py:163


def draw_random_H(B, device, generator=None):
    """model.py:161-163 / eval_train.py:55-57: H = uniform[-1,1) * scale + identity, [B,8] float32 on `device`.
    torch.rand stands in for tf.random_uniform (another generator: the VALUES differ from TensorFlow's, the
    distribution does not)."""
    gdev = device if generator is None else generator.device
    u = torch.rand((B, 8), generator=generator, device=gdev).to(device) * 2.0 - 1.0           # :161
    return u * torch.tensor(RANDOM_MASK_SCALE, device=device) + torch.tensor(RANDOM_MASK_OFFSET, device=device)


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
        # "f32x3": float32 tensors and accumulation, conv products from three bfloat16 pieces per operand (all 24 bits of both
        # operands at any magnitude; as close to float64 as "f32" is, tests/test_gpu_f32x3.py); "f16": float16 activations,
        # hi / lo float16 conv weights
        self.precision = "f32"
        # eval_train.py's graph (eval_train.py:43-45): F_t = localizationNet(patches_t * random mask); False = model.py's
        self.masked = False
        self.mask_generator = None   # torch.Generator of the in-graph draw when inputs['random_H'] is not fed
        self.locnet = None
        self.inputs = None
        self.outputs = None

    # -- weights (eval.py:56 / ckpt_manager.py:15-33) ---------------------------------------
    def load_weights(self, weights):
        self.locnet = LocNet(weights)
        self._host_weights = weights   # what `init_vars(sess)` merges the slim checkpoint into
        return self

    def load_ckpt(self, ckpt_dir, by_score=True):
        return self.load_weights(_weights.load_ckpt_dir(ckpt_dir, by_score))

    def calibrate_f16(self, patches):
        """Optional, for precision = "f16": `LocNet.calibrate_f16` on a few windows [B,H,W,3S] of the clip (no counterpart
        in the reference, which has one precision)."""
        if self.locnet is None:
            raise _lib.DvsgError("StabNet has no weights: call load_weights()/load_ckpt() first")
        self.locnet.calibrate_f16(patches)
        return self

    def init_vars(self, sess=None, ckpt_path=None, weights=None):
        """model.py:125-154, callable as the reference calls it -- `net.init_vars(sess)` -- once the variables exist
        (`load_weights` / `load_ckpt` stand in for `tf.global_variables_initializer()`): every `resnet_v1_50/...`
        variable of the regressor EXCEPT the 21-channel root `conv1` (excluded, :126) is overwritten from the slim
        checkpoint at `ckpt_path` (default './pretrained/resnet_v1_50.ckpt', :149; V1 single file or V2 bundle prefix,
        read without TensorFlow by `tf_checkpoint`, parity unpinned), `ignore_missing_vars=False`; `conv1` and the
        tensorlayer dense head keep the values they have.  `weights` (a dict in the reference's variable naming) may
        supply those instead of a previous `load_weights`.  `sess` is accepted for call-site symmetry and not used
        (a dict passed in its place is taken as `weights`: the round-3 signature)."""
        from .tf_checkpoint import init_from_slim_checkpoint
        if isinstance(sess, dict) and weights is None:
            sess, weights = None, sess
        if weights is None:
            weights = getattr(self, "_host_weights", None)
        if weights is None:
            raise _lib.DvsgError("init_vars: no variables to initialise -- call load_weights()/load_ckpt() first or pass "
                                 "weights= (conv1 and the dense head are not in the slim checkpoint, model.py:126)")
        if ckpt_path is None:
            ckpt_path = './pretrained/{}.ckpt'.format(self.stabNet_model)
        return self.load_weights(init_from_slim_checkpoint(weights, ckpt_path, model=self.stabNet_model))

    # -- graph (model.py:98-123) ------------------------------------------------------------
    def get_evaluation_model(self, sample_num):
        self.sample_num = sample_num
        inputs = collections.OrderedDict()
        inputs['patches_t'] = Placeholder('input_frames_t', 3 * sample_num)
        inputs['u_t'] = Placeholder('unstable_frame_t', 3)
        outputs = collections.OrderedDict()
        if self.masked:   # eval_train.py:25-51 (no 'num_control_points' entry there)
            # the graph's tf.random_uniform draw, exposed as an OPTIONAL feed: H [B,8] after the scale / offset of :56-57
            inputs['random_H'] = Placeholder('random_H', 8)
            keys = ('V_src', 'patches_masked_t', 'random_masks_t', 'F_t', 's_t_pred', 'x_offset_t', 'y_offset_t',
                    's_t_pred_mask')
        else:
            keys = ('V_src', 'num_control_points', 'F_t', 's_t_pred', 'x_offset_t', 'y_offset_t', 's_t_pred_mask')
        for key in keys:
            outputs[key] = Fetch(self, key)
        if not self.masked:
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
        p = as_dev(patches)
        B = p.shape[0]
        c_hist = 3 * (sample_num - 1)
        Ht = draw_random_H(B, p.device, generator) if H is None else as_dev(H).reshape(B, 8)        # :161-163
        # :164 -- the warp of an all-ones image is the same in every channel: one plane (dvsg_random_mask_plane_f32),
        # broadcast over the history channels only because this operator returns the reference's [B,H,W,3S] tensors;
        # the evaluation graph (masked = True) hands the plane itself to conv1's load stage instead
        plane = random_mask_plane(Ht, int(out_size[0]), int(out_size[1]))
        if tuple(plane.shape[1:]) != tuple(p.shape[1:3]):
            raise ValueError("out_size %s must equal the patch size %s (patches * mask)" % (tuple(out_size), tuple(p.shape[1:3])))
        mask = torch.cat([plane.unsqueeze(3).expand(-1, -1, -1, c_hist), torch.ones_like(p[..., :3])], dim=3)   # :165
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
        need_cnn = any(k in keys for k in ('F_t', 's_t_pred', 'x_offset_t', 'y_offset_t', 's_t_pred_mask',
                                           'patches_masked_t', 'random_masks_t'))
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
        plane = None
        if need_cnn and self.masked:   # eval_train.py:43: the graph's random mask, as one [B,H,W] plane
            Bp = int(patches.shape[0])
            if tuple(patches.shape[1:3]) != (self.h, self.w):
                raise ValueError("fed frames must be [B,%d,%d,*] (random_mask's out_size)" % (self.h, self.w))
            fed = feed.get(self.inputs['random_H'])
            Ht = draw_random_H(Bp, patches.device, self.mask_generator) if fed is None else as_dev(fed).reshape(Bp, 8)
            plane = random_mask_plane(Ht, self.h, self.w)
            if 'random_masks_t' in keys or 'patches_masked_t' in keys:   # debugging fetches: materialised only when asked for
                m21 = torch.cat([plane.unsqueeze(3).expand(-1, -1, -1, 3 * (self.sample_num - 1)),
                                 torch.ones_like(patches[..., :3])], dim=3)
                vals['random_masks_t'] = m21
                vals['patches_masked_t'] = patches * m21
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
            self.locnet.stabilize(patches, u_t, out, F, xs, ys, n_streams=self.n_streams, precision=self.precision,
                                  mask=plane)
            vals.update(F_t=F, s_t_pred=out, x_offset_t=xs, y_offset_t=ys)
            if 's_t_pred_mask' in keys:  # model.py:121
                V = torch.from_numpy(V_SRC).to(u_t.device).unsqueeze(0).repeat(B, 1, 1)
                vals['s_t_pred_mask'], _, _ = stn(torch.ones_like(u_t), V, F, [self.h, self.w])
        elif 'F_t' in keys:
            if plane is not None:
                vals['F_t'] = self.locnet.forward_masked(patches, plane, precision=self.precision)
            else:
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
