"""Oracle (test infrastructure): model.py `StabNet.get_evaluation_model` and the eval.py
frame loop restated in NumPy float32.  parity unpinned (see oracle/__init__.py).

Reference: model.py:14-25 (ctor), model.py:98-123 (evaluation graph), eval.py:93-124
(autoregressive clip loop).
"""
import collections

import numpy as np

from .networks import localizationNet
from .thin_plate_spline import ThinPlateSpline as stn

F32 = np.float32


def v_src(num_control_points=5):
    """model.py:105-110: 5x5 grid on [-1,1]^2, x fastest."""
    lin = np.linspace(-1.0, 1.0, num_control_points)
    pts = [[x, y] for y in lin for x in lin]
    return np.array(pts, dtype=F32)


class StabNet:
    def __init__(self, h, w):
        self.h = h
        self.w = w
        self.c = 3
        self.num_control_points = 5
        self.param_dim = self.num_control_points ** 2

    def run(self, weights, patches_t, u_t, fetch=("s_t_pred",), taps=None):
        """Equivalent of ``sess.run([outputs[k] for k in fetch], {patches_t, u_t})`` on the
        graph of model.py:98-123."""
        patches_t = np.asarray(patches_t, dtype=F32)
        u_t = np.asarray(u_t, dtype=F32)
        B = u_t.shape[0]
        out = collections.OrderedDict()
        out["V_src"] = np.tile(v_src(self.num_control_points)[None], (B, 1, 1))     # :111
        out["num_control_points"] = self.num_control_points
        out["F_t"] = localizationNet(patches_t, self.param_dim, weights, taps=taps)  # :117
        if any(k in fetch for k in ("s_t_pred", "x_offset_t", "y_offset_t")):
            out["s_t_pred"], out["x_offset_t"], out["y_offset_t"] = stn(
                u_t, out["V_src"], out["F_t"], [self.h, self.w])                    # :120
        if "s_t_pred_mask" in fetch:
            out["s_t_pred_mask"], _, _ = stn(
                np.ones_like(u_t), out["V_src"], out["F_t"], [self.h, self.w])      # :121
        return [out[k] for k in fetch]


def random_mask(patches, out_size, sample_num, H):
    """model.py:156-167 with the homography parameters H [B,8] supplied by the caller (the reference
    draws them with tf.random_uniform and applies the scale / identity offset of lines 162-163)."""
    from .spatial_transformer import ProjectiveTransformer
    patches = np.asarray(patches, dtype=F32)
    c_hist = 3 * (sample_num - 1)
    mask = np.ones_like(patches[..., :c_hist])                                   # :160
    mask = ProjectiveTransformer(out_size).transform(mask, np.asarray(H, F32))    # :164
    mask = np.concatenate([mask, np.ones_like(patches[..., :3])], axis=3)        # :165
    return (patches * mask).astype(F32), mask.astype(F32)                        # :167


def eval_clip(weights, frames, h, w, skip_length=(0, 16, 24, 28, 30, 31, 32), grids=None):
    """eval.py:93-124 on an in-memory clip ``frames`` [N,h,w,3] float (already RGB, /255,
    resized: eval.py:76-81 is cv2 I/O and out of scope).

    Returns (stabilised [N,h,w,3] float32, side_by_side uint8 [N,h,2w,3]).  A list passed as `grids`
    receives each step's source grid (x_offset_t, y_offset_t) -- the tests mask sampler A's border jumps with it.
    """
    skip_length = np.array(skip_length)
    span = int(skip_length[-1] - skip_length[0])
    total = [np.asarray(f, dtype=np.float64) for f in frames]        # frame/255. is float64 (eval.py:80)
    for _ in range(span):
        total.insert(0, total[0])                                    # :93-94
    total = np.array(total)
    net = StabNet(h, w)
    outs, sbs = [], []
    sample_idx = skip_length.copy()
    for frame_idx in range(span, len(total)):                        # :101
        batch = total[sample_idx]                                    # :103
        batch = np.expand_dims(np.concatenate(batch, axis=2), 0)     # :104
        s_t_pred, xs, ys = net.run(weights, batch.astype(F32), batch[:, :, :, 18:].astype(F32),
                                   fetch=("s_t_pred", "x_offset_t", "y_offset_t"))                     # :106-110
        s_t_pred = np.squeeze(s_t_pred)
        if grids is not None:
            grids.append((xs, ys))
        side = np.uint8(np.concatenate([total[sample_idx[-1]].copy(), s_t_pred], axis=1) * 255.)        # :112
        total[sample_idx[-1]] = s_t_pred                             # :116
        if frame_idx == span:                                        # :118-120
            for i in range(span):
                total[i] = s_t_pred
        outs.append(s_t_pred.astype(F32))
        sbs.append(side)
        sample_idx = sample_idx + 1                                  # :124
    return np.stack(outs), np.stack(sbs)


def eval_train_clip(weights, unstab, stab, h, w, skip_length=(0, 16, 24, 28, 30, 31, 32)):
    """eval_train.py:137-165 on in-memory clips [N,h,w,3] float (RGB, /255, resized): the history
    is teacher-forced from the stable clip.  Returns stabilised frames [N-32,h,w,3] float32."""
    skip_length = np.array(skip_length)
    span = int(skip_length[-1] - skip_length[0])
    total_stab = np.array([np.asarray(f, dtype=np.float64) for f in stab])
    total_unstab = np.array([np.asarray(f, dtype=np.float64) for f in unstab])
    for i in range(span):                                           # :137-138
        total_unstab[i] = total_stab[i]
    net = StabNet(h, w)
    outs = []
    sample_idx = skip_length.copy()
    for frame_idx in range(span, len(total_unstab)):                # :146
        batch = total_unstab[sample_idx]                            # :148
        batch = np.expand_dims(np.concatenate(batch, axis=2), 0)    # :149
        s_t_pred = np.squeeze(net.run(weights, batch.astype(F32), batch[:, :, :, 18:].astype(F32))[0])  # :151-155
        total_unstab[sample_idx[-1]] = total_stab[sample_idx[-1]]   # :162
        outs.append(s_t_pred.astype(F32))
        sample_idx = sample_idx + 1                                 # :165
    return np.stack(outs)
