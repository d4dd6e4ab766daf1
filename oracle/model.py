"""Oracle (test infrastructure): model.py `StabNet.get_evaluation_model` and the eval.py
frame loop restated in NumPy float32.  parity unpinned (see oracle/__init__.py).

Reference: model.py:14-25 (ctor), model.py:98-123 (evaluation graph), eval.py:93-124
(autoregressive clip loop); eval_train.py:25-51 (ITS OWN evaluation graph: the CNN sees
`patches_t * mask`), :53-64 (`random_mask`), :137-165 (teacher-forced clip loop).
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


RANDOM_MASK_SCALE = np.array([0.1, 0.1, 0.5, 0.1, 0.1, 0.5, 0.1, 0.1], dtype=F32)     # model.py:162 / eval_train.py:56
RANDOM_MASK_OFFSET = np.array([1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], dtype=F32)    # model.py:163 / eval_train.py:57


def random_mask_H(uniform):
    """model.py:161-163 / eval_train.py:55-57 after the draw: `uniform` [B,8] in [-1,1) stands in for
    tf.random_uniform (the graph's only stochastic op; the caller owns the generator), H = u * scale + identity
    in float32."""
    u = np.asarray(uniform, dtype=F32)
    return ((u * RANDOM_MASK_SCALE).astype(F32) + RANDOM_MASK_OFFSET).astype(F32)


def random_mask(patches, out_size, sample_num, H):
    """model.py:156-167 = eval_train.py:53-64 with the homography parameters H [B,8] supplied by the caller:
    H is the value AFTER lines 162-163 (`random_mask_H` applies them to a uniform draw)."""
    from .spatial_transformer import ProjectiveTransformer
    patches = np.asarray(patches, dtype=F32)
    c_hist = 3 * (sample_num - 1)
    mask = np.ones_like(patches[..., :c_hist])                                   # :160
    mask = ProjectiveTransformer(out_size).transform(mask, np.asarray(H, F32))    # :164
    mask = np.concatenate([mask, np.ones_like(patches[..., :3])], axis=3)        # :165
    return (patches * mask).astype(F32), mask.astype(F32)                        # :167


class EvalTrainNet:
    """eval_train.py:25-51 `get_evaluation_model(sample_num, param_dim, num_control_points, h, w)`: NOT model.py's
    graph -- the regressor's input is `patches_masked_t` (:43-45); the warps sample the unmasked u_t (:48-49)."""

    def __init__(self, h, w, sample_num=7, num_control_points=5):
        self.h, self.w, self.sample_num = h, w, sample_num
        self.num_control_points = num_control_points
        self.param_dim = num_control_points ** 2

    def run(self, weights, patches_t, u_t, H, fetch=("s_t_pred",), taps=None):
        """`sess.run([outputs[k] for k in fetch], {patches_t, u_t})` with the graph's random H [B,8] supplied."""
        patches_t = np.asarray(patches_t, dtype=F32)
        u_t = np.asarray(u_t, dtype=F32)
        B = u_t.shape[0]
        out = collections.OrderedDict()
        out["V_src"] = np.tile(v_src(self.num_control_points)[None], (B, 1, 1))                       # :33-39
        out["patches_masked_t"], out["random_masks_t"] = random_mask(patches_t, [self.h, self.w],
                                                                     self.sample_num, H)               # :43
        out["F_t"] = localizationNet(out["patches_masked_t"], self.param_dim, weights, taps=taps)       # :45
        if any(k in fetch for k in ("s_t_pred", "x_offset_t", "y_offset_t")):
            out["s_t_pred"], out["x_offset_t"], out["y_offset_t"] = stn(
                u_t, out["V_src"], out["F_t"], [self.h, self.w])                                      # :48
        if "s_t_pred_mask" in fetch:
            out["s_t_pred_mask"], _, _ = stn(np.ones_like(u_t), out["V_src"], out["F_t"], [self.h, self.w])  # :49
        return [out[k] for k in fetch]


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


def eval_train_clip(weights, unstab, stab, h, w, mask_H, skip_length=(0, 16, 24, 28, 30, 31, 32), grids=None,
                    f_t=None):
    """eval_train.py:86,137-165 on in-memory clips [N,h,w,3] float (RGB, /255, resized): the history is
    teacher-forced from the stable clip and every step runs eval_train.py's OWN graph (:25-51, `EvalTrainNet`),
    whose CNN input is masked.  mask_H [N-32,8]: the homography of step k (the graph draws a fresh one per
    `sess.run`; `random_mask_H` makes one from a uniform draw).  Returns stabilised frames [N-32,h,w,3] float32;
    lists passed as `grids` / `f_t` receive each step's source grid / F_t."""
    skip_length = np.array(skip_length)
    span = int(skip_length[-1] - skip_length[0])
    total_stab = np.array([np.asarray(f, dtype=np.float64) for f in stab])
    total_unstab = np.array([np.asarray(f, dtype=np.float64) for f in unstab])
    mask_H = np.asarray(mask_H, dtype=F32).reshape(len(total_unstab) - span, 8)
    for i in range(span):                                           # :137-138
        total_unstab[i] = total_stab[i]
    net = EvalTrainNet(h, w, len(skip_length))                      # :86
    outs = []
    sample_idx = skip_length.copy()
    for frame_idx in range(span, len(total_unstab)):                # :146
        batch = total_unstab[sample_idx]                            # :148
        batch = np.expand_dims(np.concatenate(batch, axis=2), 0)    # :149
        s_t_pred, xs, ys, F = net.run(weights, batch.astype(F32), batch[:, :, :, 18:].astype(F32),
                                      mask_H[frame_idx - span][None],
                                      fetch=("s_t_pred", "x_offset_t", "y_offset_t", "F_t"))   # :151-155
        s_t_pred = np.squeeze(s_t_pred)
        if grids is not None:
            grids.append((xs, ys))
        if f_t is not None:
            f_t.append(F[0])
        total_unstab[sample_idx[-1]] = total_stab[sample_idx[-1]]   # :162
        outs.append(s_t_pred.astype(F32))
        sample_idx = sample_idx + 1                                 # :165
    return np.stack(outs)
