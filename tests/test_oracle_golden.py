"""The committed golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py
from the oracle) must be reproduced by the oracle on any machine: pins the oracle against
drift and checks the eval.py harness restatement."""
import os

import numpy as np
import pytest

import inputs
from oracle import model as omodel
from oracle import networks as onet
from oracle import spatial_transformer as ost
from oracle import thin_plate_spline as otps
from oracle import warp_with_optical_flow as oflow

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with np.load(os.path.join(GOLD, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def test_warps_golden():
    g = _load("warps.npz")
    B, H, W = 2, 32, 48
    U = inputs.smooth_frames(1001, B, H, W)
    coord, vec = inputs.v_src(B), inputs.control_vectors(1002, B)
    out, xs, ys = otps.ThinPlateSpline(U, coord, vec, (H, W))
    # LAPACK's float32 inverse may differ across builds by its own rounding noise
    assert np.abs(xs - g["tps_xs"]).max() < 2e-5 and np.abs(ys - g["tps_ys"]).max() < 2e-5
    mask = otps.border_discontinuity_mask(g["tps_xs"], g["tps_ys"], H, W, delta=3e-2).reshape(B, H, W)
    assert np.abs(out - g["tps_out"]).max(axis=3)[~mask].max() < 1e-3
    assert np.array_equal(oflow.tf_warp(U, inputs.smooth_flow(1003, B, H, W), H, W), g["flow_out"])
    assert np.abs(ost.AffineTransformer((H, W)).transform(U, g["theta_affine"]) - g["affine_out"]).max() < 1e-6
    assert np.abs(ost.ProjectiveTransformer((H, W)).transform(U, g["theta_projective"])
                  - g["projective_out"]).max() < 1e-5
    eo, ex, ey = ost.ElasticTransformer((H, W)).transform(U, g["theta_elastic"])
    assert np.abs(ex - g["elastic_xs"]).max() < 2e-5 and np.abs(eo - g["elastic_out"]).max() < 1e-3


def test_locnet_golden(synthetic_weights):
    g = _load("locnet.npz")
    x = inputs.window_frames(2001, 2, 64, 96)
    taps = {}
    F = onet.localizationNet(x, 25, synthetic_weights, taps=taps)
    assert np.abs(F - g["F_t"]).max() < 5e-6
    assert np.abs(taps["pool5"] - g["pool5"]).max() < 1e-4 * np.abs(g["pool5"]).max()


def test_clip_golden(synthetic_weights):
    """N = 40: long enough for every window slot to read stabilised history."""
    g = _load("clip.npz")
    N = 40
    frames = inputs.smooth_frames(3001, N, 32, 48)
    outs, side = omodel.eval_clip(synthetic_weights, frames, 32, 48)
    assert outs.shape == g["stabilised"].shape == (N, 32, 48, 3)
    assert side.shape == (N, 32, 96, 3) and side.dtype == np.uint8
    # the recurrence feeds LAPACK's float32-inverse noise back 40 times; pixels on sampler A's border
    # jump may flip (counted, not bounded)
    diff = np.abs(outs - g["stabilised"])
    assert np.median(diff) < 1e-6 and (diff > 5e-3).mean() < 1e-3
    assert np.array_equal(side, g["side_by_side"]) or (side != g["side_by_side"]).mean() < 1e-3
    # left half of the side-by-side is the (unstabilised) input frame, truncating cast
    assert np.array_equal(side[0, :, :48], np.uint8(frames[0].astype(np.float64) * 255.))


def test_cfg0_golden(synthetic_weights):
    """BASELINE.json configs[0]: one 256x256 window through the evaluation graph on the CPU."""
    g = _load("cfg0_256.npz")
    H = W = 256
    x = inputs.window_frames(5001, 1, H, W)
    F, pred, xs, ys = omodel.StabNet(H, W).run(synthetic_weights, x, x[..., 18:],
                                               fetch=("F_t", "s_t_pred", "x_offset_t", "y_offset_t"))
    assert F.shape == (1, 25, 2) and pred.shape == (1, H, W, 3)
    assert np.abs(F - g["F_t"]).max() < 5e-6
    assert np.abs(xs[::16] - g["xs_sub"]).max() < 2e-5 and np.abs(ys[::16] - g["ys_sub"]).max() < 2e-5
    mask = np.unpackbits(g["border_mask_bits"])[:H * W].astype(bool).reshape(1, H, W)
    assert 0 < mask.sum() < 0.02 * H * W
    assert np.abs(pred - g["s_t_pred"]).max(axis=3)[~mask].max() < 1e-3


def test_eval_train_golden(synthetic_weights):
    """eval_train.py:25-51,137-165: the teacher-forced loop through eval_train.py's own graph -- the CNN sees the
    window times the random projective mask (:43-45), the warp the unmasked frame (:48)."""
    g = _load("eval_train.npz")
    N, H, W = 38, 32, 48
    stab, unstab = inputs.stable_unstable_clips(6001, N, H, W)
    assert np.array_equal(g["mask_H"], inputs.mask_homographies(6002, N - 32))
    f_t = []
    outs = omodel.eval_train_clip(synthetic_weights, unstab, stab, H, W, g["mask_H"], f_t=f_t)
    assert outs.shape == g["stabilised"].shape == (N - 32, H, W, 3)
    assert np.abs(np.stack(f_t) - g["F_t"]).max() < 5e-6
    mask = np.unpackbits(g["border_mask_bits"])[:(N - 32) * H * W].astype(bool).reshape(N - 32, H, W)
    assert np.abs(outs - g["stabilised"]).max(axis=3)[~mask].max() < 1e-3
    # the mask matters: model.py's graph on the same windows (what an unmasked loop computes) gives other frames
    assert np.abs(g["stabilised"] - g["stabilised_identity_mask"]).max() > 1e-2


def test_eval_train_graph_is_the_masked_graph(synthetic_weights):
    """EvalTrainNet (eval_train.py:25-51) = StabNet's graph (model.py:98-123) on patches * mask, warp on the unmasked u_t."""
    B, H, W = 2, 32, 48
    x = inputs.window_frames(6101, B, H, W)
    Hm = inputs.mask_homographies(6102, B)
    F, pm, m, pred = omodel.EvalTrainNet(H, W).run(synthetic_weights, x, x[..., 18:], Hm,
                                                    fetch=("F_t", "patches_masked_t", "random_masks_t", "s_t_pred"))
    assert np.array_equal(pm, x * m) and np.all(m[..., 18:] == 1.0) and (m[..., :18] < 1.0).any()
    for c in range(1, 18):
        assert np.array_equal(m[..., c], m[..., 0])          # ONE plane: what the GPU path multiplies in conv1's load stage
    F2, pred2 = omodel.StabNet(H, W).run(synthetic_weights, pm, x[..., 18:], fetch=("F_t", "s_t_pred"))
    assert np.array_equal(F, F2) and np.array_equal(pred, pred2)
    F3 = omodel.StabNet(H, W).run(synthetic_weights, x, x[..., 18:], fetch=("F_t",))[0]
    assert np.abs(F - F3).max() > 1e-5                       # and it is not the unmasked prediction
    # identity homography: the mask is exactly one everywhere (sampler B on the unit grid), i.e. model.py's graph
    ident = np.tile(omodel.RANDOM_MASK_OFFSET, (B, 1))
    Fi, mi = omodel.EvalTrainNet(H, W).run(synthetic_weights, x, x[..., 18:], ident, fetch=("F_t", "random_masks_t"))
    assert np.all(mi == 1.0) and np.array_equal(Fi, F3)
