#!/usr/bin/env python3
"""Generates tests/golden/*.npz: outputs of the CPU oracle (oracle/) on seeded inputs.

The reference is a TF 1.11 program with no tests / fixtures and cannot be imported here, so
these vectors come from the oracle restatement, not from the reference ("parity unpinned",
oracle/__init__.py).  They pin the oracle against drift and give the GPU tests a committed
target.  Inputs are regenerated from seeds (tests/inputs.py), only outputs are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import inputs  # noqa: E402
from coupe.dvsg_amd.weights import make_synthetic_weights  # noqa: E402
from oracle import model as omodel  # noqa: E402
from oracle import networks as onet  # noqa: E402
from oracle import spatial_transformer as ost  # noqa: E402
from oracle import thin_plate_spline as otps  # noqa: E402
from oracle import warp_with_optical_flow as oflow  # noqa: E402

CASES = {
    "warps": dict(B=2, H=32, W=48),
    "locnet": dict(B=2, H=64, W=96),
    # N ~ 40 at tiny resolution, so that every window slot (offsets 0, 16, 24,
    # 28, 30, 31, 32; eval.py:101-124) reads a stabilised frame at least once (slot 0 from step 33 on)
    "clip": dict(N=40, H=32, W=48),
    # BASELINE.json configs[0]: one 256x256 window through the evaluation graph (eval.py path)
    "cfg0_256": dict(B=1, H=256, W=256),
    # eval_train.py:25-51,137-165: the teacher-forced loop through eval_train.py's OWN graph (masked CNN input), the
    # homography of every step drawn from a seed (inputs.mask_homographies) instead of tf.random_uniform
    "eval_train": dict(N=38, H=32, W=48),
}


def warps_case():
    c = CASES["warps"]
    B, H, W = c["B"], c["H"], c["W"]
    U = inputs.smooth_frames(1001, B, H, W)
    coord = inputs.v_src(B)
    vec = inputs.control_vectors(1002, B)
    out, xs, ys = otps.ThinPlateSpline(U, coord, vec, (H, W))
    T = otps.solve_system(coord, (coord + vec).astype(np.float32))
    flow = inputs.smooth_flow(1003, B, H, W)
    rng = np.random.default_rng(1004)
    th_aff = (np.array([1, 0, 0, 0, 1, 0], np.float32) + 0.1 * rng.standard_normal((B, 6))).astype(np.float32)
    th_proj = (np.array([1, 0, 0, 0, 1, 0, 0, 0]) + rng.uniform(-1, 1, (B, 8))
               * np.array([0.1, 0.1, 0.5, 0.1, 0.1, 0.5, 0.1, 0.1])).astype(np.float32)
    th_el = (0.05 * rng.standard_normal((B, 32))).astype(np.float32)
    el_out, el_x, el_y = ost.ElasticTransformer((H, W)).transform(U, th_el)
    return dict(
        tps_T=T, tps_out=out, tps_xs=xs, tps_ys=ys,
        flow_out=oflow.tf_warp(U, flow, H, W),
        theta_affine=th_aff, affine_out=ost.AffineTransformer((H, W)).transform(U, th_aff),
        theta_projective=th_proj, projective_out=ost.ProjectiveTransformer((H, W)).transform(U, th_proj),
        theta_elastic=th_el, elastic_out=el_out, elastic_xs=el_x, elastic_ys=el_y,
    )


def locnet_case(weights):
    c = CASES["locnet"]
    B, H, W = c["B"], c["H"], c["W"]
    x = inputs.window_frames(2001, B, H, W)
    taps = {}
    F = onet.localizationNet(x, 25, weights, taps=taps)
    s_t_pred, xs, ys = omodel.StabNet(H, W).run(weights, x, x[..., 18:], fetch=("s_t_pred", "x_offset_t", "y_offset_t"))
    return dict(F_t=F, pool5=taps["pool5"], conv1_mean=taps["conv1"].mean(axis=(1, 2)),
                block4_mean=taps["block4/unit_3"].mean(axis=(1, 2)), s_t_pred=s_t_pred,
                xs=xs.astype(np.float32), ys=ys.astype(np.float32))


def clip_case(weights):
    c = CASES["clip"]
    frames = inputs.smooth_frames(3001, c["N"], c["H"], c["W"])
    grids = []
    outs, side = omodel.eval_clip(weights, frames, c["H"], c["W"], grids=grids)
    # per step, the pixels whose source coordinate sits within 3e-2 px of one of sampler A's jumps (packed bits)
    mask = np.stack([otps.border_discontinuity_mask(xs, ys, c["H"], c["W"], delta=3e-2) for xs, ys in grids])
    return dict(stabilised=outs, side_by_side=side, border_mask_bits=np.packbits(mask))


def cfg0_case(weights):
    """configs[0]: a single 256x256 7-frame window, `sess.run([F_t, s_t_pred, x_offset_t, y_offset_t])`.
    The source grid is kept as the packed border-discontinuity mask it is needed for (sampler A's
    jump pixels, SURVEY.md section 7 hard part 3) plus a 1-in-16 subsample."""
    c = CASES["cfg0_256"]
    B, H, W = c["B"], c["H"], c["W"]
    x = inputs.window_frames(5001, B, H, W)
    F, s_t_pred, xs, ys = omodel.StabNet(H, W).run(weights, x, x[..., 18:],
                                                   fetch=("F_t", "s_t_pred", "x_offset_t", "y_offset_t"))
    mask = otps.border_discontinuity_mask(xs, ys, H, W, delta=3e-2)
    return dict(F_t=F, s_t_pred=s_t_pred, border_mask_bits=np.packbits(mask),
                xs_sub=xs[::16].astype(np.float32), ys_sub=ys[::16].astype(np.float32))


def eval_train_case(weights):
    c = CASES["eval_train"]
    N, H, W = c["N"], c["H"], c["W"]
    stab, unstab = inputs.stable_unstable_clips(6001, N, H, W)
    mask_H = inputs.mask_homographies(6002, N - 32)
    grids, f_t = [], []
    outs = omodel.eval_train_clip(weights, unstab, stab, H, W, mask_H, grids=grids, f_t=f_t)
    mask = np.stack([otps.border_discontinuity_mask(xs, ys, H, W, delta=3e-2) for xs, ys in grids])
    # what the round-3 loop computed instead (model.py's graph on the same windows): must DIFFER from the masked result
    plain = omodel.eval_train_clip(weights, unstab, stab, H, W, np.tile(omodel.RANDOM_MASK_OFFSET, (N - 32, 1)))
    return dict(stabilised=outs, F_t=np.stack(f_t), border_mask_bits=np.packbits(mask), mask_H=mask_H,
                stabilised_identity_mask=plain)


def main():
    weights = make_synthetic_weights(seed=0)
    np.savez_compressed(os.path.join(HERE, "eval_train.npz"), **eval_train_case(weights))
    np.savez_compressed(os.path.join(HERE, "warps.npz"), **warps_case())
    np.savez_compressed(os.path.join(HERE, "locnet.npz"), **locnet_case(weights))
    np.savez_compressed(os.path.join(HERE, "clip.npz"), **clip_case(weights))
    np.savez_compressed(os.path.join(HERE, "cfg0_256.npz"), **cfg0_case(weights))
    for f in ("warps.npz", "locnet.npz", "clip.npz", "cfg0_256.npz", "eval_train.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
