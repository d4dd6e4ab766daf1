#!/usr/bin/env python3
"""Parity numbers of SURVEY.md 8d at 1280x720 (B=2 windows): F_t, source grid and warped pixels of
the HIP path against the CPU oracle (torch-CPU CNN + NumPy TPS), end to end and with the oracle's F_t
fed to the GPU warp (so that the warp is judged on identical coefficients)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs  # noqa: E402
from coupe.dvsg_amd.ThinPlateSpline import ThinPlateSpline  # noqa: E402
from coupe.dvsg_amd.model import Session, StabNet  # noqa: E402
from coupe.dvsg_amd.weights import make_synthetic_weights  # noqa: E402
from oracle import thin_plate_spline as otps  # noqa: E402
from oracle.cnn_torch import TorchLocNet  # noqa: E402

B, H, W = 2, 720, 1280
weights = make_synthetic_weights(seed=0)
x = inputs.window_frames(7, B, H, W)
u = np.ascontiguousarray(x[..., 18:])
for prec in ("f32", "f16"):
    model = StabNet(H, W).load_weights(weights)
    model.precision = prec
    ins, outs = model.get_evaluation_model(7)
    pred, F, xs, ys = Session().run([outs["s_t_pred"], outs["F_t"], outs["x_offset_t"], outs["y_offset_t"]],
                                    {ins["patches_t"]: x, ins["u_t"]: u})
    if prec == "f32":
        F_ref = TorchLocNet(weights).forward(x)
        V = inputs.v_src(B)
        r_pred, r_xs, r_ys = otps.ThinPlateSpline(u, V, F_ref, (H, W))
        border = otps.border_discontinuity_mask(r_xs, r_ys, H, W, delta=3e-2).reshape(B, H, W)
        g_pred, g_xs, g_ys = ThinPlateSpline(u, V, F_ref, (H, W))     # GPU warp on the oracle's coefficients
        gerr = max(np.abs(g_xs - r_xs).max() * W / 2, np.abs(g_ys - r_ys).max() * H / 2)
        perr = np.abs(g_pred - r_pred).max(axis=3)
        print("warp on identical F_t : grid %.2e px, pixels %.2e (outside the %d border-discontinuity pixels of %d; %.2e with them)"
              % (gerr, perr[~border].max(), int(border.sum()), border.size, perr.max()))
    gerr = max(np.abs(xs - r_xs).max() * W / 2, np.abs(ys - r_ys).max() * H / 2)
    perr = np.abs(pred - r_pred).max(axis=3)
    print("%s end to end        : F_t %.2e, grid %.2e px, pixels %.2e outside the border pixels (%.2e with them), median %.1e"
          % (prec, np.abs(F - F_ref).max(), gerr, perr[~border].max(), perr.max(), np.median(perr)))
