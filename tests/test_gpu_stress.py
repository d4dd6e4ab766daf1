"""GPU: the CNN on a STRESS checkpoint (oracle/stress_weights.py): BatchNorm gamma from 0 to 2.5, moving variances over
decades (calibrated to per-channel convolution scales spanning 1e-3 .. 1), dead channels, all-zero filters -- the
operand magnitudes of a trained checkpoint instead of the friendly O(0.03 .. 1) of `make_synthetic_weights`, on which
every other parity figure rides.  Such a network is ill-conditioned in float32 itself: channels whose moving variance
is tiny amplify rounding noise, and the torch-CPU float32 oracle sits 1.7e-5 (64x96) from a float64 evaluation of the
same graph.  So the yardstick is that float64 ARBITER (oracle/cnn_torch.py `TorchLocNet(dtype=float64)`): F_t of each
precision RELATIVE to max |F_t| (the stress head predicts displacements of O(1), not O(0.05)); the exact-float32 path
may be no further from it than 2 x the float32 oracle's own distance; the other bounds are ~3x the values measured on
MI355X (round 4) and are what the reduced precisions cost on such a checkpoint -- more than on the friendly one."""
import numpy as np
import pytest

import inputs

pytestmark = pytest.mark.gpu

# precision -> (relative F_t bound at 64x96, at 288x512, relative pool5 bound)
# measured (MI355X, round 4; the float32 ORACLE is 1.6e-5 / 2.0e-6 from the arbiter): f32 2.4e-5 / 2.4e-6, f32s 2.0e-5 / 2.1e-6
# -- its 22-bit products cost nothing here --, f16 1.2e-2 / 2.4e-3 (pool5 1.8e-2 / 3.1e-3): float16 ACTIVATIONS (11 bits)
# through channels whose folded BatchNorm scale is large; at 720p |F_t| ~ 1 means 1.2e-2 x 640 = 8 px -- on a checkpoint
# like this one the float16 mode does not meet the 1e-3 pixel tolerance it meets on the friendly one.
BOUNDS = {"f32": (None, None, None), "f32x3": (None, None, None), "f32s": (6e-5, 8e-6, 6e-5), "f16": (4e-2, 8e-3, 6e-2)}


@pytest.fixture(scope="module")
def stress():
    from coupe.dvsg_amd.networks import LocNet
    import torch
    from oracle.cnn_torch import TorchLocNet
    from oracle.stress_weights import make_stress_weights
    w = make_stress_weights(seed=0)
    return w, LocNet(w), TorchLocNet(w), TorchLocNet(w, dtype=torch.float64)


def test_the_stress_checkpoint_is_one(stress):
    """What the generator promises: gamma in {0} + [0.05, 2.5], moving variances over >= 4 decades, BatchNorm-folded
    weights whose per-layer median goes down to ~1e-3 and below, and a network that still predicts something."""
    w, _, ref, _ = stress
    gam = np.concatenate([v for k, v in w.items() if k.endswith("gamma:0")])
    var = np.concatenate([v for k, v in w.items() if k.endswith("moving_variance:0")])
    assert (gam == 0).mean() > 0.03 and gam.max() > 2.4 and gam[gam > 0].min() < 0.06
    live = var[var > 1e-11]
    assert np.log10(np.percentile(live, 99) / np.percentile(live, 1)) >= 4.0
    meds = []
    for _, (wt, sc, _) in ref.convs.items():
        fw = (wt * sc.view(-1, 1, 1, 1)).abs()
        meds.append(float(fw[fw > 0].median()))
    assert min(meds) < 2e-3 and max(meds) > 1e-2
    F = ref.forward(inputs.window_frames(8101, 2, 64, 96))
    assert 0.05 < np.abs(F).max() < 50 and np.isfinite(F).all()


@pytest.mark.parametrize("H,W,seed", [(64, 96, 8101), (288, 512, 8102)])
def test_stress_checkpoint_f_t_in_every_precision(stress, H, W, seed):
    import torch
    _, net, ref32, ref64 = stress
    x = inputs.window_frames(seed, 2, H, W)
    rF = ref64.forward(x)
    rpool = ref64.features(x).numpy()
    fnorm, pnorm = np.abs(rF).max(), np.abs(rpool).max()
    o_rel = np.abs(ref32.forward(x) - rF).max() / fnorm
    o_prel = np.abs(ref32.features(x).numpy() - rpool).max() / pnorm
    print("stress %dx%d float32 oracle vs float64 arbiter: F_t rel %.3g, pool5 rel %.3g (|F_t| max %.3g)" % (H, W, o_rel, o_prel, fnorm))
    xt = torch.from_numpy(x).cuda()
    failures = []
    for precision, (b_small, b_big, b_pool) in BOUNDS.items():
        F = net.forward(xt, precision=precision).cpu().numpy()
        pool = net.tap(xt, 18, precision=precision).cpu().numpy().reshape(rpool.shape)
        rel = np.abs(F - rF).max() / fnorm
        prel = np.abs(pool - rpool).max() / pnorm
        print("stress %dx%d %s vs float64 arbiter: F_t rel %.3g (abs %.3g), pool5 rel %.3g" % (H, W, precision, rel, np.abs(F - rF).max(), prel))
        if precision in ("f32", "f32x3"):   # f32x3 is held to the exact path's bound: within twice the float32 oracle's own error
            if not (rel <= 2.0 * o_rel + 1e-6 and prel <= 2.0 * o_prel + 1e-6):
                failures.append((precision, rel, prel))
        elif not (rel < (b_small if H < 100 else b_big) and prel < (b_pool if H < 100 else b_pool / 4)):
            failures.append((precision, rel, prel))
        assert np.array_equal(F, net.forward(xt, precision=precision).cpu().numpy())       # deterministic here too
    assert not failures, failures
