"""GPU: eval_train.py's evaluation graph (eval_train.py:25-51) -- the regressor's input is `patches_t * mask`
(:43-45), `random_mask` (:53-64 = model.py:156-167) being the projective warp of an all-ones image of the 18 history
channels; the warps sample the unmasked u_t (:48).  The product multiplies ONE mask plane [B,H,W] into the history
channels inside conv1's load stage (dvsg_stabilize_masked_f32, dvsg_stabilize_ring_masked_{f32,u8}); no
[B,H,W,21] product tensor exists.  Checked against the oracle's restatement of that graph (oracle/model.py
`EvalTrainNet`, `eval_train_clip`) with the homographies supplied instead of drawn, and against the committed golden."""
import os

import numpy as np
import pytest

import inputs
from oracle import model as omodel
from oracle import networks as onet
from oracle.thin_plate_spline import border_discontinuity_mask

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
IDENT = np.array([1, 0, 0, 0, 1, 0, 0, 0], np.float32)


@pytest.fixture(scope="module")
def net(synthetic_weights):
    import torch
    assert torch.cuda.is_available()
    from coupe.dvsg_amd.networks import LocNet
    return LocNet(synthetic_weights)


def _gather(pool, table):
    import torch
    from coupe.dvsg_amd import _lib
    B, (n, H, W, _) = table.shape[0], pool.shape
    out = torch.empty((B, H, W, 21), dtype=torch.float32, device=pool.device)
    _lib.call("dvsg_window_gather_f32", pool.data_ptr(), n, H, W, table.data_ptr(), B, 7, out.data_ptr(),
              torch.cuda.current_stream().cuda_stream)
    return out


def _strong_homographies(seed, B):
    """Homographies at up to three times eval_train.py's range: large parts of the mask are zero or fractional."""
    u = np.random.default_rng(seed).uniform(-1.0, 1.0, (B, 8)).astype(np.float32)
    return (u * 3.0 * omodel.RANDOM_MASK_SCALE + omodel.RANDOM_MASK_OFFSET).astype(np.float32)


# aligned rows, ragged rows, several 128-pixel column tiles, frames down to 1x1
SHAPES = [(2, 64, 96), (1, 37, 53), (3, 20, 4), (1, 8, 8), (1, 1, 1), (2, 30, 600), (1, 5, 301)]


@pytest.mark.parametrize("B,H,W", SHAPES + [(2, 288, 512)])
def test_mask_plane_is_the_projective_warp_of_ones(B, H, W):
    """dvsg_random_mask_plane_f32 against ProjectiveTransformer(out_size).transform(ones[B,H,W,18], H)[..., c] for every c
    (model.py:160-164): the same operations in the same order -- bit for bit but for the reciprocal of the division."""
    from coupe.dvsg_amd.networks import random_mask_plane
    for seed, Hm in ((1, inputs.mask_homographies(7000 + H, B)), (2, _strong_homographies(7100 + W, B)),
                     (3, np.tile(IDENT, (B, 1)))):
        got = random_mask_plane(Hm, H, W).cpu().numpy()
        _, rmask = omodel.random_mask(np.ones((B, H, W, 21), np.float32), (H, W), 7, Hm)
        for c in (0, 5, 17):
            assert np.abs(got - rmask[..., c]).max() <= 2e-6, (seed, c, np.abs(got - rmask[..., c]).max())
        assert got.min() >= 0.0 and got.max() <= 1.0 + 1e-6
        if seed == 3:
            assert np.all(got == 1.0)            # identity: exactly one everywhere, like the oracle's
        if seed == 2 and H * W > 64:
            assert (got == 0.0).any() or (got < 1.0).mean() > 0.05


@pytest.mark.parametrize("precision,tol", [("f32", 2e-5), ("f32s", 2e-5), ("f32x3", 2e-5), ("f16", 4e-3)])
@pytest.mark.parametrize("B,H,W", SHAPES)
def test_masked_conv1_matches_the_oracle_and_the_sources_agree(net, synthetic_weights, precision, tol, B, H, W):
    """conv1 (+ fused mask and scale_RGB) from a masked window against the oracle's conv1 of `patches * mask`; a float
    ring gives the gathered window's bits, a uint8 ring the bits of the float ring of the converted frames."""
    import torch
    from coupe.dvsg_amd import _lib
    from coupe.dvsg_amd.networks import random_mask_plane
    rng = np.random.default_rng(300 + H + W)
    pool8 = torch.from_numpy(rng.integers(0, 256, (9, H, W, 3), dtype=np.uint8)).cuda()
    table = torch.from_numpy(rng.integers(0, 9, (B, 7)).astype(np.int32)).cuda()
    if B > 1:
        table[1, 2] = 9           # outside the pool: a frame of zeros
    poolf = torch.empty(pool8.shape, dtype=torch.float32, device="cuda")
    _lib.call("dvsg_frames_u8_to_f32", pool8.data_ptr(), 9 * H * W, 0, poolf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    x = _gather(poolf, table)
    plane = random_mask_plane(_strong_homographies(400 + H, B), H, W)
    taps = {}
    masked = x.cpu().numpy().copy()
    masked[..., :18] *= plane.cpu().numpy()[..., None]                      # eval_train.py:62-64
    onet.localizationNet(masked, 25, synthetic_weights, taps=taps)
    want0 = taps["conv1"]
    got0 = net.forward_masked(x, plane, precision=precision, stage=0)
    scale = max(1.0, float(np.abs(want0).max()))
    assert np.abs(got0.cpu().numpy() - want0).max() <= tol * scale, np.abs(got0.cpu().numpy() - want0).max() / scale
    unmasked0 = net.tap(x, 0, precision=precision)
    if H * W > 64:
        assert not torch.equal(got0, unmasked0)                             # the mask really reached conv1
    for stage in (0, 1, -1):
        w = net.forward_masked(x, plane, precision=precision, stage=stage)
        rf = net.forward_masked(poolf, plane, table=table, precision=precision, stage=stage)
        r8 = net.forward_masked(pool8, plane, table=table, precision=precision, stage=stage)
        assert torch.equal(w, rf), "%s stage %d float ring: %g" % (precision, stage, float((w - rf).abs().max()))
        assert torch.equal(w, r8), "%s stage %d uint8 ring: %g" % (precision, stage, float((w - r8).abs().max()))


@pytest.mark.parametrize("precision", ["f32", "f16", "f32s", "f32x3"])
@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 37, 53), (2, 30, 600)])
def test_a_mask_of_ones_reproduces_the_unmasked_entry_points_bit_for_bit(net, precision, B, H, W):
    """(x * 1.0f) * 255 - mean == x * 255 - mean, and for uint8 frames float32(v / 255.) * 255 == float(v): the identity
    homography (whose plane is exactly one) gives today's outputs, from the window, the float ring and the uint8 ring."""
    import torch
    from coupe.dvsg_amd import _lib
    from coupe.dvsg_amd.networks import random_mask_plane
    rng = np.random.default_rng(500 + W)
    pool8 = torch.from_numpy(rng.integers(0, 256, (8, H, W, 3), dtype=np.uint8)).cuda()
    pool8[0] = torch.arange(256, dtype=torch.uint8).repeat(-(-H * W * 3 // 256))[:H * W * 3].reshape(H, W, 3)   # every byte value
    table = torch.from_numpy(rng.integers(0, 8, (B, 7)).astype(np.int32)).cuda()
    table[0, 0] = 0
    poolf = torch.empty(pool8.shape, dtype=torch.float32, device="cuda")
    _lib.call("dvsg_frames_u8_to_f32", pool8.data_ptr(), 8 * H * W, 0, poolf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    x = _gather(poolf, table)
    plane = random_mask_plane(np.tile(IDENT, (B, 1)), H, W)
    assert bool((plane == 1.0).all())
    u = torch.stack([poolf[int(table[b, 6])] for b in range(B)]).contiguous()
    outs = []
    for mask in (None, plane):
        o = torch.empty((B, H, W, 3), device="cuda")
        F = torch.empty((B, 25, 2), device="cuda")
        net.stabilize(x, u, o, F, precision=precision, mask=mask)
        outs.append((o, F))
        for pool in (poolf, pool8):
            o2 = torch.empty((B, H, W, 3), device="cuda")
            F2 = torch.empty((B, 25, 2), device="cuda")
            net.stabilize_ring(pool, table, o2, F2, precision=precision, mask=mask)
            outs.append((o2, F2))
    for o, F in outs[1:]:
        assert torch.equal(F, outs[0][1]) and torch.equal(o, outs[0][0])


def test_eval_train_graph_matches_the_oracle(synthetic_weights):
    """coupe.dvsg_amd.eval_train.get_evaluation_model (eval_train.py:25-51) with the graph's random H fed."""
    from coupe.dvsg_amd import eval_train
    B, H, W = 2, 64, 96
    x = inputs.window_frames(6201, B, H, W)
    Hm = inputs.mask_homographies(6202, B)
    ins, outs = eval_train.get_evaluation_model(7, 25, 5, H, W)
    assert list(ins) == ["patches_t", "u_t", "random_H"]
    assert list(outs) == ["V_src", "patches_masked_t", "random_masks_t", "F_t", "s_t_pred", "x_offset_t", "y_offset_t",
                          "s_t_pred_mask"]
    eval_train.model_of(outs).load_weights(synthetic_weights)
    sess = eval_train.Session()
    feed = {ins["patches_t"]: x, ins["u_t"]: x[..., 18:], ins["random_H"]: Hm}
    F, pred, xs, ys, pm, m, pmask = sess.run([outs[k] for k in ("F_t", "s_t_pred", "x_offset_t", "y_offset_t",
                                                                 "patches_masked_t", "random_masks_t", "s_t_pred_mask")], feed)
    rF, rpred, rx, ry, rpm, rm, rpmask = omodel.EvalTrainNet(H, W).run(
        synthetic_weights, x, x[..., 18:], Hm,
        fetch=("F_t", "s_t_pred", "x_offset_t", "y_offset_t", "patches_masked_t", "random_masks_t", "s_t_pred_mask"))
    assert np.abs(m - rm).max() <= 2e-6 and np.abs(pm - rpm).max() <= 2e-6 and (m[..., :18] < 1.0).any()
    assert np.abs(F - rF).max() <= 1e-5, np.abs(F - rF).max()
    border = border_discontinuity_mask(rx, ry, H, W, delta=3e-2).reshape(B, H, W)
    assert np.abs(pred - rpred).max(axis=3)[~border].max() < 1e-3
    assert np.abs(pmask - rpmask).max(axis=3)[~border].max() < 1e-3
    # F_t alone (no warp): the same prediction through dvsg_locnet_forward_masked
    F_only = sess.run(outs["F_t"], feed)
    assert np.array_equal(F_only, F)
    # the unmasked graph (model.py:98-123) predicts something else on this window
    from coupe.dvsg_amd.model import Session, StabNet
    plain = StabNet(H, W).load_weights(synthetic_weights)
    pi, po = plain.get_evaluation_model(7)
    Fp = Session().run(po["F_t"], {pi["patches_t"]: x, pi["u_t"]: x[..., 18:]})
    assert np.abs(Fp - F).max() > 1e-5
    # without the feed the graph draws H itself, as the reference does (tf.random_uniform): reproducible from a generator
    import torch
    model = eval_train.model_of(outs)
    model.mask_generator = torch.Generator().manual_seed(3)
    a = sess.run(outs["random_masks_t"], {ins["patches_t"]: x, ins["u_t"]: x[..., 18:]})
    model.mask_generator = torch.Generator().manual_seed(3)
    b = sess.run(outs["random_masks_t"], {ins["patches_t"]: x, ins["u_t"]: x[..., 18:]})
    assert np.array_equal(a, b) and np.all(a[..., 18:] == 1.0) and not np.array_equal(a, m)
    with pytest.raises(ValueError):
        eval_train.get_evaluation_model(7, 16, 4, H, W)


def test_teacher_forced_clip_is_eval_train_py(synthetic_weights):
    """eval_train.py:86,137-165 against the committed golden (tests/golden/eval_train.npz) and the oracle: every step
    through the MASKED graph; float and uint8-resident rings, two batch sizes."""
    from coupe.dvsg_amd.clip import stabilize_clip_teacher_forced
    from coupe.dvsg_amd.model import StabNet
    with np.load(os.path.join(GOLD, "eval_train.npz"), allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    N, H, W = 38, 32, 48
    stab, unstab = inputs.stable_unstable_clips(6001, N, H, W)
    model = StabNet(H, W).load_weights(synthetic_weights)
    border = np.unpackbits(g["border_mask_bits"])[:(N - 32) * H * W].astype(bool).reshape(N - 32, H, W)
    for batch in (4, 16):
        out = stabilize_clip_teacher_forced(model, unstab, stab, batch=batch, mask_H=g["mask_H"])
        assert out.shape == (N - 32, H, W, 3) and out.dtype == np.float32
        err = np.abs(out - g["stabilised"])
        assert err.max(axis=3)[~border].max() < 1e-3 and np.median(err) < 1e-5, (err.max(axis=3)[~border].max(), np.median(err))
    # an identity homography per step is model.py's graph on the same windows: bit-identical to mask_H=None
    ident = np.tile(IDENT, (N - 32, 1))
    a = stabilize_clip_teacher_forced(model, unstab, stab, batch=4, mask_H=ident)
    b = stabilize_clip_teacher_forced(model, unstab, stab, batch=4, mask_H=None)
    assert np.array_equal(a, b)
    assert np.abs(a - g["stabilised_identity_mask"]).max(axis=3)[~border].max() < 2e-2 and np.median(np.abs(a - g["stabilised_identity_mask"])) < 1e-5
    assert np.abs(a - out).max() > 1e-2                                          # and NOT what eval_train.py computes
    # uint8-resident clips (3 bytes per pixel in HBM, / 255. and the mask fused into conv1's staging)
    un8, st8 = (unstab * 255).astype(np.uint8), (stab * 255).astype(np.uint8)
    u8 = stabilize_clip_teacher_forced(model, un8, st8, batch=3, as_uint8=True, mask_H=g["mask_H"])
    ref8 = omodel.eval_train_clip(synthetic_weights, un8 / 255., st8 / 255., H, W, g["mask_H"])
    from oracle import frames as oframes
    diff = np.abs(u8.astype(int) - oframes.to_uint8(ref8).astype(int))
    assert u8.dtype == np.uint8 and (diff > 1).mean() < 0.01 and np.median(diff) == 0
    # the default draws the homographies as the reference does; a generator makes the draw reproducible
    import torch
    r1 = stabilize_clip_teacher_forced(model, unstab, stab, batch=4, mask_H=torch.Generator().manual_seed(11))
    r2 = stabilize_clip_teacher_forced(model, unstab, stab, batch=16, mask_H=torch.Generator().manual_seed(11))
    assert np.abs(r1 - r2).max() <= 1e-6 and np.abs(r1 - a).max() > 1e-3
    r3 = stabilize_clip_teacher_forced(model, unstab, stab, batch=4)
    assert r3.shape == a.shape and np.isfinite(r3).all()
    with pytest.raises(ValueError):
        stabilize_clip_teacher_forced(model, unstab, stab, mask_H=np.zeros((3, 8), np.float32))


@pytest.mark.parametrize("precision,f_tol", [("f32", 1e-5), ("f32s", 1e-5), ("f32x3", 1e-5), ("f16", 5e-5)])
def test_masked_graph_at_720p(synthetic_weights, precision, f_tol):
    """One 1280x720 window through the masked graph in each precision: F_t against the oracle (torch-CPU CNN on
    patches * mask), pixels < 1e-3 outside sampler A's counted border pixels."""
    import torch
    from coupe.dvsg_amd import eval_train
    from oracle.cnn_torch import TorchLocNet
    from oracle.thin_plate_spline import ThinPlateSpline as ostn
    B, H, W = 1, 720, 1280
    x = inputs.window_frames(6301, B, H, W)
    Hm = inputs.mask_homographies(6302, B)
    ins, outs = eval_train.get_evaluation_model(7, 25, 5, H, W)
    model = eval_train.model_of(outs)
    model.load_weights(synthetic_weights)
    model.precision = precision
    F, pred, m = eval_train.Session().run([outs["F_t"], outs["s_t_pred"], outs["random_masks_t"]],
                                          {ins["patches_t"]: x, ins["u_t"]: x[..., 18:], ins["random_H"]: Hm})
    _, rm = omodel.random_mask(np.ones((B, H, W, 21), np.float32), (H, W), 7, Hm)
    assert np.abs(m - rm).max() <= 2e-6 and (m < 1.0).mean() > 0.01
    rF = TorchLocNet(synthetic_weights).forward(x * rm)
    assert np.abs(F - rF).max() <= f_tol, np.abs(F - rF).max()
    rpred, rx, ry = ostn(x[..., 18:], np.tile(omodel.v_src()[None], (B, 1, 1)), rF, (H, W))
    border = border_discontinuity_mask(rx, ry, H, W, delta=3e-2).reshape(B, H, W)
    err = np.abs(pred - rpred).max(axis=3)[~border]
    print("masked 720p %s: F_t %.2e, pixels %.2e (border pixels excluded: %d)" % (precision, np.abs(F - rF).max(), err.max(), border.sum()))
    assert err.max() < 1e-3
