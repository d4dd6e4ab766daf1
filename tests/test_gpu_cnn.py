"""GPU parity: localizationNet (HIP, exact-f32 MFMA) against the NumPy oracle, stage by
stage and end to end, then the fused evaluation graph behind the StabNet facade."""
import numpy as np
import pytest

import inputs
from oracle import model as omodel
from oracle import networks as onet
from oracle import thin_plate_spline as otps

pytestmark = pytest.mark.gpu

STAGES = ["conv1", "pool1"] + ["%s/unit_%d" % (b, u) for b, n in
                                (("block1", 3), ("block2", 4), ("block3", 6), ("block4", 3))
                                for u in range(1, n + 1)] + ["pool5"]


@pytest.fixture(scope="module")
def net(synthetic_weights):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from coupe.dvsg_amd.networks import LocNet
    return LocNet(synthetic_weights)


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 70, 100), (3, 33, 47)])
def test_every_stage_matches_oracle(net, synthetic_weights, B, H, W):
    """Float32 on both sides; the only difference is summation order (MFMA k-order chain vs
    BLAS), so the bound is a small multiple of eps * sqrt(K) relative to the activation
    scale, growing slowly with depth."""
    x = inputs.window_frames(201, B, H, W)
    taps = {}
    F_ref = onet.localizationNet(x, 25, synthetic_weights, taps=taps)
    for stage, name in enumerate(STAGES):
        act = net.tap(x, stage).cpu().numpy()
        ref = taps[name]
        if name == "pool5":
            ref = ref.reshape(B, 1, 1, 2048)
        assert act.shape == ref.shape, name
        scale = np.abs(ref).max()
        err = np.abs(act - ref).max()
        assert err <= 2e-5 * scale, "%s: err %.3g scale %.3g" % (name, err, scale)
    F = net.forward(x).cpu().numpy()
    assert F.shape == (B, 25, 2)
    assert np.abs(F - F_ref).max() <= 1e-5     # SURVEY.md 8d parity target for the f32 path


def test_head_batch_chunks(net, synthetic_weights):
    """B > 16 exercises the chunked dense head."""
    B, H, W = 19, 32, 32
    x = inputs.window_frames(211, B, H, W)
    F = net.forward(x).cpu().numpy()
    F_ref = onet.localizationNet(x, 25, synthetic_weights)
    assert np.abs(F - F_ref).max() <= 1e-5
    # and samples are independent of batch composition
    F1 = net.forward(x[3:4]).cpu().numpy()
    assert np.abs(F1 - F[3:4]).max() <= 1e-6


def test_localization_net_facade(net, synthetic_weights):
    from coupe.dvsg_amd import DvsgError
    from coupe.dvsg_amd import networks
    x = inputs.window_frames(221, 1, 48, 64)
    with pytest.raises(DvsgError):
        networks.localizationNet(x, 25, scope="nobody/loaded/this")
    networks.load_localizationNet(synthetic_weights, scope="stabNet/localizationNet")
    F = networks.localizationNet(x, 25, False, False, scope="stabNet/localizationNet")
    assert isinstance(F, np.ndarray) and F.shape == (1, 25, 2)
    assert np.abs(F - onet.localizationNet(x, 25, synthetic_weights)).max() <= 1e-5
    with pytest.raises(NotImplementedError):
        networks.localizationNet(x, 25, is_train=True)


def test_missing_weight_is_an_error(synthetic_weights):
    from coupe.dvsg_amd.networks import LocNet
    w = dict(synthetic_weights)
    del w["stabNet/localizationNet/resnet_v1_50/block3/unit_2/bottleneck_v1/conv2/weights:0"]
    with pytest.raises(ValueError, match="missing"):
        LocNet(w)


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 72, 128)])
def test_stabnet_evaluation_model(synthetic_weights, B, H, W):
    """model.py:98-123 behind the reference call surface, eval.py:106-110 style."""
    from coupe.dvsg_amd.model import Session, StabNet
    x = inputs.window_frames(231, B, H, W)
    u = x[..., 18:]
    model = StabNet(H, W).load_weights(synthetic_weights)
    ins, outs = model.get_evaluation_model(7)
    assert list(ins) == ["patches_t", "u_t"]
    assert list(outs) == ["V_src", "num_control_points", "F_t", "s_t_pred", "x_offset_t", "y_offset_t",
                          "s_t_pred_mask"]
    sess = Session()
    feed = {ins["patches_t"]: x, ins["u_t"]: u}
    s_t_pred = sess.run(outs["s_t_pred"], feed)
    ref = omodel.StabNet(H, W).run(synthetic_weights, x, u,
                                   fetch=("s_t_pred", "F_t", "x_offset_t", "y_offset_t", "s_t_pred_mask", "V_src"))
    r_pred, r_F, r_x, r_y, r_mask, r_V = ref
    assert s_t_pred.shape == (B, H, W, 3) and isinstance(s_t_pred, np.ndarray)
    got = sess.run([outs["F_t"], outs["x_offset_t"], outs["y_offset_t"], outs["s_t_pred_mask"], outs["V_src"],
                    outs["num_control_points"]], feed)
    F, xs, ys, mask, V, ncp = got
    assert ncp == 5 and np.array_equal(V, r_V)
    assert np.abs(F - r_F).max() <= 1e-5
    gerr = max(np.abs(xs - r_x).max() * W / 2, np.abs(ys - r_y).max() * H / 2)
    assert gerr < 2e-2, "grid error %.3g px" % gerr
    border = otps.border_discontinuity_mask(r_x, r_y, H, W, delta=3e-2).reshape(B, H, W)
    err = np.abs(s_t_pred - r_pred).max(axis=3)
    assert err[~border].max() < 3e-3      # end to end (includes F_t and float32-LU differences)
    merr = np.abs(mask - r_mask).max(axis=3)
    assert merr[~border].max() < 3e-3
    with pytest.raises(ValueError):
        sess.run(outs["s_t_pred"], {ins["patches_t"]: x[:, :-1], ins["u_t"]: u[:, :-1]})
