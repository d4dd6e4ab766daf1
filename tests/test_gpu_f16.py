"""GPU: the float16 variant of localizationNet (BASELINE.json configs[4] "fp16 MFMA convs").
float16 storage cannot be bit-compatible with the float32 reference; this file STATES the measured
error against the float32 oracle (tolerances are ~3x the values observed on MI355X) and checks the
properties that must still hold exactly."""
import numpy as np
import pytest

import inputs
from oracle import networks as onet

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net(synthetic_weights):
    import torch
    assert torch.cuda.is_available()
    from coupe.dvsg_amd.networks import LocNet
    return LocNet(synthetic_weights)


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 70, 100)])
def test_f16_stages_track_the_f32_oracle(net, synthetic_weights, B, H, W):
    x = inputs.window_frames(301, B, H, W)
    taps = {}
    F_ref = onet.localizationNet(x, 25, synthetic_weights, taps=taps)
    names = ["conv1", "pool1", "block1/unit_3", "block2/unit_4", "block3/unit_6", "block4/unit_3"]
    stages = [0, 1, 4, 8, 14, 17]
    for stage, name in zip(stages, names):
        act = net.tap(x, stage, precision="f16").cpu().numpy()
        ref = taps[name]
        assert act.shape == ref.shape
        rel = np.abs(act - ref).max() / np.abs(ref).max()
        # float16 has an 11-bit significand: ~5e-4 per rounding, accumulated over up to 50 layers
        assert rel < 2e-2, "%s: relative error %.3g" % (name, rel)
    F = net.forward(x, precision="f16").cpu().numpy()
    err = np.abs(F - F_ref).max()
    assert err < 2e-3, "F_t error %.3g (|F| ~ %.3g)" % (err, np.abs(F_ref).max())
    F32 = net.forward(x, precision="f32").cpu().numpy()
    assert np.abs(F32 - F_ref).max() <= 1e-5          # the f32 path is untouched by the f16 weights


def test_f16_stabilize_and_determinism(net, synthetic_weights):
    import torch
    from coupe.dvsg_amd.model import Session, StabNet
    from oracle import model as omodel
    from oracle.thin_plate_spline import border_discontinuity_mask
    B, H, W = 2, 72, 128
    x = inputs.window_frames(311, B, H, W)
    u = x[..., 18:]
    model = StabNet(H, W).load_weights(synthetic_weights)
    model.precision = "f16"
    ins, outs = model.get_evaluation_model(7)
    feed = {ins["patches_t"]: x, ins["u_t"]: u}
    got, F, xs, ys = Session().run([outs["s_t_pred"], outs["F_t"], outs["x_offset_t"], outs["y_offset_t"]], feed)
    again = Session().run(outs["s_t_pred"], feed)
    assert np.array_equal(got, again)                  # deterministic
    ref, rF, rx, ry = omodel.StabNet(H, W).run(synthetic_weights, x, u, fetch=("s_t_pred", "F_t", "x_offset_t", "y_offset_t"))
    gerr = max(np.abs(xs - rx).max() * W / 2, np.abs(ys - ry).max() * H / 2)
    mask = border_discontinuity_mask(rx, ry, H, W, delta=0.2).reshape(B, H, W)
    perr = np.abs(got - ref).max(axis=3)[~mask].max()
    # stated values for the float16 path (float32 path: < 2e-2 px, < 3e-3)
    assert gerr < 0.2, "grid error %.3g px" % gerr
    assert perr < 3e-2, "pixel error %.3g" % perr


def test_conv_gemm_f16_layer(net):
    """One 3x3 and one 1x1 layer through dvsg_conv_gemm_f16 against float32 math on the same
    (float16-rounded) operands: only the float32 accumulation order differs."""
    import torch
    from coupe.dvsg_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    for k, stride, cin, cout, h, w in [(3, 1, 64, 64, 20, 28), (3, 2, 128, 128, 21, 17), (1, 1, 256, 64, 9, 13)]:
        B = 2
        x = (torch.rand((B, h, w, cin), generator=g, device=dev) - 0.3).half()
        K = k * k * cin
        wt = ((torch.rand((cout, K), generator=g, device=dev) - 0.5) * (2.0 / K ** 0.5)).half()
        bias = torch.rand((cout,), generator=g, device=dev) - 0.5
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        res = (torch.rand((B, ho, wo, cout), generator=g, device=dev) - 0.5).half()
        y = torch.empty((B, ho, wo, cout), device=dev, dtype=torch.float16)
        _lib.call("dvsg_conv_gemm_f16", x.data_ptr(), wt.data_ptr(), bias.data_ptr(), res.data_ptr(), y.data_ptr(),
                  B, h, w, cin, cout, k, stride, 1, 1, 0, 0, torch.cuda.current_stream().cuda_stream)
        w4 = wt.float().reshape(cout, k, k, cin).permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w4, bias, stride=stride, padding=k // 2)
        ref = torch.relu(ref.permute(0, 2, 3, 1) + res.float())
        err = float((y.float() - ref).abs().max())
        assert err < 4e-3 * max(1.0, float(ref.abs().max())), (k, stride, cin, err)
