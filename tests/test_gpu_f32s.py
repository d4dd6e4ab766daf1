"""GPU: the "f32s" precision -- float32 storage and accumulation, every product of the 52 1x1 / 3x3
convolutions formed on the float16 matrix cores from two float16 pieces per operand (22 significant
bits for |x| >= 2^-3, an absolute step of 2^-24 below; include/dvsg_amd.h, conv_gemm.hip).  It is held to the SAME bounds as the exact float32 path in
tests/test_gpu_cnn.py / test_gpu_configs.py: stage activations <= 2e-5 relative, F_t <= 1e-5, warped
pixels < 1e-3 at 720p outside the counted border-discontinuity pixels -- and, against the exact path
itself, to the level at which two float32 GEMMs with different summation orders differ."""
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
    assert torch.cuda.is_available()
    from coupe.dvsg_amd.networks import LocNet
    return LocNet(synthetic_weights)


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 70, 100), (3, 33, 47), (1, 8, 8), (1, 20, 4)])
def test_every_stage_matches_oracle(net, synthetic_weights, B, H, W):
    x = inputs.window_frames(201, B, H, W)
    taps = {}
    F_ref = onet.localizationNet(x, 25, synthetic_weights, taps=taps)
    worst = 0.0
    for stage, name in enumerate(STAGES):
        act = net.tap(x, stage, precision="f32s").cpu().numpy()
        ref = taps[name]
        if name == "pool5":
            ref = ref.reshape(B, 1, 1, 2048)
        assert act.shape == ref.shape, name
        rel = np.abs(act - ref).max() / np.abs(ref).max()
        worst = max(worst, rel)
        assert rel <= 2e-5, "%s: relative error %.3g" % (name, rel)     # the float32 path's own bound
    F = net.forward(x, precision="f32s").cpu().numpy()
    assert np.abs(F - F_ref).max() <= 1e-5
    F32 = net.forward(x, precision="f32").cpu().numpy()
    print("f32s %dx%dx%d: worst stage %.2e, F_t vs oracle %.2e, vs the exact path %.2e"
          % (B, H, W, worst, np.abs(F - F_ref).max(), np.abs(F - F32).max()))
    assert np.abs(F - F32).max() <= 2e-6


def _to_pieces(t):
    """float32 tensor -> the mode's internal activation format (float16 pieces), as a byte tensor."""
    import torch
    from coupe.dvsg_amd import _lib
    out = torch.empty(t.numel() * 4, dtype=torch.uint8, device=t.device)
    _lib.call("dvsg_f32_to_pieces", t.data_ptr(), out.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream)
    return out


def _from_pieces(pcs, shape):
    import torch
    from coupe.dvsg_amd import _lib
    out = torch.empty(shape, dtype=torch.float32, device=pcs.device)
    _lib.call("dvsg_pieces_to_f32", pcs.data_ptr(), out.data_ptr(), out.numel(), torch.cuda.current_stream().cuda_stream)
    return out


def test_pieces_round_trip():
    """hi + lo carries 22 significant bits: |v - join(split(v))| <= 2^-22 |v| (+ float16's subnormal step for
    tiny values), zeros and float16-representable values are exact."""
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    v = (torch.rand((4, 9, 7, 64), generator=g, device=dev) - 0.5) * 300.0
    v[0, 0, 0, :8] = torch.tensor([0.0, 1.0, -2.5, 1e-3, 65504.0 / 2, 3.0e-6, -1e-7, 0.333], device=dev)
    back = _from_pieces(_to_pieces(v), v.shape)
    err = (back - v).abs()
    assert float((err - (v.abs() * 2.0 ** -21 + 6e-8)).max()) <= 0.0
    assert float(err[0, 0, 0, :3].max()) == 0.0


def test_pieces_conversions_reject_ragged_sizes_and_accept_none():
    import torch
    from coupe.dvsg_amd import _lib
    t = torch.zeros(64, device="cuda")
    out = torch.zeros(256, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("dvsg_f32_to_pieces", t.data_ptr(), out.data_ptr(), 0, st)       # nothing to do: fine
    _lib.call("dvsg_pieces_to_f32", out.data_ptr(), t.data_ptr(), 0, st)
    with pytest.raises(_lib.DvsgError):
        _lib.call("dvsg_f32_to_pieces", t.data_ptr(), out.data_ptr(), 40, st)  # not whole 32-element groups
    with pytest.raises(_lib.DvsgError):
        _lib.call("dvsg_pieces_to_f32", 0, t.data_ptr(), 64, st)


def test_layers_against_float64(net):
    """dvsg_conv_gemm_f32s on single layers (3x3 / 1x1, stride 2, residual; small batch = split-K, large =
    plain tiles and the stream-K tail) against float64 math on the same operands (the inputs as the mode
    holds them, 22 significant bits): the split products must be as close to it as the exact float32
    kernel is, up to the 22-bit rounding of the stored output."""
    import torch
    from coupe.dvsg_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    stream = torch.cuda.current_stream().cuda_stream
    scratch = torch.empty(66 << 20, dtype=torch.uint8, device=dev)
    for k, stride, cin, cout, h, w, Bs in [(3, 1, 64, 64, 20, 28, 2), (3, 2, 128, 128, 21, 17, 3), (1, 1, 256, 64, 9, 13, 2),
                                           (1, 1, 1024, 256, 23, 40, 16), (3, 1, 256, 256, 45, 80, 16), (1, 1, 128, 512, 30, 40, 16)]:
        K = k * k * cin
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        xp = _to_pieces(torch.rand((Bs, h, w, cin), generator=g, device=dev) * 4.0 - 1.0)
        rp = _to_pieces(torch.rand((Bs, ho, wo, cout), generator=g, device=dev) - 0.5)
        x, res = _from_pieces(xp, (Bs, h, w, cin)), _from_pieces(rp, (Bs, ho, wo, cout))   # what the mode holds
        wt = (torch.rand((cout, K), generator=g, device=dev) - 0.5) * (2.0 / K ** 0.5)
        bias = torch.rand((cout,), generator=g, device=dev) - 0.5
        hi = wt.half()
        lo = (wt - hi.float()).half()
        pieces = torch.cat([hi.reshape(cout, K // 32, 32), lo.reshape(cout, K // 32, 32)], 2).contiguous()
        y32 = torch.empty((Bs, ho, wo, cout), device=dev)
        yp = torch.empty(y32.numel() * 4, dtype=torch.uint8, device=dev)
        _lib.call("dvsg_conv_gemm_f32", x.data_ptr(), wt.data_ptr(), bias.data_ptr(), res.data_ptr(), y32.data_ptr(), Bs, h, w,
                  cin, cout, k, stride, 1, 1, scratch.data_ptr(), scratch.numel(), stream)
        _lib.call("dvsg_conv_gemm_f32s", xp.data_ptr(), pieces.data_ptr(), bias.data_ptr(), rp.data_ptr(), yp.data_ptr(), Bs, h,
                  w, cin, cout, k, stride, 1, 1, scratch.data_ptr(), scratch.numel(), stream)
        ys = _from_pieces(yp, y32.shape)
        w4 = wt.double().reshape(cout, k, k, cin).permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w4, bias.double(), stride=stride, padding=k // 2)
        ref = torch.relu(ref.permute(0, 2, 3, 1) + res.double())
        e32 = float((y32.double() - ref).abs().max())
        es = float((ys.double() - ref).abs().max())
        scale = float(ref.abs().max())
        assert es <= max(4.0 * e32, 2e-6 * scale), (k, stride, cin, cout, Bs, es, e32, scale)


def test_stabilize_720p_end_to_end(synthetic_weights):
    """The float32 path's end-to-end bounds at 1280x720 (tests/test_gpu_configs.py::test_cfg1_*), two windows."""
    from coupe.dvsg_amd.model import Session, StabNet
    from oracle.cnn_torch import TorchLocNet
    B, H, W = 2, 720, 1280
    x = inputs.window_frames(7, B, H, W)
    u = np.ascontiguousarray(x[..., 18:])
    model = StabNet(H, W).load_weights(synthetic_weights)
    model.precision = "f32s"
    ins, outs = model.get_evaluation_model(7)
    pred, F, xs, ys = Session().run([outs["s_t_pred"], outs["F_t"], outs["x_offset_t"], outs["y_offset_t"]],
                                    {ins["patches_t"]: x, ins["u_t"]: u})
    F_ref = TorchLocNet(synthetic_weights).forward(x)
    r_pred, r_xs, r_ys = otps.ThinPlateSpline(u, inputs.v_src(B), F_ref, (H, W))
    border = otps.border_discontinuity_mask(r_xs, r_ys, H, W, delta=3e-2).reshape(B, H, W)
    ferr = np.abs(F - F_ref).max()
    gerr = max(np.abs(xs - r_xs).max() * W / 2, np.abs(ys - r_ys).max() * H / 2)
    perr = np.abs(pred - r_pred).max(axis=3)[~border].max()
    print("f32s at 720p: F_t %.2e, grid %.2e px, pixels %.2e" % (ferr, gerr, perr))
    assert ferr <= 1e-5 and gerr < 2e-2 and perr < 1e-3 and border.mean() < 0.01


def test_clip_loop_and_determinism(synthetic_weights):
    """eval.py's 40-frame clip in this mode against the committed golden (the float32 test's bounds), twice:
    bitwise the same both times."""
    import os
    from coupe.dvsg_amd.clip import stabilize_clip
    from coupe.dvsg_amd.model import Session, StabNet
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip.npz")
    with np.load(gold, allow_pickle=False) as z:
        ref, bits = z["stabilised"], z["border_mask_bits"]
    N, H, W = 40, 32, 48
    frames = inputs.smooth_frames(3001, N, H, W)
    model = StabNet(H, W).load_weights(synthetic_weights)
    model.precision = "f32s"
    model.get_evaluation_model(7)
    out = stabilize_clip(model, Session(), frames)
    err = np.abs(out - ref).reshape(N, -1)
    assert np.median(err, axis=1).max() < 3e-6 and (err > 1e-3).mean(axis=1).max() <= 1e-3
    mask = np.unpackbits(bits)[:N * H * W].astype(bool).reshape(N, H, W)
    assert np.abs(out - ref).max(axis=3)[~mask].max() < 1e-4     # nothing flips away from sampler A's border jumps
    assert np.array_equal(out, stabilize_clip(model, Session(), frames))


def test_small_operands_lose_bits_as_documented():
    """The second piece is stored UNSCALED (lo = f16(v - hi)), so for |v| < 2^-3 it is a float16 subnormal with an
    absolute step of 2^-24: the pair carries ~20 bits at |v| = 0.03, ~15 at 1e-3 (include/dvsg_amd.h).  The parity
    tests above run on weights of ~0.03 .. 0.1; this one states what the mode gives with BatchNorm-folded weights of
    magnitude ~1e-3 against O(1) activations, measured against float64 math on the TRUE float32 operands (CPU
    emulation of the same arithmetic: 3.0e-5 of the output scale; weights of 0.06: 5e-7; plain float16 operands:
    3e-4; exact float32: 1e-7).  Better than float16 by 10x, short of float32 by 100x: a checkpoint with such
    magnitudes is unpinned in this mode."""
    import torch
    from coupe.dvsg_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(9)
    stream = torch.cuda.current_stream().cuda_stream
    scratch = torch.empty(66 << 20, dtype=torch.uint8, device=dev)
    Bs, h, w, cin, cout, k = 2, 20, 28, 256, 128, 1
    x = (torch.rand((Bs, h, w, cin), generator=g, device=dev) - 0.25) * 2.0
    wt = (torch.rand((cout, cin), generator=g, device=dev) - 0.5) * 2e-3
    bias = torch.zeros((cout,), device=dev)
    xp = _to_pieces(x)
    hi = wt.half()
    lo = (wt - hi.float()).half()
    pieces = torch.cat([hi.reshape(cout, cin // 32, 32), lo.reshape(cout, cin // 32, 32)], 2).contiguous()
    yp = torch.empty(Bs * h * w * cout * 4, dtype=torch.uint8, device=dev)
    y32 = torch.empty((Bs, h, w, cout), device=dev)
    _lib.call("dvsg_conv_gemm_f32s", xp.data_ptr(), pieces.data_ptr(), bias.data_ptr(), 0, yp.data_ptr(), Bs, h, w, cin, cout,
              k, 1, 0, 1, scratch.data_ptr(), scratch.numel(), stream)
    _lib.call("dvsg_conv_gemm_f32", x.data_ptr(), wt.data_ptr(), bias.data_ptr(), 0, y32.data_ptr(), Bs, h, w, cin, cout,
              k, 1, 0, 1, scratch.data_ptr(), scratch.numel(), stream)
    ref = x.double().reshape(-1, cin) @ wt.double().t()
    ys = _from_pieces(yp, y32.shape).double().reshape(-1, cout)
    scale = float(ref.abs().max())
    es, e32 = float((ys - ref).abs().max()) / scale, float((y32.double().reshape(-1, cout) - ref).abs().max()) / scale
    e16 = float(((x.half().double().reshape(-1, cin) @ wt.half().double().t()) - ref).abs().max()) / scale
    print("weights ~1e-3: relative error f32s %.2e, exact float32 %.2e, plain float16 operands %.2e" % (es, e32, e16))
    assert e32 < 1e-6
    assert es < 1e-4 and es < e16     # bounded and stated: NOT float32-level here
