"""GPU: the "f32x3" precision -- float32 tensors and accumulation exactly as the path of record, every product of the 52
1x1 / 3x3 convolutions formed on the bfloat16 matrix cores from THREE bfloat16 pieces per operand (all 24 significant bits
of both float32 operands at any magnitude; six of the nine cross terms: include/dvsg_amd.h, conv_gemm_tile.h X3).  Held to
the exact float32 path's own bounds against the oracle (stages <= 2e-5 relative, F_t <= 1e-5, 720p pixels < 1e-3) and, against
FLOAT64 evaluations of the same definitions, to the exact path's own error: the claim of the mode is that it is as close
to the real-number result as the float32 matrix instructions are."""
import numpy as np
import pytest

import inputs
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
        act = net.tap(x, stage, precision="f32x3").cpu().numpy()
        ref = taps[name]
        if name == "pool5":
            ref = ref.reshape(B, 1, 1, 2048)
        assert act.shape == ref.shape, name
        rel = np.abs(act - ref).max() / np.abs(ref).max()
        worst = max(worst, rel)
        assert rel <= 2e-5, "%s: relative error %.3g" % (name, rel)     # the float32 path's own bound
    F = net.forward(x, precision="f32x3").cpu().numpy()
    assert np.abs(F - F_ref).max() <= 1e-5
    F32 = net.forward(x, precision="f32").cpu().numpy()
    print("f32x3 %dx%dx%d: worst stage %.2e, F_t vs oracle %.2e, vs the exact path %.2e"
          % (B, H, W, worst, np.abs(F - F_ref).max(), np.abs(F - F32).max()))
    assert np.abs(F - F32).max() <= 1e-6


def _bf16_planes_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def test_packed_pieces_sum_to_the_weight_exactly():
    """dvsg_pack_weights_f32x3: p1 + p2 + p3 == w bit for bit for every weight -- O(1), 1e-3, 1e-20, 1e20, exact powers of two
    -- and every piece sits where the kernel reads it (group of 64 rows, 32-k stage, plane, row, swizzled 16-byte chunk)."""
    import torch
    from coupe.dvsg_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(11)
    rows, K = 128, 96
    wt = (torch.rand((rows, K), generator=g, device=dev) - 0.5) * 2.0
    wt[0, :8] = torch.tensor([0.0, 1.0, -2.5, 1e-3, 3.0e-20, -7.7e19, 2.0 ** -40, 0.333], device=dev)
    wt[1] *= 1e-3
    wt[2] *= 1e-20
    wt[3] *= 1e20
    packed = torch.empty((rows * K * 3,), dtype=torch.int16, device=dev)
    _lib.call("dvsg_pack_weights_f32x3", wt.data_ptr(), packed.data_ptr(), rows, K, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    pk = packed.cpu().numpy().view(np.uint16).reshape(rows // 64, K // 32, 3, 64, 4, 8)
    w = wt.cpu().numpy()
    back = np.zeros((3, rows, K), dtype=np.float32)
    for n in range(rows):
        R = n % 64
        for c in range(4):
            pos = c ^ ((R >> 2) & 3)
            for st in range(K // 32):
                back[:, n, st * 32 + c * 8: st * 32 + c * 8 + 8] = _bf16_planes_to_f32(pk[n // 64, st, :, R, pos, :])
    total = (back[0].astype(np.float64) + back[1] + back[2]).astype(np.float32)
    assert np.array_equal(total, w)
    assert np.abs(back[1]).max() <= np.abs(w).max() * 2.0 ** -8 and np.abs(back[2]).max() <= np.abs(w).max() * 2.0 ** -16
    with pytest.raises(_lib.DvsgError):
        _lib.call("dvsg_pack_weights_f32x3", wt.data_ptr(), packed.data_ptr(), 100, K, 0)     # rows % 64
    with pytest.raises(_lib.DvsgError):
        _lib.call("dvsg_pack_weights_f32x3", wt.data_ptr(), packed.data_ptr(), rows, 40, 0)    # K % 32


LAYERS = [  # k, stride, cin, cout, h, w, B, activation scale, weight scale
    (3, 1, 64, 64, 20, 28, 2, 1.0, 1.0), (3, 2, 128, 128, 21, 17, 3, 1.0, 1.0), (1, 1, 256, 64, 9, 13, 2, 1.0, 1.0),
    (1, 1, 1024, 256, 23, 40, 16, 1.0, 1.0), (3, 1, 256, 256, 45, 80, 16, 1.0, 1.0), (1, 1, 128, 512, 30, 40, 16, 1.0, 1.0),
    (3, 1, 512, 512, 23, 40, 16, 1.0, 1.0),            # 460 wide tiles: one stream-K round
    (3, 1, 512, 512, 23, 40, 1, 1.0, 1.0), (1, 1, 512, 2048, 23, 40, 1, 1.0, 1.0),   # batch 1: split-K
    (1, 1, 256, 128, 20, 28, 2, 1.0, 1e-3),            # BatchNorm-folded weights of ~1e-3: where two float16 pieces lose bits
    (3, 1, 128, 128, 20, 28, 2, 1e-4, 1e-2), (3, 1, 128, 128, 20, 28, 2, 1e4, 1e3), (1, 1, 64, 64, 1, 1, 1, 1.0, 1.0)]


@pytest.mark.parametrize("k,stride,cin,cout,h,w,B,ascale,wscale", LAYERS)
def test_layers_against_float64(k, stride, cin, cout, h, w, B, ascale, wscale):
    """dvsg_conv_gemm_f32x3 on single layers (plain tiles, split-K at batch 1, the stream-K round, stride 2, residual, ragged
    M) against float64 math on the same float32 operands: at most the exact float32 kernel's own error (x 1.25 + one unit of
    float32 rounding of the output scale), whatever the operands' magnitudes; the same bits on a second launch."""
    import torch
    from coupe.dvsg_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    stream = torch.cuda.current_stream().cuda_stream
    scratch = torch.empty(66 << 20, dtype=torch.uint8, device=dev)
    K = k * k * cin
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    x = (torch.rand((B, h, w, cin), generator=g, device=dev) * 4.0 - 1.0) * ascale
    wt = (torch.rand((cout, K), generator=g, device=dev) - 0.5) * (2.0 / K ** 0.5) * wscale
    bias = (torch.rand((cout,), generator=g, device=dev) - 0.5) * ascale * wscale
    res = (torch.rand((B, ho, wo, cout), generator=g, device=dev) - 0.5) * ascale * wscale
    packed = torch.empty((cout * K * 6,), dtype=torch.uint8, device=dev)
    _lib.call("dvsg_pack_weights_f32x3", wt.data_ptr(), packed.data_ptr(), cout, K, stream)
    y32 = torch.empty((B, ho, wo, cout), device=dev)
    ys = [torch.empty_like(y32), torch.empty_like(y32)]
    _lib.call("dvsg_conv_gemm_f32", x.data_ptr(), wt.data_ptr(), bias.data_ptr(), res.data_ptr(), y32.data_ptr(), B, h, w,
              cin, cout, k, stride, 1, 1, scratch.data_ptr(), scratch.numel(), stream)
    for y in ys:
        _lib.call("dvsg_conv_gemm_f32x3", x.data_ptr(), packed.data_ptr(), bias.data_ptr(), res.data_ptr(), y.data_ptr(), B, h,
                  w, cin, cout, k, stride, 1, 1, scratch.data_ptr(), scratch.numel(), stream)
    torch.cuda.synchronize()
    assert torch.equal(ys[0], ys[1])
    w4 = wt.double().reshape(cout, k, k, cin).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w4, bias.double(), stride=stride, padding=k // 2)
    ref = torch.relu(ref.permute(0, 2, 3, 1) + res.double())
    scale = float(ref.abs().max())
    e32 = float((y32.double() - ref).abs().max())
    e3 = float((ys[0].double() - ref).abs().max())
    r32 = float(((y32.double() - ref) ** 2).mean().sqrt())
    r3 = float(((ys[0].double() - ref) ** 2).mean().sqrt())
    print("k%d s%d %d->%d B=%d: max err f32 %.2e f32x3 %.2e, rms f32 %.2e f32x3 %.2e (of max |y| %.3g)"
          % (k, stride, cin, cout, B, e32 / scale, e3 / scale, r32 / scale, r3 / scale, scale))
    assert e3 <= 1.25 * e32 + 6e-8 * scale and r3 <= 1.1 * r32 + 1e-9 * scale


@pytest.mark.parametrize("B,h,w,cin,cout,stride,res_stride", [(2, 45, 80, 64, 256, 1, 1), (1, 37, 53, 64, 256, 2, 2), (3, 20, 31, 128, 128, 1, 1),
                                                             (1, 90, 160, 64, 256, 2, 1), (16, 180, 320, 64, 256, 1, 1), (1, 1, 1, 64, 64, 1, 1)])
def test_fused_conv3x3_conv1x1_against_float64(B, h, w, cin, cout, stride, res_stride):
    """Block 1's conv2 + conv3 in the precision's own fused kernel (conv_fused_x3.hip) against float64 math on the same
    float32 operands, next to the exact fused kernel's error: ragged M tiles, stride 2 with the subsampled-shortcut residual,
    other channel counts, a one-pixel frame, the full-size launch; the same bits on a second launch."""
    import torch
    from coupe.dvsg_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(17)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.rand((B, h, w, cin), generator=g, device=dev) - 0.3
    w2 = (torch.rand((64, 9 * cin), generator=g, device=dev) - 0.5) * (2.0 / (9 * cin) ** 0.5)
    b2 = torch.rand((64,), generator=g, device=dev) - 0.5
    w3 = (torch.rand((cout, 64), generator=g, device=dev) - 0.5) * 0.25
    b3 = torch.rand((cout,), generator=g, device=dev) - 0.5
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    res = torch.rand((B, (ho - 1) * res_stride + 1, (wo - 1) * res_stride + 1, cout), generator=g, device=dev) - 0.5
    p2 = torch.empty((64 * 9 * cin * 6,), dtype=torch.uint8, device=dev)
    p3 = torch.empty((cout * 64 * 6,), dtype=torch.uint8, device=dev)
    _lib.call("dvsg_pack_weights_f32x3", w2.data_ptr(), p2.data_ptr(), 64, 9 * cin, st)
    _lib.call("dvsg_pack_weights_f32x3", w3.data_ptr(), p3.data_ptr(), cout, 64, st)
    ys = [torch.full((B, ho, wo, cout), float("nan"), device=dev) for _ in range(3)]
    for y in ys[:2]:
        _lib.call("dvsg_conv3x3_1x1_f32x3", x.data_ptr(), p2.data_ptr(), b2.data_ptr(), p3.data_ptr(), b3.data_ptr(), res.data_ptr(),
                  y.data_ptr(), B, h, w, cin, cout, stride, res_stride, st)
    exact_ok = cout % 128 == 0      # the exact fused kernel wants whole 128-channel halves
    if exact_ok:
        _lib.call("dvsg_conv3x3_1x1_f32", x.data_ptr(), w2.data_ptr(), b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), res.data_ptr(),
                  ys[2].data_ptr(), B, h, w, cin, cout, stride, res_stride, st)
    torch.cuda.synchronize()
    assert torch.equal(ys[0], ys[1]) and bool(torch.isfinite(ys[0]).all())
    w2c = w2.double().reshape(64, 3, 3, cin).permute(0, 3, 1, 2)
    mid = torch.relu(torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w2c, b2.double(), stride=stride, padding=1))
    ref = torch.nn.functional.conv2d(mid, w3.double().reshape(cout, 64, 1, 1), b3.double()).permute(0, 2, 3, 1)
    ref = torch.relu(ref + res.double()[:, ::res_stride, ::res_stride])
    scale = max(float(ref.abs().max()), 1.0)
    e3 = float((ys[0].double() - ref).abs().max())
    assert e3 <= 2e-6 * scale, e3 / scale
    if exact_ok:
        e32 = float((ys[2].double() - ref).abs().max())
        print("fused %d->64->%d s%d B=%d: max err f32 %.2e f32x3 %.2e (of %.3g)" % (cin, cout, stride, B, e32 / scale, e3 / scale, scale))
        assert e3 <= 1.25 * e32 + 6e-8 * scale
    with pytest.raises(_lib.DvsgError):     # Cout % 64
        _lib.call("dvsg_conv3x3_1x1_f32x3", x.data_ptr(), p2.data_ptr(), b2.data_ptr(), p3.data_ptr(), b3.data_ptr(), res.data_ptr(),
                  ys[0].data_ptr(), B, h, w, cin, 96, stride, res_stride, st)


def test_f_t_against_the_float64_arbiter(net, synthetic_weights):
    """The whole CNN at 288x512 against the float64 evaluation of the same graph on the same float32 weights and frames:
    the mode's F_t and pooled features are at most as far from it as the exact float32 path's (x 1.25)."""
    import torch
    from oracle.cnn_torch import TorchLocNet
    x = inputs.window_frames(4242, 2, 288, 512)
    ref64 = TorchLocNet(synthetic_weights, dtype=torch.float64)
    rF, rpool = ref64.forward(x), ref64.features(x).numpy()
    xt = torch.from_numpy(x).cuda()
    errs = {}
    for precision in ("f32", "f32x3"):
        F = net.forward(xt, precision=precision).cpu().numpy()
        pool = net.tap(xt, 18, precision=precision).cpu().numpy().reshape(rpool.shape)
        e = (pool - rpool) / np.abs(rpool).max()
        errs[precision] = (np.abs(F - rF).max(), np.sqrt((e ** 2).mean()), abs(e.mean()), np.abs(e).max())
        assert np.array_equal(F, net.forward(xt, precision=precision).cpu().numpy())
    print("288x512 vs float64: F_t f32 %.3g f32x3 %.3g; pool5 (relative) rms f32 %.3g f32x3 %.3g, |mean| f32 %.3g f32x3 %.3g"
          % (errs["f32"][0], errs["f32x3"][0], errs["f32"][1], errs["f32x3"][1], errs["f32"][2], errs["f32x3"][2]))
    # the 4096 pooled features give stable statistics (rms, and the MEAN: a bias is what an average pool keeps -- one shared
    # accumulator for all six cross terms showed up here as 4 x the exact path's mean error); F_t's maximum over 100 values of
    # ~2e-8 moves by +-40 % between roundings
    assert errs["f32x3"][1] <= 1.1 * errs["f32"][1] + 2e-9 and errs["f32x3"][2] <= 1.25 * errs["f32"][2] + 3e-9
    assert errs["f32x3"][3] <= 1.5 * errs["f32"][3] + 1e-8 and errs["f32x3"][0] <= 2.0 * errs["f32"][0] + 5e-9


def test_stabilize_720p_end_to_end(synthetic_weights):
    """The float32 path's end-to-end bounds at 1280x720 (tests/test_gpu_configs.py::test_cfg1_*), two windows."""
    from coupe.dvsg_amd.model import Session, StabNet
    from oracle.cnn_torch import TorchLocNet
    B, H, W = 2, 720, 1280
    x = inputs.window_frames(7, B, H, W)
    u = np.ascontiguousarray(x[..., 18:])
    model = StabNet(H, W).load_weights(synthetic_weights)
    model.precision = "f32x3"
    ins, outs = model.get_evaluation_model(7)
    pred, F, xs, ys = Session().run([outs["s_t_pred"], outs["F_t"], outs["x_offset_t"], outs["y_offset_t"]],
                                    {ins["patches_t"]: x, ins["u_t"]: u})
    F_ref = TorchLocNet(synthetic_weights).forward(x)
    r_pred, r_xs, r_ys = otps.ThinPlateSpline(u, inputs.v_src(B), F_ref, (H, W))
    border = otps.border_discontinuity_mask(r_xs, r_ys, H, W, delta=3e-2).reshape(B, H, W)
    ferr = np.abs(F - F_ref).max()
    gerr = max(np.abs(xs - r_xs).max() * W / 2, np.abs(ys - r_ys).max() * H / 2)
    perr = np.abs(pred - r_pred).max(axis=3)[~border].max()
    print("f32x3 at 720p: F_t %.2e, grid %.2e px, pixels %.2e" % (ferr, gerr, perr))
    assert ferr <= 1e-5 and gerr < 2e-2 and perr < 1e-3 and border.mean() < 0.01


def test_clip_loop_ring_and_determinism(synthetic_weights):
    """eval.py's 40-frame clip in this mode (through the frame ring) against the committed golden, the float32 test's
    bounds; twice: bitwise the same both times."""
    import os
    from coupe.dvsg_amd.clip import stabilize_clip
    from coupe.dvsg_amd.model import Session, StabNet
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip.npz")
    with np.load(gold, allow_pickle=False) as z:
        ref, bits = z["stabilised"], z["border_mask_bits"]
    N, H, W = 40, 32, 48
    frames = inputs.smooth_frames(3001, N, H, W)
    model = StabNet(H, W).load_weights(synthetic_weights)
    model.precision = "f32x3"
    model.get_evaluation_model(7)
    out = stabilize_clip(model, Session(), frames)
    err = np.abs(out - ref).reshape(N, -1)
    assert np.median(err, axis=1).max() < 3e-6 and (err > 1e-3).mean(axis=1).max() <= 1e-3
    mask = np.unpackbits(bits)[:N * H * W].astype(bool).reshape(N, H, W)
    assert np.abs(out - ref).max(axis=3)[~mask].max() < 1e-4     # nothing flips away from sampler A's border jumps
    assert np.array_equal(out, stabilize_clip(model, Session(), frames))
