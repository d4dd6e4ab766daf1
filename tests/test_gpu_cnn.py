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


@pytest.mark.parametrize("B,H,W", [(1, 8, 8), (2, 6, 6), (1, 20, 4), (3, 1, 1), (1, 4, 37)])
def test_tiny_frames_match_oracle(net, synthetic_weights, B, H, W):
    """Frames so small that h*w*channels GROWS in the later blocks (the ceil in (h-1)/stride+1: an
    8x8 frame is 1x1x2048 in block 4 but only 2x2x256 in block 1).  The workspace plan sizes every
    buffer by its largest tenant; before that fix r1 spilled into r2 and a unit's output into its
    own input here, silently."""
    x = inputs.window_frames(241, B, H, W)
    taps = {}
    F_ref = onet.localizationNet(x, 25, synthetic_weights, taps=taps)
    for stage, name in ((4, "block1/unit_3"), (8, "block2/unit_4"), (14, "block3/unit_6"), (17, "block4/unit_3")):
        act = net.tap(x, stage).cpu().numpy()
        ref = taps[name]
        assert act.shape == ref.shape, name
        assert np.abs(act - ref).max() <= 2e-5 * np.abs(ref).max(), name
    assert np.abs(net.forward(x).cpu().numpy() - F_ref).max() <= 1e-5


def test_window_width_must_match_the_checkpoint(net, synthetic_weights):
    """The C entry points take no channel count (conv1 reads 21 floats per pixel): a narrower window
    must be rejected on the host, not read past its end."""
    import torch
    from coupe.dvsg_amd.clip import stabilize_clip
    from coupe.dvsg_amd.model import Session, StabNet
    H, W = 32, 48
    x = inputs.window_frames(251, 1, H, W)
    model = StabNet(H, W).load_weights(synthetic_weights)
    ins, outs = model.get_evaluation_model(5)
    with pytest.raises(ValueError, match="conv1"):
        Session().run(outs["s_t_pred"], {ins["patches_t"]: x[..., :15], ins["u_t"]: x[..., 12:15]})
    with pytest.raises(ValueError, match="conv1"):
        stabilize_clip(model, Session(), inputs.smooth_frames(252, 2, H, W), skip_length=(0, 16, 24, 28, 32))
    dev = torch.device("cuda:0")
    p15 = torch.zeros((1, H, W, 15), device=dev)
    u = torch.zeros((1, H, W, 3), device=dev)
    with pytest.raises(ValueError, match="patches_t"):
        net.stabilize(p15, u, torch.empty_like(u), torch.empty((1, 25, 2), device=dev))
    with pytest.raises(ValueError, match="contiguous float32"):
        net.stabilize(torch.zeros((1, H, W, 21), device=dev, dtype=torch.float64), u, torch.empty_like(u),
                      torch.empty((1, 25, 2), device=dev))


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


def test_split_k_reduction_is_exact_and_deterministic():
    """Small-batch launches split K over several workgroups per tile, large launches cut their
    partly filled last round of tiles into equal (tile, K-stage) shares (stream-K tail); both
    reduce last-arriver in K order: same result as the plain kernel up to float32 re-association,
    bitwise equal from run to run, and correct when many tiles reduce concurrently."""
    import torch
    from coupe.dvsg_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    s = torch.cuda.current_stream().cuda_stream
    scratch = torch.empty(66 << 20, dtype=torch.uint8, device=dev)
    for k, stride, cin, cout, h, w, B in [(3, 1, 256, 256, 23, 40, 1), (1, 1, 2048, 512, 23, 40, 1),
                                          (3, 2, 256, 256, 45, 80, 1), (3, 1, 512, 512, 6, 10, 2),
                                          # stream-K tails: 573 tiles (ragged last one) and 900 tiles
                                          (3, 1, 128, 128, 91, 161, 5), (1, 1, 512, 256, 90, 160, 4),
                                          # whole launch as one stream-K round: 450 and 460 wide tiles
                                          (3, 1, 256, 256, 45, 80, 8), (1, 1, 1024, 512, 23, 40, 16),
                                          # 64-wide tiles, several rounds: plain path
                                          (3, 1, 64, 64, 90, 160, 6),
                                          # 16 wide / 20 narrow tiles left over after full rounds
                                          (1, 1, 256, 1024, 90, 160, 4), (3, 1, 64, 64, 266, 256, 1)]:
        x = torch.rand((B, h, w, cin), generator=g, device=dev) - 0.3
        K = k * k * cin
        wt = (torch.rand((cout, K), generator=g, device=dev) - 0.5) * (2.0 / K ** 0.5)
        bias = torch.rand((cout,), generator=g, device=dev) - 0.5
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        res = torch.rand((B, ho, wo, cout), generator=g, device=dev) - 0.5
        outs = []
        for sc, nb in ((0, 0), (scratch.data_ptr(), scratch.numel()), (scratch.data_ptr(), scratch.numel())):
            y = torch.empty((B, ho, wo, cout), device=dev)
            _lib.call("dvsg_conv_gemm_f32", x.data_ptr(), wt.data_ptr(), bias.data_ptr(), res.data_ptr(),
                      y.data_ptr(), B, h, w, cin, cout, k, stride, 1, 1, sc, nb, s)
            outs.append(y)
        torch.cuda.synchronize()
        assert torch.equal(outs[1], outs[2])
        scale = float(outs[0].abs().max())
        assert float((outs[0] - outs[1]).abs().max()) <= 2e-6 * max(scale, 1.0)
        w4 = wt.reshape(cout, k, k, cin).permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w4, bias, stride=stride, padding=k // 2)
        ref = torch.relu(ref.permute(0, 2, 3, 1) + res)
        assert float((outs[1] - ref).abs().max()) <= 2e-5 * max(scale, 1.0)


@pytest.mark.parametrize("B,h,w,cin,cout,stride,res_stride", [(2, 45, 80, 64, 256, 1, 1), (1, 37, 53, 64, 256, 2, 2),
                                                             (3, 20, 31, 128, 128, 1, 1), (1, 90, 160, 64, 256, 2, 1)])
def test_fused_conv3x3_conv1x1(B, h, w, cin, cout, stride, res_stride):
    """Block 1's conv2 + conv3 in one kernel against two torch convolutions (float32): ragged M tiles,
    stride 2 with the subsampled-shortcut residual, other channel counts."""
    import torch
    from coupe.dvsg_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(17)
    x = torch.rand((B, h, w, cin), generator=g, device=dev) - 0.3
    w2 = (torch.rand((64, 9 * cin), generator=g, device=dev) - 0.5) * (2.0 / (9 * cin) ** 0.5)
    b2 = torch.rand((64,), generator=g, device=dev) - 0.5
    w3 = (torch.rand((cout, 64), generator=g, device=dev) - 0.5) * 0.25
    b3 = torch.rand((cout,), generator=g, device=dev) - 0.5
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    res = torch.rand((B, (ho - 1) * res_stride + 1, (wo - 1) * res_stride + 1, cout), generator=g, device=dev) - 0.5
    y = torch.full((B, ho, wo, cout), float("nan"), device=dev)
    _lib.call("dvsg_conv3x3_1x1_f32", x.data_ptr(), w2.data_ptr(), b2.data_ptr(), w3.data_ptr(), b3.data_ptr(),
              res.data_ptr(), y.data_ptr(), B, h, w, cin, cout, stride, res_stride, torch.cuda.current_stream().cuda_stream)
    w2c = w2.reshape(64, 3, 3, cin).permute(0, 3, 1, 2)
    mid = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w2c, b2, stride=stride, padding=1))
    ref = torch.nn.functional.conv2d(mid, w3.reshape(cout, 64, 1, 1), b3).permute(0, 2, 3, 1)
    ref = torch.relu(ref + res[:, ::res_stride, ::res_stride])
    assert bool(torch.isfinite(y).all())
    assert float((y - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1.0)


@pytest.mark.parametrize("B,H,W", [(1, 17, 23), (1, 40, 300), (33, 32, 32)])
def test_ragged_and_extreme_shapes(net, synthetic_weights, B, H, W):
    """Tiny frames (every late feature map is 1x1), a width whose conv1 row needs a full and a
    partial 128-pixel tile, and a batch that needs three head chunks."""
    x = inputs.window_frames(241, B, H, W)
    F = net.forward(x).cpu().numpy()
    F_ref = onet.localizationNet(x, 25, synthetic_weights)
    assert np.abs(F - F_ref).max() <= 1e-5
    F16 = net.forward(x, precision="f16").cpu().numpy()
    assert np.abs(F16 - F_ref).max() < 3e-3


@pytest.mark.parametrize("precision", ["f32", "f32s", "f32x3"])
def test_a_step_replays_from_a_captured_hip_graph(net, precision):
    """The library allocates and synchronises nothing inside a call: a dvsg_stabilize_* step can be
    captured into a HIP graph as it stands and replays bit for bit, split-K tickets included."""
    import torch
    B, H, W = 2, 96, 160
    dev = torch.device("cuda")
    x = torch.from_numpy(inputs.window_frames(261, B, H, W)).to(dev)
    u = x[..., 18:].contiguous()
    out = torch.empty((B, H, W, 3), device=dev)
    F = torch.empty((B, 25, 2), device=dev)
    net.stabilize(x, u, out, F, precision=precision)
    torch.cuda.synchronize()
    ref_out, ref_F = out.clone(), F.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        net.stabilize(x, u, out, F, precision=precision)   # workspace sized before the capture
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        net.stabilize(x, u, out, F, precision=precision)
    for _ in range(3):
        out.zero_()
        F.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(F, ref_F), float((F - ref_F).abs().max())
        assert torch.equal(out, ref_out), float((out - ref_out).abs().max())


@pytest.mark.parametrize("precision,tol", [("f32", 1e-5), ("f32s", 1e-5), ("f16", 5e-5)])
def test_window_tensor_off_the_16_byte_grid(net, synthetic_weights, precision, tol):
    """conv1 stages its input rows as aligned 16-byte groups when it can; a window tensor that starts 4 bytes off
    (a view into a larger buffer) takes the element-wise path and must give the same answer."""
    import torch
    B, H, W = 2, 40, 64
    x = inputs.window_frames(251, B, H, W)
    F_ref = onet.localizationNet(x, 25, synthetic_weights)
    flat = torch.zeros(x.size + 8, dtype=torch.float32, device="cuda")
    outs = []
    for off in (0, 1, 3):
        view = flat[off:off + x.size].view(B, H, W, x.shape[3])
        view.copy_(torch.from_numpy(x))
        assert view.data_ptr() % 16 == (4 * off) % 16
        outs.append(net.forward(view, precision=precision).cpu().numpy())
        assert np.abs(outs[-1] - F_ref).max() <= tol, (precision, off)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])   # same products, same order


@pytest.mark.parametrize("B,H,W", [(3, 200, 320), (2, 288, 512), (5, 150, 270), (7, 96, 416), (4, 360, 640)])
def test_tile_count_regimes(net, synthetic_weights, B, H, W):
    """Shapes whose conv launches fall into the different work decompositions (split-K for a
    handful of tiles, one fat round, whole-launch stream-K, full rounds + stream-K tail, ragged last
    M tile): F_t against the independent torch-CPU oracle."""
    from oracle.cnn_torch import TorchLocNet
    x = inputs.window_frames(977 + B, B, H, W)
    F = net.forward(x).cpu().numpy()
    F_ref = TorchLocNet(synthetic_weights).forward(x)
    assert np.abs(F - F_ref).max() <= 1e-5
    # the other precisions pick their own decompositions at these sizes (f32s: fat tiles; float16: 256 x 128 tiles for the
    # launches of >= 256 of them, the fused block-1 kernel): same oracle, their own bounds
    assert np.abs(net.forward(x, precision="f32s").cpu().numpy() - F_ref).max() <= 1e-5
    assert np.abs(net.forward(x, precision="f16").cpu().numpy() - F_ref).max() <= 5e-5


@pytest.mark.parametrize("precision", ["f32", "f32s"])
@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 33, 47), (1, 8, 8), (1, 1, 1), (3, 130, 200), (1, 288, 512)])
def test_shortcut_and_conv1_as_one_launch(net, precision, B, H, W):
    """Blocks 2-4's opening units run `shortcut` (no ReLU) and `conv1` (ReLU) -- same input -- as ONE conv GEMM over their
    concatenated weight rows: the output [M, depth + base] sits in one buffer, conv2 reads its last `base` columns and conv3
    its first `depth` as residual through row strides ("concat_sc"; dvsg_debug_set_option 0 = two launches).  Same products in
    the same K order; the tile decomposition (and with it where split-K / stream-K cut a K loop) may differ: every unit's
    output and F_t agree to float32 re-association."""
    import torch
    from coupe.dvsg_amd import _lib
    x = torch.from_numpy(inputs.window_frames(17 * H + W, B, H, W)).cuda()
    got = {}
    try:
        for v in (0, 1):
            _lib.call("dvsg_debug_set_option", b"concat_sc", v)
            got[v] = [net.tap(x, st, precision=precision).clone() for st in (5, 9, 15)] + [net.forward(x, precision=precision).clone()]
    finally:
        _lib.call("dvsg_debug_set_option", b"concat_sc", 1)
    for a, b in zip(got[0], got[1]):
        scale = max(1.0, float(a.abs().max()))
        assert float((a - b).abs().max()) <= 2e-6 * scale, float((a - b).abs().max()) / scale


def test_bad_calls_are_rejected(net):
    import torch
    from coupe.dvsg_amd import DvsgError, _lib
    x = torch.zeros((1, 32, 32, 21), device="cuda")
    F = torch.zeros((1, 25, 2), device="cuda")
    ws, nbytes = net.workspace(1, 32, 32)
    with pytest.raises(DvsgError, match="workspace"):
        _lib.call("dvsg_locnet_forward_f32", net.handle, x.data_ptr(), 1, 32, 32, F.data_ptr(), ws.data_ptr(), 1024, 0)
    with pytest.raises(DvsgError, match="aligned"):
        _lib.call("dvsg_locnet_forward_f32", net.handle, x.data_ptr(), 1, 32, 32, F.data_ptr(), ws.data_ptr() + 4,
                  nbytes, 0)
    with pytest.raises(DvsgError, match="bad shape"):
        _lib.call("dvsg_locnet_forward_f32", net.handle, x.data_ptr(), 0, 32, 32, F.data_ptr(), ws.data_ptr(), nbytes, 0)
    with pytest.raises(ValueError, match="input channels"):
        net.forward(torch.zeros((1, 32, 32, 3), device="cuda"))
