"""GPU: the frame ring (include/dvsg_amd.h dvsg_stabilize_ring_*).  conv1 assembles the 7-frame
window of eval.py:103-104 in its load stage from a pool of RGB frames and -- for a uint8 pool -- applies the
`/ 255.` of eval.py:80 there.  Index and byte work: the bar is BIT equality with the path it replaces
(dvsg_window_gather_f32 + dvsg_stabilize_*, dvsg_frames_u8_to_f32 for uint8), in every precision."""
import numpy as np
import pytest

import inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net(synthetic_weights):
    import torch
    assert torch.cuda.is_available()
    from coupe.dvsg_amd.networks import LocNet
    return LocNet(synthetic_weights)


def _gather(pool, table):
    """The window tensor the ring replaces: dvsg_window_gather_f32 (eval.py:103-104)."""
    import torch
    from coupe.dvsg_amd import _lib
    B, (n, H, W, _) = table.shape[0], pool.shape
    out = torch.empty((B, H, W, 21), dtype=torch.float32, device=pool.device)
    _lib.call("dvsg_window_gather_f32", pool.data_ptr(), n, H, W, table.data_ptr(), B, 7, out.data_ptr(),
              torch.cuda.current_stream().cuda_stream)
    return out


def _pool_and_table(n, B, H, W, seed, dtype="f32"):
    import torch
    rng = np.random.default_rng(seed)
    if dtype == "u8":
        pool = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    else:
        pool = rng.uniform(0.0, 1.0, (n, H, W, 3)).astype(np.float32)
    table = rng.integers(0, n, (B, 7)).astype(np.int32)
    return torch.from_numpy(pool).cuda(), torch.from_numpy(table).cuda()


# aligned rows (W % 4 == 0), ragged rows, several 128-pixel tiles per output row, frames down to 1x1
SHAPES = [(2, 64, 96), (1, 37, 53), (3, 20, 4), (1, 8, 8), (1, 1, 1), (2, 30, 600), (1, 5, 301)]


@pytest.mark.parametrize("precision", ["f32", "f16", "f32s", "f32x3"])
@pytest.mark.parametrize("B,H,W", SHAPES)
def test_conv1_from_a_float_ring_is_the_gathered_window(net, precision, B, H, W):
    """conv1 (+ fused scale_RGB) and the max pool from the ring, against the same kernels on the gathered window."""
    import torch
    pool, table = _pool_and_table(11, B, H, W, 100 + H)
    if B > 1:
        table[1, 2] = 11          # outside the pool: a frame of zeros, like dvsg_window_gather_f32
        table[0, 5] = -3
    x = _gather(pool, table)
    for stage in (0, 1):
        got = net.forward_ring(pool, table, precision=precision, stage=stage)
        want = net.tap(x, stage, precision=precision)
        assert got.shape == want.shape
        assert torch.equal(got, want), "%s stage %d: max diff %g" % (precision, stage, float((got - want).abs().max()))
    assert torch.equal(net.forward_ring(pool, table, precision=precision), net.forward(x, precision=precision))


@pytest.mark.parametrize("precision", ["f32", "f16", "f32s", "f32x3"])
@pytest.mark.parametrize("B,H,W", SHAPES)
def test_uint8_ring_is_the_float_ring_of_the_converted_frames(net, precision, B, H, W):
    """eval.py:80 (`frame / 255.`, float64, fed as float32) fused: float32(v / 255.) * 255 == v for every byte, so
    conv1's scaled input float(v) - mean equals the float path's bit for bit (tests/test_frames_cpu.py)."""
    import torch
    from coupe.dvsg_amd import _lib
    pool8, table = _pool_and_table(9, B, H, W, 200 + W, dtype="u8")
    pool8[0] = torch.arange(256, dtype=torch.uint8).repeat(-(-H * W * 3 // 256))[:H * W * 3].reshape(H, W, 3)   # every byte value
    poolf = torch.empty(pool8.shape, dtype=torch.float32, device="cuda")
    _lib.call("dvsg_frames_u8_to_f32", pool8.data_ptr(), 9 * H * W, 0, poolf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(poolf.cpu().numpy(), (pool8.cpu().numpy() / 255.).astype(np.float32))
    for stage in (0, -1):
        got = net.forward_ring(pool8, table, precision=precision, stage=stage)
        want = net.forward_ring(poolf, table, precision=precision, stage=stage)
        assert torch.equal(got, want), "%s stage %d: max diff %g" % (precision, stage, float((got - want).abs().max()))
    # an unaligned view of the pool (frame 1 on, one byte in) takes the element-wise loads: same values
    if H * W > 1:
        flat = pool8.reshape(-1)[1:1 + 7 * H * W * 3].reshape(7, H, W, 3)
        t7 = (table % 7).contiguous()
        a = net.forward_ring(flat, t7, precision=precision, stage=0)
        b = net.forward_ring(flat.clone(), t7, precision=precision, stage=0)
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", ["f32", "u8"])
@pytest.mark.parametrize("variant", [4, 3, 0])
def test_float16_conv1_from_a_ring_at_full_frames(net, dtype, variant):
    """Full 1280x720 frames, a batch small enough for the two-rows-per-workgroup kernel (variant 4; 3 = the one-row
    kernel, 0 = what the launcher picks), several pools: every value equals the gathered window's.  Regression: the
    kernels that scatter a ring row into LDS zero the row images first, and without a barrier between the two a zeroed
    word could land AFTER a neighbour thread's float16 half of it -- a few pixels per launch at this size, never at the
    small shapes above (found by tools/soak_march.py, round 3)."""
    import torch
    from coupe.dvsg_amd import _lib
    B, H, W = 4, 720, 1280
    gen = torch.Generator(device="cuda").manual_seed(7)
    try:
        for rep in range(6):
            if dtype == "u8":
                pool = torch.randint(0, 256, (7 * B, H, W, 3), generator=gen, device="cuda", dtype=torch.uint8)
                fr = torch.empty((7 * B, H, W, 3), device="cuda")
                _lib.call("dvsg_frames_u8_to_f32", pool.data_ptr(), pool.numel() // 3, 0, fr.data_ptr(),
                          torch.cuda.current_stream().cuda_stream)
            else:
                pool = fr = torch.rand((7 * B, H, W, 3), generator=gen, device="cuda")
            table = torch.randint(0, 7 * B, (B, 7), generator=gen, device="cuda", dtype=torch.int32)
            x = _gather(fr, table)
            _lib.call("dvsg_debug_set_option", b"conv1_variant", variant)
            got = net.forward_ring(pool, table, precision="f16", stage=0)
            want = net.tap(x, 0, precision="f16")
            bad = int((got != want).sum())
            assert bad == 0, "pool %d: %d values differ, max %g" % (rep, bad, float((got - want).abs().max()))
    finally:
        _lib.call("dvsg_debug_set_option", b"conv1_variant", 0)


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_stabilize_ring_is_gather_plus_stabilize(net, precision):
    """The whole evaluation graph from the ring: F_t, the source grid and the warped frames of
    dvsg_stabilize_ring_{f32,u8} against dvsg_window_gather_f32 + dvsg_stabilize_* on the same frames."""
    import torch
    from coupe.dvsg_amd import _lib
    B, H, W, n = 3, 72, 128, 12
    frames = np.concatenate([inputs.smooth_frames(300 + i, 4, H, W) for i in range(3)])
    pool8 = torch.from_numpy((frames * 255).astype(np.uint8)).cuda()
    poolf = torch.empty(pool8.shape, dtype=torch.float32, device="cuda")
    _lib.call("dvsg_frames_u8_to_f32", pool8.data_ptr(), n * H * W, 0, poolf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    table = torch.from_numpy(np.random.default_rng(5).integers(0, n, (B, 7)).astype(np.int32)).cuda()
    x = _gather(poolf, table)
    u = x[..., 18:].contiguous()

    def outs():
        return (torch.empty((B, H, W, 3), device="cuda"), torch.empty((B, 25, 2), device="cuda"),
                torch.empty((B * H * W,), device="cuda"), torch.empty((B * H * W,), device="cuda"))
    want = outs()
    net.stabilize(x, u, *want, precision=precision)
    for pool in (poolf, pool8):
        got = outs()
        net.stabilize_ring(pool, table, *got, precision=precision)
        torch.cuda.synchronize()
        for g, w_, name in zip(got, want, ("s_t_pred", "F_t", "x_s", "y_s")):
            assert torch.equal(g, w_), "%s (%s pool): max diff %g" % (name, pool.dtype, float((g - w_).abs().max()))


def test_clip_loop_through_the_ring_is_the_gathered_loop(synthetic_weights):
    """`stabilize_clip` (one dvsg_stabilize_ring_f32 call per frame, eval.py:101-124) against the loop it replaced --
    dvsg_window_gather_f32 + dvsg_stabilize_f32 per frame -- on the 40-frame golden clip: bit-identical."""
    import torch
    from coupe.dvsg_amd.clip import stabilize_clip, window_index_table
    from coupe.dvsg_amd.model import Session, StabNet
    N, H, W = 40, 32, 48
    frames = inputs.smooth_frames(3001, N, H, W)
    model = StabNet(H, W).load_weights(synthetic_weights)
    model.get_evaluation_model(7)
    got = stabilize_clip(model, Session(), frames)
    table = torch.from_numpy(window_index_table(N)).cuda()
    pool = torch.empty((2 * N, H, W, 3), device="cuda")
    pool[:N] = torch.from_numpy(frames).cuda()
    F = torch.empty((1, 25, 2), device="cuda")
    for k in range(N):
        x = _gather(pool, table[k:k + 1])
        model.locnet.stabilize(x, pool[k:k + 1], pool[N + k:N + k + 1], F)
    assert np.array_equal(got, pool[N:].cpu().numpy())


def test_ring_rejects_bad_arguments(net):
    import torch
    from coupe.dvsg_amd import DvsgError, _lib
    pool, table = _pool_and_table(4, 1, 16, 16, 1)
    out, F = torch.empty((1, 16, 16, 3), device="cuda"), torch.empty((1, 25, 2), device="cuda")
    with pytest.raises(ValueError):
        net.stabilize_ring(pool, table.long(), out, F)
    with pytest.raises(ValueError):
        net.stabilize_ring(pool[..., :2].contiguous(), table, out, F)
    ws, nbytes = net.workspace(1, 16, 16)
    with pytest.raises(DvsgError, match="precision"):
        _lib.call("dvsg_stabilize_ring_f32", net.handle, 7, pool.data_ptr(), 4, table.data_ptr(), 1, 16, 16, out.data_ptr(),
                  F.data_ptr(), 0, 0, ws.data_ptr(), nbytes, 0)
    with pytest.raises(DvsgError, match="NULL"):
        _lib.call("dvsg_stabilize_ring_f32", net.handle, 0, pool.data_ptr(), 4, 0, 1, 16, 16, out.data_ptr(),
                  F.data_ptr(), 0, 0, ws.data_ptr(), nbytes, 0)


@pytest.mark.parametrize("precision,variant", [("f32", 0), ("f16", 0), ("f16", 5)])
def test_a_ring_step_replays_from_a_captured_hip_graph(net, precision, variant):
    """`dvsg_stabilize_ring_*` allocates and synchronises nothing: it is captured into a HIP graph as it stands -- the
    marching float16 conv1 kernel (conv1_variant 5 forces it at this size) with its role-split waves and hand-counted
    barriers included -- and replays bit for bit."""
    import torch
    from coupe.dvsg_amd import _lib
    B, H, W, n = 2, 96, 160, 14
    frames = inputs.smooth_frames(281, n, H, W)
    pool = torch.from_numpy((frames * 255).astype(np.uint8)).cuda()
    table = torch.from_numpy(np.random.default_rng(3).integers(0, n, (B, 7)).astype(np.int32)).cuda()
    out = torch.empty((B, H, W, 3), device="cuda")
    F = torch.empty((B, 25, 2), device="cuda")
    _lib.call("dvsg_debug_set_option", b"conv1_variant", variant)
    try:
        net.stabilize_ring(pool, table, out, F, precision=precision)
        torch.cuda.synchronize()
        ref_out, ref_F = out.clone(), F.clone()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            net.stabilize_ring(pool, table, out, F, precision=precision)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            net.stabilize_ring(pool, table, out, F, precision=precision)
        for _ in range(3):
            out.zero_()
            F.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(F, ref_F) and torch.equal(out, ref_out)
    finally:
        _lib.call("dvsg_debug_set_option", b"conv1_variant", 0)


def test_init_vars_feeds_the_gpu_model(tmp_path, synthetic_weights):
    """model.py:125-154 end to end: the trunk from a (synthetic) slim checkpoint, conv1 and the dense head from the
    caller's arrays, and the model that results is the oracle's on the merged weights."""
    import torch
    from coupe.dvsg_amd import tf_checkpoint as tfc
    from coupe.dvsg_amd.model import Session, StabNet
    from coupe.dvsg_amd.weights import PREFIX, make_synthetic_weights
    from oracle import networks as onet
    donor = make_synthetic_weights(seed=5)
    ckpt = {k[:-2][len(PREFIX):]: v for k, v in donor.items() if k.startswith(PREFIX + "resnet_v1_50/")}
    ckpt["resnet_v1_50/conv1/weights"] = np.zeros((7, 7, 3, 64), np.float32)
    path = str(tmp_path / "resnet_v1_50.ckpt")
    tfc.write_bundle(path, ckpt)
    H, W = 40, 64
    model = StabNet(H, W).load_weights(synthetic_weights)      # the variables exist (tf.global_variables_initializer)
    sess = Session()
    assert model.init_vars(sess, ckpt_path=path) is model      # model.py:125: init_vars(sess)
    ins, outs = model.get_evaluation_model(7)
    x = inputs.window_frames(291, 1, H, W)
    F = Session().run(outs["F_t"], {ins["patches_t"]: x, ins["u_t"]: x[..., 18:]})
    merged = tfc.init_from_slim_checkpoint(synthetic_weights, path)
    assert np.abs(F - onet.localizationNet(x, 25, merged)).max() <= 1e-5
    assert np.abs(F - onet.localizationNet(x, 25, synthetic_weights)).max() > 1e-4      # it really is another network
    F3 = Session().run(outs["F_t"], {ins["patches_t"]: x, ins["u_t"]: x[..., 18:]})
    model3 = StabNet(H, W).init_vars(synthetic_weights, ckpt_path=path)                 # round 3's call form still works
    ins3, outs3 = model3.get_evaluation_model(7)
    assert np.array_equal(F3, Session().run(outs3["F_t"], {ins3["patches_t"]: x, ins3["u_t"]: x[..., 18:]}))
    with pytest.raises(Exception):
        StabNet(H, W).init_vars(sess, ckpt_path=path)                                   # nothing to initialise
