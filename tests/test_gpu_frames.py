"""GPU parity of the frame-format kernels (include/dvsg_amd.h: dvsg_frames_*, dvsg_window_gather_f32)
and of the uint8-in / uint8-out clip loop against the oracle.  Byte and index work: bit-exact."""
import numpy as np
import pytest

import inputs
from oracle import frames as oframes
from oracle import model as omodel

pytestmark = pytest.mark.gpu


def _call(name, *args):
    from coupe.dvsg_amd import _lib
    _lib.call(name, *args)


@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 5, 7), (3, 32, 48), (1, 37, 101)])
@pytest.mark.parametrize("flip", [0, 1])
def test_u8_to_f32_is_the_float64_division(shape, flip):
    import torch
    n, h, w = shape
    u = np.random.default_rng(n * 131 + w).integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    src = torch.from_numpy(u).cuda()
    dst = torch.empty((n, h, w, 3), dtype=torch.float32, device="cuda")
    _call("dvsg_frames_u8_to_f32", src.data_ptr(), n * h * w, flip, dst.data_ptr(), 0)
    want = ((u[..., ::-1] if flip else u) / 255.).astype(np.float32)
    assert np.array_equal(dst.cpu().numpy(), want)
    # a source pointer that is not 4-byte aligned takes the byte path
    if n * h * w > 1:
        flat = src.reshape(-1)[3:]
        d2 = torch.empty((n * h * w - 1) * 3, dtype=torch.float32, device="cuda")
        _call("dvsg_frames_u8_to_f32", flat.data_ptr(), n * h * w - 1, flip, d2.data_ptr(), 0)
        assert np.array_equal(d2.cpu().numpy().reshape(-1, 3), want.reshape(-1, 3)[1:])


@pytest.mark.parametrize("B,S,h,w", [(1, 7, 32, 48), (3, 7, 9, 11), (2, 3, 5, 5), (1, 1, 1, 1)])
def test_window_gather_concatenates_frames_on_the_channel_axis(B, S, h, w):
    import torch
    rng = np.random.default_rng(B * 17 + S)
    n_pool = 9
    pool = rng.uniform(0, 1, (n_pool, h, w, 3)).astype(np.float32)
    idx = rng.integers(0, n_pool, (B, S)).astype(np.int32)
    idx[0, 0] = n_pool + 3            # outside the pool: reads as zeros, never faults
    if S > 1:
        idx[-1, 1] = -1
    got = torch.full((B, h, w, 3 * S), -7.0, device="cuda")
    d_pool, d_idx = torch.from_numpy(pool).cuda(), torch.from_numpy(idx).cuda()
    _call("dvsg_window_gather_f32", d_pool.data_ptr(), n_pool, h, w, d_idx.data_ptr(), B, S, got.data_ptr(), 0)
    padded = np.concatenate([pool, np.zeros((1, h, w, 3), np.float32)])
    safe = np.where((idx >= 0) & (idx < n_pool), idx, n_pool)
    want = np.stack([np.concatenate(padded[safe[b]], axis=2) for b in range(B)])      # eval.py:103-104
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("flip", [0, 1])
def test_f32_to_u8_truncates_the_float64_product_side_by_side(flip):
    import torch
    n, h, w = 2, 7, 13
    rng = np.random.default_rng(77)
    a = rng.uniform(0, 1, (n, h, w, 3)).astype(np.float32)
    b = (rng.integers(0, 256, (n, h, w, 3)) / 255.).astype(np.float32)     # exact pixel values survive
    b[0, 0, 0] = [1.5, -0.25, np.nan]                                      # saturate / zero instead of wrapping
    side = torch.zeros((n, h, 2 * w, 3), dtype=torch.uint8, device="cuda")
    for x, x0 in ((a, 0), (b, w)):
        d_x = torch.from_numpy(x).cuda()
        _call("dvsg_frames_f32_to_u8", d_x.data_ptr(), n, h, w, flip, side.data_ptr(), 2 * w, x0, 0)
        torch.cuda.synchronize()
    got = side.cpu().numpy()
    fa, fb = (a[..., ::-1], b[..., ::-1]) if flip else (a, b)
    assert np.array_equal(got[:, :, :w], oframes.to_uint8(fa))
    wb = oframes.to_uint8(np.clip(np.nan_to_num(fb, nan=0.0), 0.0, 1.0))
    assert np.array_equal(got[:, :, w:], wb)
    from coupe.dvsg_amd import DvsgError
    with pytest.raises(DvsgError, match="do not fit"):
        _call("dvsg_frames_f32_to_u8", side.data_ptr(), n, h, w, 0, side.data_ptr(), 2 * w, w + 1, 0)


@pytest.mark.parametrize("src,dst", [((48, 64), (32, 48)), ((20, 30), (40, 60)), ((17, 23), (32, 48)), ((32, 48), (32, 48)),
                                     ((1080, 1920), (288, 512))])
def test_resize_matches_the_cv2_restatement(src, dst):
    import torch
    n = 2
    u = np.random.default_rng(src[0]).integers(0, 256, (n, src[0], src[1], 3), dtype=np.uint8)
    out = torch.empty((n, dst[0], dst[1], 3), dtype=torch.float32, device="cuda")
    d_u = torch.from_numpy(u).cuda()
    side = torch.zeros((n, dst[0], 2 * dst[1] + 1, 3), dtype=torch.uint8, device="cuda")
    _call("dvsg_frames_resize_u8_f32", d_u.data_ptr(), n, src[0], src[1], 1, out.data_ptr(), dst[0], dst[1],
          side.data_ptr(), 2 * dst[1] + 1, 1, 0)
    want = np.stack([oframes.read_frame(f, dst[1], dst[0]) for f in u])                       # eval.py:76-81 (BGR in)
    assert np.array_equal(out.cpu().numpy(), want.astype(np.float32))
    got = side.cpu().numpy()
    assert np.array_equal(got[:, :, 1:dst[1] + 1], oframes.to_uint8(want)[..., ::-1])         # eval.py:112-113
    assert not got[:, :, 0].any() and not got[:, :, dst[1] + 1:].any()


def test_uint8_clip_in_uint8_clip_out(synthetic_weights):
    """eval.py:76-124 from decoded BGR frames of another size to the side-by-side BGR output."""
    from coupe.dvsg_amd.clip import stabilize_clip
    from coupe.dvsg_amd.model import StabNet
    H, W, N = 32, 48, 4
    raw = (inputs.smooth_frames(5001, N, 40, 60) * 255).astype(np.uint8)[..., ::-1].copy()    # "decoded" BGR
    model = StabNet(H, W).load_weights(synthetic_weights)
    out, side = stabilize_clip(model, None, raw, side_by_side=True, channel_order="bgr", as_uint8=True)
    assert out.dtype == np.uint8 and out.shape == (N, H, W, 3) and side.shape == (N, H, 2 * W, 3)
    frames = np.stack([oframes.read_frame(f, W, H) for f in raw])
    ref, rside = omodel.eval_clip(synthetic_weights, frames, H, W)
    rside = rside[..., ::-1]                                                                  # eval.py:113
    assert np.array_equal(side[:, :, :W], rside[:, :, :W])          # left half: the resized unstable input
    assert np.array_equal(side[:, :, W:], out)
    diff = np.abs(out.astype(int) - rside[:, :, W:].astype(int))
    assert (diff > 1).mean() < 0.01 and np.median(diff) == 0
    # float frames in, float frames out: same sequence
    fout = stabilize_clip(model, None, frames)
    assert fout.dtype == np.float32 and np.abs(fout - ref).max() < 2e-2 and np.median(np.abs(fout - ref)) < 1e-5
    fout64, fside = stabilize_clip(model, None, frames, side_by_side=True)                   # float64 history in
    assert np.array_equal(fout64, fout) and np.array_equal(fside[:, :, :W], rside[:, :, :W, ::-1])
    with pytest.raises(ValueError):
        stabilize_clip(model, None, frames[:, :-1])


def test_teacher_forced_clip_without_a_mask_is_model_py_on_the_same_windows(synthetic_weights):
    """`mask_H=None`: model.py's graph (model.py:98-123) on eval_train.py's teacher-forced windows (:137-165) -- NOT
    eval_train.py's own graph, whose CNN input is masked (tests/test_gpu_masked.py covers that one).  The oracle runs
    the masked graph with an identity homography per step, whose mask is exactly one."""
    from coupe.dvsg_amd.clip import stabilize_clip_teacher_forced
    from coupe.dvsg_amd.model import StabNet
    H, W, N = 32, 48, 37
    stab = inputs.smooth_frames(6001, N, H, W)
    unstab = np.roll(stab, 2, axis=2) * 0.9 + 0.05
    model = StabNet(H, W).load_weights(synthetic_weights)
    ident = np.tile(omodel.RANDOM_MASK_OFFSET, (N - 32, 1))
    ref = omodel.eval_train_clip(synthetic_weights, unstab, stab, H, W, ident)
    for batch in (2, 16):
        out = stabilize_clip_teacher_forced(model, unstab, stab, batch=batch, mask_H=None)
        assert out.shape == (N - 32, H, W, 3) and out.dtype == np.float32
        assert np.abs(out - ref).max() < 2e-2 and np.median(np.abs(out - ref)) < 1e-5
    u8 = stabilize_clip_teacher_forced(model, (unstab * 255).astype(np.uint8), (stab * 255).astype(np.uint8), batch=3,
                                       as_uint8=True, mask_H=None)
    assert u8.dtype == np.uint8 and u8.shape == (N - 32, H, W, 3)
    ref8 = omodel.eval_train_clip(synthetic_weights, (unstab * 255).astype(np.uint8) / 255.,
                                  (stab * 255).astype(np.uint8) / 255., H, W, ident)
    diff = np.abs(u8.astype(int) - oframes.to_uint8(ref8).astype(int))
    assert (diff > 1).mean() < 0.01 and np.median(diff) == 0


def _tf_worker(rank, world, port, out_dir):
    import os
    import torch
    import torch.distributed as dist
    from coupe.dvsg_amd.clip import stabilize_clip_teacher_forced
    from coupe.dvsg_amd.model import StabNet
    from coupe.dvsg_amd.weights import make_synthetic_weights
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)                       # one-GPU box: both ranks share the card, gather over gloo
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H, W, N = 32, 48, 39
        stab = inputs.smooth_frames(6001, N, H, W)
        unstab = np.roll(stab, 2, axis=2) * 0.9 + 0.05
        model = StabNet(H, W).load_weights(make_synthetic_weights(seed=0))
        out = stabilize_clip_teacher_forced(model, unstab, stab, batch=2, mask_H=inputs.mask_homographies(6003, N - 32))
        if rank == 0:
            np.save(os.path.join(out_dir, "out.npy"), out)
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def test_teacher_forced_clip_sharded_over_two_ranks(tmp_path, synthetic_weights):
    """The N > 1 path with the real kernels: two processes shard the 7 windows 4 + 3, no data-path
    collective, one gather -- same frames as a single process; eval_train.py's masked graph, each rank making the
    mask planes of its own windows from the shared table of homographies."""
    import socket
    import torch.multiprocessing as mp
    from coupe.dvsg_amd.clip import stabilize_clip_teacher_forced
    from coupe.dvsg_amd.model import StabNet
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_tf_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    H, W, N = 32, 48, 39
    stab = inputs.smooth_frames(6001, N, H, W)
    unstab = np.roll(stab, 2, axis=2) * 0.9 + 0.05
    single = stabilize_clip_teacher_forced(StabNet(H, W).load_weights(synthetic_weights), unstab, stab, batch=2,
                                           mask_H=inputs.mask_homographies(6003, N - 32))
    plain = stabilize_clip_teacher_forced(StabNet(H, W).load_weights(synthetic_weights), unstab, stab, batch=2, mask_H=None)
    assert np.abs(single - plain).max() > 1e-3                      # the mask is really in the sharded path
    got = np.load(tmp_path / "out.npy")
    assert got.shape == single.shape == (N - 32, H, W, 3)
    assert np.abs(got - single).max() <= 1e-6
