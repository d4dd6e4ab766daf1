"""Callers of the hot path: the eval.py clip loop and the multi-GPU window sharding.

* `stabilize_clip` reproduces the reference driver's frame loop (eval.py:76-124) with the clip
  resident in HBM from the uint8 frames in to the uint8 frames out: 32 copies of frame 0 are
  prepended, each step feeds the 7-frame dilated window k + [0,16,24,28,30,31,32]
  (config.py:48), and the stabilised frame is written back into the history (eval.py:116-120).
  That write-back makes frame t depend on stabilised frame t-1, so ONE clip cannot be sharded
  across GPUs ("replicas only": different clips on different GPUs).
* `shard_range` / `stabilize_windows_sharded` cover the case that does shard: independent
  windows (BASELINE.json configs[3]; the teacher-forced regime of eval_train.py).  Rank r owns
  a contiguous block of windows, there is no data-path collective, and the stabilised frames
  are gathered to one rank at the end (RCCL over xGMI on GPUs; any torch.distributed backend
  works -- the CPU tests use gloo).
"""
import numpy as np
import torch

SKIP_LENGTH = (0, 16, 24, 28, 30, 31, 32)   # config.py:48


def shard_range(n, world, rank):
    """Contiguous block [lo, hi) of `n` windows owned by `rank` (sizes differ by at most 1)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def stabilize_windows_sharded(run_fn, patches_t, u_t, batch=16, group=None, dst=0):
    """Stabilise `patches_t` [N,H,W,21] / `u_t` [N,H,W,3] (every rank passes the same N) with
    the windows sharded over the ranks of `group`; returns [N,H,W,3] on rank `dst`, None
    elsewhere.  `run_fn(patches, u) -> [b,H,W,3]` is the per-batch hot path (e.g.
    `lambda p, u: sess.run(outputs['s_t_pred'], {inputs['patches_t']: p, inputs['u_t']: u})`)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    N = patches_t.shape[0]
    lo, hi = shard_range(N, world, rank)
    outs = []
    for b0 in range(lo, hi, batch):
        b1 = min(hi, b0 + batch)
        outs.append(torch.as_tensor(run_fn(patches_t[b0:b1], u_t[b0:b1])))
    H, W = u_t.shape[1], u_t.shape[2]
    like = outs[0] if outs else torch.as_tensor(u_t[:0])
    local = torch.cat(outs, 0) if outs else like.new_zeros((0, H, W, 3))
    if world == 1:
        return local
    # gather needs equal shapes: pad every shard to the largest one
    cap = -(-N // world)
    padded = local.new_zeros((cap, H, W, 3))
    padded[:local.shape[0]] = local
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        rlo, rhi = shard_range(N, world, r)
        parts.append(bufs[r][:rhi - rlo])
    return torch.cat(parts, 0)


def window_index_table(n_frames, skip_length=SKIP_LENGTH):
    """The frame loop of eval.py:93-124 as an index table.  With the clip kept in a pool of
    2 N frames -- [0, N) the unstable inputs, [N, 2 N) the stabilised outputs -- entry [k, s] is
    the pool frame that window slot s holds at step k: slot k + skip[s] of the reference's
    padded history list, which is
      * the unstable frame k for the last slot (eval.py:103, sample_idx[-1]);
      * a stabilised frame for every slot the write-back of eval.py:116 already replaced;
      * one of the 32 prepended copies of frame 0 (eval.py:93-94) otherwise -- the unstable
        frame 0 at step 0, the stabilised frame 0 afterwards (eval.py:118-120).
    Returns int32 [N, S]."""
    skip = np.asarray(skip_length, dtype=np.int64)
    if skip.ndim != 1 or skip.size < 1 or skip[0] != 0 or np.any(np.diff(skip) <= 0):
        raise ValueError("skip_length must start at 0 and increase strictly (config.py:48)")
    N = int(n_frames)
    span = int(skip[-1])
    k = np.arange(N, dtype=np.int64)[:, None]
    j = k + skip[None, :]                                    # slot in the padded history
    table = np.where(j >= span, N + (j - span), np.where(k == 0, 0, N))
    table[:, -1] = k[:, 0]
    return table.astype(np.int32)


def stabilize_clip(model, session, frames, skip_length=SKIP_LENGTH, side_by_side=False, channel_order="rgb",
                   as_uint8=False):
    """eval.py:76-124 for one clip, entirely on the device.

    frames: [N,h0,w0,3], NumPy or torch.
      * uint8 -- raw decoded frames.  They are converted as eval.py:79-80 does (optional
        BGR->RGB with channel_order="bgr", / 255. in float64, cv2.resize to the model's (w, h)
        when the size differs) by `dvsg_frames_u8_to_f32` / `dvsg_frames_resize_u8_f32`;
      * float -- RGB in [0,1], already (model.h, model.w); rounded to float32 once, which is the
        cast TF applies to the fed window (eval.py:106-110).
    Video decode / encode (cv2.VideoCapture / VideoWriter) is host I/O and out of scope.

    The clip lives in one HBM pool [2N,h,w,3] float32 (unstable | stabilised).  Each step is
    `dvsg_window_gather_f32` (the 21-channel window, eval.py:103-104, picked through
    `window_index_table`) + `dvsg_stabilize_*` writing straight into the pool; the write-back of
    eval.py:116-120 is the index table, not a copy.  `session` is accepted for call-site
    symmetry with eval.py and not used.

    Returns the stabilised frames [N,h,w,3] -- float32, or uint8 (np.uint8(x * 255.),
    eval.py:112) with as_uint8 -- and, with side_by_side, also the reference's output video
    frames uint8 [N,h,2w,3] (unstable | stabilised, eval.py:112; BGR if channel_order="bgr",
    eval.py:113).  NumPy in -> NumPy out.
    """
    from . import _lib
    from ._tensor import device, ptr, stream
    if channel_order not in ("rgb", "bgr"):
        raise ValueError("channel_order must be 'rgb' or 'bgr'")
    if model.locnet is None:
        raise _lib.DvsgError("StabNet has no weights: call load_weights()/load_ckpt() first")
    flip = 1 if channel_order == "bgr" else 0
    host = not isinstance(frames, torch.Tensor)
    dev = device()
    fr = torch.as_tensor(np.ascontiguousarray(frames)) if host else frames
    if fr.dim() != 4 or fr.shape[3] != 3 or fr.shape[0] < 1:
        raise ValueError("frames must be [N,h,w,3]")
    N, h, w = int(fr.shape[0]), model.h, model.w
    S = len(skip_length)
    table = torch.from_numpy(window_index_table(N, skip_length)).to(dev)
    pool = torch.empty((2 * N, h, w, 3), dtype=torch.float32, device=dev)
    fr = fr.to(dev).contiguous()
    # the unstable half of the output video is np.uint8(float64 frame * 255.) (eval.py:112): rendered
    # from float64 wherever the float32 pool would not hold the same value
    side = torch.empty((N, h, 2 * w, 3), dtype=torch.uint8, device=dev) if side_by_side else None
    left_done = False
    if fr.dtype == torch.uint8:
        if tuple(fr.shape[1:3]) == (h, w):
            _lib.call("dvsg_frames_u8_to_f32", ptr(fr), N * h * w, flip, ptr(pool), stream())
        else:
            _lib.call("dvsg_frames_resize_u8_f32", ptr(fr), N, int(fr.shape[1]), int(fr.shape[2]), flip, ptr(pool),
                      h, w, ptr(side), 2 * w, 0, stream())
            left_done = side_by_side
    elif fr.dtype.is_floating_point:
        if tuple(fr.shape[1:3]) != (h, w):
            raise ValueError("float frames must already be [N,%d,%d,3] (StabNet(h, w) fixes the STN out_size)" % (h, w))
        pool[:N] = fr        # one rounding to float32 (the feed cast); a plain device copy for float32 input
        if side_by_side and fr.dtype == torch.float64:
            _lib.call("dvsg_frames_f64_to_u8", ptr(fr), N, h, w, flip, ptr(side), 2 * w, 0, stream())
            left_done = True
    else:
        raise TypeError("frames must be uint8 or floating point, got %s" % fr.dtype)
    patches = torch.empty((1, h, w, 3 * S), dtype=torch.float32, device=dev)
    F = torch.empty((1, model.param_dim, 2), dtype=torch.float32, device=dev)
    for k in range(N):                                                             # eval.py:101
        _lib.call("dvsg_window_gather_f32", ptr(pool), 2 * N, h, w, ptr(table[k]), 1, S, ptr(patches), stream())
        model.locnet.stabilize(patches, pool[k:k + 1], pool[N + k:N + k + 1], F,   # :106-110, :116
                               precision=model.precision)
    stab = pool[N:]
    if side_by_side:                                                               # eval.py:112-113
        if not left_done:   # uint8 / float32 input: the float32 pool holds the frame exactly
            _lib.call("dvsg_frames_f32_to_u8", ptr(pool), N, h, w, flip, ptr(side), 2 * w, 0, stream())
        _lib.call("dvsg_frames_f32_to_u8", ptr(stab), N, h, w, flip, ptr(side), 2 * w, w, stream())
    if as_uint8:
        out = torch.empty((N, h, w, 3), dtype=torch.uint8, device=dev)
        _lib.call("dvsg_frames_f32_to_u8", ptr(stab), N, h, w, flip, ptr(out), w, 0, stream())
    else:
        out = stab.clone()
    if host:
        out = out.cpu().numpy()
        side = side.cpu().numpy() if side is not None else None
    return (out, side) if side_by_side else out
