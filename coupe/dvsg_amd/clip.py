"""Callers of the hot path: the eval.py clip loop and the multi-GPU window sharding.

* `stabilize_clip` reproduces the reference driver's frame loop (eval.py:76-124) with the clip
  resident in HBM from the uint8 frames in to the uint8 frames out: 32 copies of frame 0 are
  prepended, each step feeds the 7-frame dilated window k + [0,16,24,28,30,31,32]
  (config.py:48), and the stabilised frame is written back into the history (eval.py:116-120).
  That write-back makes frame t depend on stabilised frame t-1, so ONE clip cannot be sharded
  across GPUs ("replicas only": different clips on different GPUs).
* `shard_range` / `sharded_map` / `stabilize_windows_sharded` cover the case that does shard:
  independent windows (BASELINE.json configs[3]).  Rank r owns a contiguous block of windows,
  there is no data-path collective, and the stabilised frames are gathered to one rank at the end
  (RCCL over xGMI on GPUs; any torch.distributed backend works -- the CPU tests use gloo).
* `stabilize_clip_teacher_forced` is the reference's other driver, eval_train.py:115-165, whose
  history frames come from the ground-truth stable clip: its windows ARE independent, so a clip
  runs batched and sharded over the GPUs of a node.  That driver evaluates eval_train.py's OWN graph
  (:25-51), whose CNN input is multiplied by a random projective mask (:43-45, 53-64): `mask_H`.
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


def sharded_map(n, batch, produce, frame_shape, like, group=None, dst=0):
    """Run `produce(b0, b1) -> [b1-b0, *frame_shape]` over this rank's contiguous share of `n`
    independent units in batches of at most `batch`, then gather the results, in unit order, on
    rank `dst` (None elsewhere).  No collective on the data path; one exchange at the end, which every rank of
    the group must reach (ranks whose shard is empty included: they join its opening all-reduce and send nothing)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(n, world, rank)
    outs = []
    for b0 in range(lo, hi, batch):
        outs.append(torch.as_tensor(produce(b0, min(hi, b0 + batch))))
    ref = outs[0] if outs else like
    local = torch.cat(outs, 0) if outs else ref.new_zeros((0,) + tuple(frame_shape))
    if world == 1:
        return local
    # One exchange at the end, only real rows: every rank that owns windows sends its slab straight
    # to `dst` (its own xGMI link on RCCL), which receives each slab in place -- no padding to a
    # common shard size, no copy after the receive.
    on_host = local.is_cuda and dist.get_backend(group) != "nccl"
    back = local.device
    # EVERY rank enters one cheap collective first, whatever it owns.  A rank with an empty shard (n < world,
    # e.g. a 2-window clip on 8 GPUs) takes no part in the point-to-point exchange below, and on NCCL / RCCL a
    # `batch_isend_irecv` that is the FIRST communication of a group must be entered by all of its ranks (it
    # creates the communicator; otherwise the behaviour is undefined and can hang).  The all-reduce creates the
    # communicator with everybody present and doubles as a consistency check of the sharding: the shard sizes
    # the ranks computed must add up to n.
    # (its device follows the BACKEND, not `local`: a rank with an empty shard may hold a CPU `like` while the populated
    # ranks hold device tensors, and a CPU tensor in an NCCL / RCCL collective raises on that rank and hangs the others)
    if dist.get_backend(group) == "nccl":
        count_dev = torch.device("cuda", torch.cuda.current_device())
        if not local.is_cuda:
            local = local.to(count_dev)
            back = count_dev
    else:
        count_dev = torch.device("cpu")
    count = torch.tensor([local.shape[0]], dtype=torch.int64, device=count_dev)
    dist.all_reduce(count, op=dist.ReduceOp.SUM, group=group)
    if int(count.item()) != n:
        raise RuntimeError("sharded_map: the ranks hold %d units in total, expected %d (every rank must pass the "
                           "same n)" % (int(count.item()), n))
    if on_host:
        local = local.cpu()   # rehearsal backends (gloo) exchange through host memory; RCCL stays on the device
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    if rank != dst:
        if local.shape[0]:
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local.contiguous(), peer(dst), group)]):
                req.wait()
        return None
    full = local.new_empty((n,) + tuple(frame_shape))
    ops = []
    for r in range(world):
        rlo, rhi = shard_range(n, world, r)
        if r == dst:
            full[rlo:rhi] = local
        elif rhi > rlo:
            ops.append(dist.P2POp(dist.irecv, full[rlo:rhi], peer(r), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return full.to(back) if on_host else full


def stabilize_windows_sharded(run_fn, patches_t, u_t, batch=16, group=None, dst=0):
    """Stabilise `patches_t` [N,H,W,21] / `u_t` [N,H,W,3] (every rank passes the same N) with
    the windows sharded over the ranks of `group`; returns [N,H,W,3] on rank `dst`, None
    elsewhere.  `run_fn(patches, u) -> [b,H,W,3]` is the per-batch hot path (e.g.
    `lambda p, u: sess.run(outputs['s_t_pred'], {inputs['patches_t']: p, inputs['u_t']: u})`)."""
    return sharded_map(patches_t.shape[0], batch, lambda b0, b1: run_fn(patches_t[b0:b1], u_t[b0:b1]),
                       (u_t.shape[1], u_t.shape[2], 3), torch.as_tensor(u_t[:0]), group, dst)


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


def _check_window(model, S):
    """A window of S frames is 3 S channels; conv1 of the loaded checkpoint fixes that number."""
    if 3 * S != model.locnet.in_channels:
        raise ValueError("skip_length has %d entries (%d channels) but conv1 of the loaded checkpoint has %d "
                         "input channels" % (S, 3 * S, model.locnet.in_channels))


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

    The clip lives in one HBM pool [2N,h,w,3] float32 (unstable | stabilised).  Each step is ONE call,
    `dvsg_stabilize_ring_f32`: conv1 assembles the 21-channel window of eval.py:103-104 in its load stage
    from the pool frames `window_index_table` names, and the result is written straight into its pool
    slot; the write-back of eval.py:116-120 is the index table, not a copy.  `session` is accepted for call-site
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
    _check_window(model, S)
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
    F = torch.empty((1, model.param_dim, 2), dtype=torch.float32, device=dev)
    # one call per frame: conv1 picks the 7 window frames out of the pool through table[k] (eval.py:103-104 fused
    # into its load stage: no window tensor, no gather launch), the warp reads u_t = pool[table[k, 6]] = pool[k],
    # and the result lands in its history slot pool[N + k] (:116), which no slot of window k reads
    for k in range(N):                                                             # eval.py:101
        model.locnet.stabilize_ring(pool, table[k:k + 1], pool[N + k:N + k + 1], F, precision=model.precision)  # :106-110
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


def teacher_forced_index_table(n_frames, skip_length=SKIP_LENGTH):
    """eval_train.py:137-165 as an index table.  There the history is never the network's own
    output: the first 32 unstable frames are replaced by the stable (ground-truth) ones up front
    (:137-138) and every processed unstable frame is replaced by its stable twin (:162), so step k
    (frame k + 32) sees stable frames in every slot but the last.  With a pool of 2 N frames --
    [0, N) unstable, [N, 2N) stable -- entry [k, s] = N + k + skip[s], and [k, -1] = k + 32.
    The windows do not depend on each other: they shard over GPUs.  Returns int32 [N-32, S]."""
    skip = np.asarray(skip_length, dtype=np.int64)
    if skip.ndim != 1 or skip.size < 1 or skip[0] != 0 or np.any(np.diff(skip) <= 0):
        raise ValueError("skip_length must start at 0 and increase strictly (config.py:48)")
    N, span = int(n_frames), int(skip[-1])
    if N <= span:
        raise ValueError("eval_train.py needs more than %d frames, got %d" % (span, N))
    k = np.arange(N - span, dtype=np.int64)[:, None]
    table = N + k + skip[None, :]
    table[:, -1] = k[:, 0] + span
    return table.astype(np.int32)


def _frames_to_pool(fr, pool, h, w, flip, what):
    """uint8 / float frames [n,h0,w0,3] on the device -> float32 [n,h,w,3] slice of the pool."""
    from . import _lib
    from ._tensor import ptr, stream
    n = int(fr.shape[0])
    if fr.dtype == torch.uint8:
        if tuple(fr.shape[1:3]) == (h, w):
            _lib.call("dvsg_frames_u8_to_f32", ptr(fr), n * h * w, flip, ptr(pool), stream())
        else:
            _lib.call("dvsg_frames_resize_u8_f32", ptr(fr), n, int(fr.shape[1]), int(fr.shape[2]), flip, ptr(pool),
                      h, w, 0, 0, 0, stream())
    elif fr.dtype.is_floating_point:
        if tuple(fr.shape[1:3]) != (h, w):
            raise ValueError("float %s frames must already be [N,%d,%d,3]" % (what, h, w))
        pool.copy_(fr)
    else:
        raise TypeError("%s frames must be uint8 or floating point, got %s" % (what, fr.dtype))


def _mask_homographies(mask_H, n, dev):
    """eval_train.py:55-57 for the n steps of a clip: [n,8] float32 on `dev`, or None (no mask)."""
    from .model import draw_random_H
    if mask_H is None:
        return None
    if isinstance(mask_H, str):
        if mask_H != "random":
            raise ValueError("mask_H must be None, 'random', a torch.Generator or an [N-32,8] array")
        return draw_random_H(n, dev)
    if isinstance(mask_H, torch.Generator):
        return draw_random_H(n, dev, mask_H)
    Ht = torch.as_tensor(np.asarray(mask_H, dtype=np.float32) if not isinstance(mask_H, torch.Tensor) else mask_H)
    if tuple(Ht.shape) != (n, 8):
        raise ValueError("mask_H must be [%d,8] (one homography per stabilised frame), got %s" % (n, tuple(Ht.shape)))
    return Ht.to(dev, torch.float32).contiguous()


def stabilize_clip_teacher_forced(model, unstable, stable, batch=16, skip_length=SKIP_LENGTH, channel_order="rgb",
                                  as_uint8=False, group=None, dst=0, mask_H="random"):
    """eval_train.py:115-165 for one pair of clips: every unstable frame k >= 32 is stabilised from
    the window [stable k-32, k-16, k-8, k-4, k-2, k-1 | unstable k].  The windows are independent,
    so they run in batches of `batch` and -- with torch.distributed initialised -- shard over the
    ranks of `group` with no data-path collective; the stabilised frames are gathered on rank
    `dst` (BASELINE.json configs[3]).  Every rank passes the same clips (frame formats as in
    `stabilize_clip`) and keeps the whole pool in its HBM.

    mask_H: eval_train.py evaluates its OWN graph (:25-51, :86), in which the six history frames the CNN sees
    are multiplied by `random_mask` (:43-45, 53-64) -- a fresh random near-identity homography per `sess.run`.
      * "random" (default: what eval_train.py does) -- drawn with torch.rand per step (other VALUES than
        tf.random_uniform would give, the same distribution); a `torch.Generator` -- the same, reproducibly
        (a CPU generator seeded alike on every rank gives every rank the same table);
      * an array [N-32,8] -- the homography of each step, AFTER the scale / identity offset of :56-57 (the parity
        tests pin the loop against the oracle this way);
      * None -- no mask: model.py's graph (model.py:98-123) on the teacher-forced windows.  NOT what eval_train.py
        computes; kept for callers that want the regressor's unmasked prediction on ground-truth history.
    The mask is ONE [b,h,w] plane per batch (`dvsg_random_mask_plane_f32`), multiplied into the history channels
    inside conv1's load stage (`dvsg_stabilize_ring_masked_{f32,u8}`); the warp samples the unmasked frame (:48).

    Returns [N-32,h,w,3] float32 (or uint8 with as_uint8) on rank `dst` -- NumPy if the clips were
    NumPy -- and None on the other ranks."""
    from . import _lib
    from ._tensor import device, ptr, stream
    from .networks import random_mask_plane
    if channel_order not in ("rgb", "bgr"):
        raise ValueError("channel_order must be 'rgb' or 'bgr'")
    if model.locnet is None:
        raise _lib.DvsgError("StabNet has no weights: call load_weights()/load_ckpt() first")
    flip = 1 if channel_order == "bgr" else 0
    host = not isinstance(unstable, torch.Tensor)
    dev = device()
    un = (torch.as_tensor(np.ascontiguousarray(unstable)) if host else unstable).to(dev).contiguous()
    st = (torch.as_tensor(np.ascontiguousarray(stable)) if not isinstance(stable, torch.Tensor) else stable).to(dev).contiguous()
    if un.dim() != 4 or un.shape[3] != 3 or st.shape != un.shape:
        raise ValueError("unstable and stable clips must both be [N,h,w,3]")
    N, h, w, S = int(un.shape[0]), model.h, model.w, len(skip_length)
    _check_window(model, S)
    span = int(skip_length[-1])
    table = torch.from_numpy(teacher_forced_index_table(N, skip_length)).to(dev)
    Ht = _mask_homographies(mask_H, N - span, dev)
    if un.dtype == torch.uint8 and st.dtype == torch.uint8 and tuple(un.shape[1:3]) == (h, w) and not flip:
        pool = torch.cat([un, st], 0)        # the raw frames ARE the ring: 3 bytes per pixel in HBM
    else:
        pool = torch.empty((2 * N, h, w, 3), dtype=torch.float32, device=dev)
        _frames_to_pool(un, pool[:N], h, w, flip, "unstable")
        _frames_to_pool(st, pool[N:], h, w, flip, "stable")
    F = torch.empty((batch, model.param_dim, 2), dtype=torch.float32, device=dev)

    def produce(b0, b1):
        b = b1 - b0
        out = torch.empty((b, h, w, 3), dtype=torch.float32, device=dev)
        # windows b0..b1 straight from the pool (uint8 when the clips came as same-size RGB uint8 frames: the / 255. of
        # eval_train.py's frame reader happens in conv1's load stage); u_t of window k is pool frame table[k, 6] = k + 32
        plane = random_mask_plane(Ht[b0:b1], h, w) if Ht is not None else None      # eval_train.py:43 (one plane per window)
        model.locnet.stabilize_ring(pool, table[b0:b1], out, F[:b], precision=model.precision, mask=plane)
        if not as_uint8:
            return out
        out8 = torch.empty((b, h, w, 3), dtype=torch.uint8, device=dev)
        _lib.call("dvsg_frames_f32_to_u8", ptr(out), b, h, w, flip, ptr(out8), w, 0, stream())
        return out8

    like = torch.empty((0, h, w, 3), dtype=torch.uint8 if as_uint8 else torch.float32, device=dev)
    res = sharded_map(N - span, batch, produce, (h, w, 3), like, group, dst)
    if res is not None and host:
        res = res.cpu().numpy()
    return res
