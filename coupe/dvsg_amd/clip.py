"""Callers of the hot path: the eval.py clip loop and the multi-GPU window sharding.

* `stabilize_clip` reproduces the reference driver's frame loop (eval.py:93-124) with the
  frame history resident on the device: 32 copies of frame 0 are prepended, each step feeds the
  7-frame dilated window k + [0,16,24,28,30,31,32] (config.py:48), and the stabilised frame is
  written back into the history (eval.py:116-120).  That write-back makes frame t depend on
  stabilised frame t-1, so ONE clip cannot be sharded across GPUs ("replicas only": different
  clips on different GPUs).
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


def stabilize_clip(model, session, frames, skip_length=SKIP_LENGTH, side_by_side=False):
    """eval.py:93-124 for one clip.  frames [N,h,w,3] float (RGB, /255, already resized: the
    cv2 decode / resize / MJPG write of eval.py:60-90,112-114 is host I/O and out of scope).

    Returns stabilised frames [N,h,w,3] float32 (NumPy in -> NumPy out) and, if asked, the
    reference's side-by-side uint8 [N,h,2w,3] (truncating cast of x*255, eval.py:112).
    """
    from ._tensor import device
    host = not isinstance(frames, torch.Tensor)
    dev = device()
    fr = torch.as_tensor(np.asarray(frames)) if host else frames
    # the reference keeps float64 frames (frame / 255., eval.py:80) and TF casts the fed
    # window to float32; stabilised float32 frames are written back into the float64 history
    hist = fr.to(device=dev, dtype=torch.float64)
    skip = torch.as_tensor(skip_length, device=dev, dtype=torch.long)
    span = int(skip_length[-1] - skip_length[0])
    hist = torch.cat([hist[:1].expand(span, -1, -1, -1), hist], 0).contiguous()   # :93-94
    ins, outs = model.inputs, model.outputs
    if ins is None:
        ins, outs = model.get_evaluation_model(len(skip_length))
    stab, sbs = [], []
    idx = skip.clone()
    for frame_idx in range(span, hist.shape[0]):                                   # :101
        window = hist[idx]                                                         # [7,h,w,3]  :103
        batch = window.permute(1, 2, 0, 3).reshape(1, window.shape[1], window.shape[2], -1)  # :104
        batch32 = batch.to(torch.float32).contiguous()
        s_t_pred = session.run(outs['s_t_pred'], {ins['patches_t']: batch32,
                                                  ins['u_t']: batch32[..., 18:].contiguous()})[0]  # :106-110
        if side_by_side:
            side = torch.cat([hist[idx[-1]], s_t_pred.to(torch.float64)], 1) * 255.0   # :112
            sbs.append(side.to(torch.uint8))
        hist[idx[-1]] = s_t_pred.to(torch.float64)                                 # :116
        if frame_idx == span:                                                      # :118-120
            hist[:span] = s_t_pred.to(torch.float64)
        stab.append(s_t_pred)
        idx = idx + 1                                                              # :124
    out = torch.stack(stab)
    side = torch.stack(sbs) if side_by_side else None
    if host:
        out = out.cpu().numpy()
        side = side.cpu().numpy() if side is not None else None
    return (out, side) if side_by_side else out
