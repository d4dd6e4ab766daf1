"""Drop-in for the inference part of the reference's networks.py on MI355X.

`scale_RGB(rgb)` (networks.py:6-16) and `localizationNet(input, param_dim, is_train=False,
reuse=False, scope=...) -> [B, param_dim, 2]` (networks.py:30-46).  The reference builds TF
variables under a scope and fills them from a checkpoint through the session; here the
scope maps to a `LocNet` handle created from the same checkpoint arrays
(`load_localizationNet`).  Training (`is_train=True`) and the correlationNet branch are out
of scope.
"""
import ctypes

import numpy as np

from . import _lib, weights as _weights
from ._tensor import as_dev, device, empty, like_input, ptr, stream

_nets = {}


def scale_RGB(rgb):
    """networks.py:6-16 (standalone op; inside localizationNet it is fused into conv1)."""
    t = as_dev(rgb)
    B, H, W, C = t.shape
    out = empty(t.shape, t)
    _lib.call("dvsg_scale_rgb_f32", ptr(t), B, H, W, C, ptr(out), stream())
    return like_input(out, rgb)


def _entry(base, precision):
    """C entry point for a precision: "f32" (exact float32 matrix cores, the reference's arithmetic),
    "f32s" (float32 storage and accumulation, products from two float16 pieces per operand: 22 significant
    bits for |x| >= 2^-3, an absolute step of 2^-24 below -- include/dvsg_amd.h), "f32x3" (float32 tensors and accumulation,
    conv products from three bfloat16 pieces per operand: all 24 bits of both operands at any magnitude) or "f16" (float16
    activations, hi / lo float16 weight pairs)."""
    if precision not in ("f32", "f32s", "f32x3", "f16"):
        raise ValueError("precision must be 'f32', 'f32s', 'f32x3' or 'f16', got %r" % (precision,))
    return "%s_%s" % (base, precision)


class LocNet(object):
    """Owns the device-side network (`dvsg_locnet_t`) and a growable workspace."""

    def __init__(self, weights):
        device()
        w = _weights.validate(weights)
        names = sorted(w)
        n = len(names)
        self._keep = [np.ascontiguousarray(w[k], dtype=np.float32) for k in names]
        c_names = (ctypes.c_char_p * n)(*[k.encode() for k in names])
        c_data = (ctypes.c_void_p * n)(*[a.ctypes.data for a in self._keep])
        c_nd = (ctypes.c_int * n)(*[a.ndim for a in self._keep])
        dims = np.ones((n, 4), dtype=np.int64)
        for i, a in enumerate(self._keep):
            dims[i, :a.ndim] = a.shape
        handle = ctypes.c_void_p()
        _lib.call("dvsg_locnet_create", n, c_names, c_data, c_nd,
                  dims.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.byref(handle))
        self.handle = handle
        self._keep = None  # the library copied everything to the device
        self._ws = None
        self.in_channels = _lib.load().dvsg_locnet_in_channels(self.handle)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().dvsg_locnet_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def workspace(self, B, H, W, slot=0):
        need = ctypes.c_size_t()
        _lib.call("dvsg_locnet_workspace_bytes", self.handle, B, H, W, ctypes.byref(need))
        if self._ws is None:
            self._ws = {}
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < need.value:
            import torch
            self._ws[slot] = None
            ws = self._ws[slot] = torch.empty(need.value, dtype=torch.uint8, device=device())
        return ws, need.value

    def stabilize(self, patches, u_t, out, F, xs=None, ys=None, n_streams=1, precision="f32", mask=None):
        """`dvsg_stabilize_f32` on device tensors (with `mask` [B,H,W]: `dvsg_stabilize_masked_f32`, eval_train.py's
        graph -- the plane multiplies the 18 history channels inside conv1's load stage), optionally with the batch split over
        `n_streams` side streams: every conv launch covers the chip in a few rounds of tiles and
        its last round is only partly full; launches from two independent half batches fill each
        other's tails (+4 % at B=16 720p before the stream-K tail, <1 % since).  Results do not
        depend on the split beyond float32 re-association (samples are independent); the
        caller's stream sees one fork / join."""
        import torch
        fn = _entry("dvsg_stabilize", precision)
        if u_t.dim() != 4 or u_t.shape[3] != 3:
            raise ValueError("u_t must be [B,H,W,3], got %s" % (tuple(u_t.shape),))
        B, H, W, _ = u_t.shape
        # the C entry point takes no channel count: conv1 reads `in_channels` floats per pixel, so a
        # narrower window (e.g. a 5-entry skip_length) would be read past its end
        if tuple(patches.shape) != (B, H, W, self.in_channels):
            raise ValueError("patches_t must be [%d,%d,%d,%d] (conv1 of the loaded checkpoint has %d input "
                             "channels), got %s" % (B, H, W, self.in_channels, self.in_channels, tuple(patches.shape)))
        if tuple(out.shape) != (B, H, W, 3) or F.numel() != B * 50:
            raise ValueError("out must be [B,H,W,3] and F_t [B,25,2]")
        for name, t, n in (("patches_t", patches, None), ("u_t", u_t, None), ("s_t_pred", out, None), ("F_t", F, None),
                           ("x_s", xs, B * H * W), ("y_s", ys, B * H * W)):
            if t is None:
                continue
            if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                raise ValueError("%s must be a contiguous float32 device tensor" % name)
            if n is not None and t.numel() != n:
                raise ValueError("%s must hold B*H*W = %d values" % (name, n))
        if mask is not None:
            self._check_mask(mask, B, H, W)
            ws, nbytes = self.workspace(B, H, W)
            _lib.call("dvsg_stabilize_masked_f32", self.handle, self._PRECISION_CODE[precision], ptr(patches), ptr(u_t),
                      ptr(mask), B, H, W, ptr(out), ptr(F), ptr(xs), ptr(ys), ptr(ws), nbytes, stream())
            return
        if n_streams <= 1 or B < 2 * n_streams:
            ws, nbytes = self.workspace(B, H, W)
            _lib.call(fn, self.handle, ptr(patches), ptr(u_t), B, H, W, ptr(out), ptr(F),
                      ptr(xs), ptr(ys), ptr(ws), nbytes, stream())
            return
        if getattr(self, "_side", None) is None or len(self._side) != n_streams:
            self._side = [torch.cuda.Stream() for _ in range(n_streams)]
        cur = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(cur)
        per = -(-B // n_streams)
        for i, side in enumerate(self._side):
            b0, b1 = i * per, min(B, (i + 1) * per)
            if b0 >= b1:
                break
            side.wait_event(fork)
            ws, nbytes = self.workspace(b1 - b0, H, W, slot=1 + i)
            _lib.call(fn, self.handle, ptr(patches[b0:b1]), ptr(u_t[b0:b1]), b1 - b0, H, W,
                      ptr(out[b0:b1]), ptr(F[b0:b1]), ptr(xs[b0 * H * W:b1 * H * W]) if xs is not None else 0,
                      ptr(ys[b0 * H * W:b1 * H * W]) if ys is not None else 0, ptr(ws), nbytes, side.cuda_stream)
            join = torch.cuda.Event()
            join.record(side)
            cur.wait_event(join)

    _PRECISION_CODE = {"f32": 0, "f16": 1, "f32s": 2, "f32x3": 3}   # DVSG_PRECISION_* of include/dvsg_amd.h

    def _check_ring(self, pool, table):
        import torch
        if self.in_channels != 21:
            raise ValueError("a frame ring feeds 7-frame windows; conv1 of the loaded checkpoint has %d input channels"
                             % self.in_channels)
        if pool.dim() != 4 or pool.shape[3] != 3 or pool.dtype not in (torch.float32, torch.uint8) \
                or not pool.is_cuda or not pool.is_contiguous():
            raise ValueError("pool must be a contiguous [n,H,W,3] float32 or uint8 device tensor")
        if table.dtype != torch.int32 or table.dim() != 2 or table.shape[1] != 7 or not table.is_cuda \
                or not table.is_contiguous():
            raise ValueError("table must be a contiguous [B,7] int32 device tensor (clip.window_index_table)")
        return int(table.shape[0]), int(pool.shape[1]), int(pool.shape[2])

    @staticmethod
    def _check_mask(mask, B, H, W):
        import torch
        if not isinstance(mask, torch.Tensor) or mask.dtype != torch.float32 or not mask.is_cuda or not mask.is_contiguous() \
                or tuple(mask.shape) != (B, H, W):
            raise ValueError("mask must be a contiguous float32 device tensor [%d,%d,%d] (random_mask_plane)" % (B, H, W))

    def random_mask_plane(self, H_theta, h, w):
        """`dvsg_random_mask_plane_f32`: model.py:156-167 / eval_train.py:53-64's mask for homographies H_theta [B,8]
        (the value after the scale and identity offset of :162-163) as ONE plane [B,h,w] -- the projective warp of an
        all-ones image is the same in each of the 18 history channels."""
        return random_mask_plane(H_theta, h, w)

    def stabilize_ring(self, pool, table, out, F, xs=None, ys=None, precision="f32", mask=None):
        """`dvsg_stabilize_ring_f32` / `_u8`: the evaluation graph on windows assembled inside conv1's load stage
        from the frame pool [n,H,W,3] (float32 in [0,1], or raw uint8) through the index table [B,7]
        (eval.py:103-104 and, for uint8, the / 255. of :80, fused); u_t of window b is pool frame table[b,6].
        With `mask` [B,H,W] (`random_mask_plane`): `dvsg_stabilize_ring_masked_*`, eval_train.py's graph."""
        import torch
        B, H, W = self._check_ring(pool, table)
        if precision not in self._PRECISION_CODE:
            raise ValueError("precision must be 'f32', 'f32s', 'f32x3' or 'f16', got %r" % (precision,))
        if tuple(out.shape) != (B, H, W, 3) or F.numel() != B * 50:
            raise ValueError("out must be [B,H,W,3] and F_t [B,25,2]")
        for name, t, n in (("s_t_pred", out, None), ("F_t", F, None), ("x_s", xs, B * H * W), ("y_s", ys, B * H * W)):
            if t is None:
                continue
            if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                raise ValueError("%s must be a contiguous float32 device tensor" % name)
            if n is not None and t.numel() != n:
                raise ValueError("%s must hold B*H*W = %d values" % (name, n))
        ws, nbytes = self.workspace(B, H, W)
        if mask is not None:
            self._check_mask(mask, B, H, W)
            fn = "dvsg_stabilize_ring_masked_u8" if pool.dtype == torch.uint8 else "dvsg_stabilize_ring_masked_f32"
            _lib.call(fn, self.handle, self._PRECISION_CODE[precision], ptr(pool), int(pool.shape[0]), ptr(table), ptr(mask),
                      B, H, W, ptr(out), ptr(F), ptr(xs), ptr(ys), ptr(ws), nbytes, stream())
            return
        fn = "dvsg_stabilize_ring_u8" if pool.dtype == torch.uint8 else "dvsg_stabilize_ring_f32"
        _lib.call(fn, self.handle, self._PRECISION_CODE[precision], ptr(pool), int(pool.shape[0]), ptr(table), B, H, W,
                  ptr(out), ptr(F), ptr(xs), ptr(ys), ptr(ws), nbytes, stream())

    def forward_masked(self, src, mask, table=None, precision="f32", stage=-1):
        """`dvsg_locnet_forward_masked`: F_t [B,25,2] (stage -1) or the parity tap of `stage` (0 conv1, 1 pool1) with
        the mask plane [B,H,W] multiplied into the history channels in conv1's load stage.  `src` is a window tensor
        [B,H,W,21] (table None) or a frame pool [n,H,W,3] float32 / uint8 with its index table [B,7]."""
        import torch
        if table is None:
            t = as_dev(src)
            B, H, W, C = t.shape
            if C != self.in_channels or C != 21:
                raise ValueError("a masked window has 21 channels, got %d (conv1: %d)" % (C, self.in_channels))
            kind, n_pool = 0, 0
        else:
            t = src
            B, H, W = self._check_ring(t, table)
            kind, n_pool = (2 if t.dtype == torch.uint8 else 1), int(t.shape[0])
        self._check_mask(mask, B, H, W)
        if stage > 1:
            raise ValueError("forward_masked taps stages 0 (conv1) and 1 (pool1); later stages do not depend on the source")
        ws, nbytes = self.workspace(B, H, W)
        dims = (ctypes.c_int * 3)()
        if stage < 0:
            buf = torch.empty((B, 25, 2), dtype=torch.float32, device=t.device)
        else:
            h1, w1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            buf = torch.empty((B * max(h1 * w1 * 64, 2048),), dtype=torch.float32, device=t.device)
        _lib.call("dvsg_locnet_forward_masked", self.handle, self._PRECISION_CODE[precision], ptr(t), kind, n_pool,
                  ptr(table), ptr(mask), B, H, W, int(stage), ptr(buf), buf.numel() * 4, dims, ptr(ws), nbytes, stream())
        if stage < 0:
            return buf
        h, w, c = dims[0], dims[1], dims[2]
        return buf[:B * h * w * c].reshape(B, h, w, c)

    def forward_ring(self, pool, table, precision="f32", stage=-1):
        """`dvsg_locnet_forward_ring`: F_t [B,25,2] (stage -1) or the parity tap of `stage` from a frame ring."""
        import torch
        B, H, W = self._check_ring(pool, table)
        ws, nbytes = self.workspace(B, H, W)
        dims = (ctypes.c_int * 3)()
        if stage < 0:
            buf = torch.empty((B, 25, 2), dtype=torch.float32, device=pool.device)
        else:
            h1, w1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            buf = torch.empty((B * max(h1 * w1 * 64, 2048),), dtype=torch.float32, device=pool.device)
            if stage > 1:
                raise ValueError("forward_ring taps stages 0 (conv1) and 1 (pool1); later stages do not depend on the source")
        _lib.call("dvsg_locnet_forward_ring", self.handle, self._PRECISION_CODE[precision], ptr(pool),
                  1 if pool.dtype == torch.uint8 else 0, int(pool.shape[0]), ptr(table), B, H, W, int(stage), ptr(buf),
                  buf.numel() * 4, dims, ptr(ws), nbytes, stream())
        if stage < 0:
            return buf
        h, w, c = dims[0], dims[1], dims[2]
        return buf[:B * h * w * c].reshape(B, h, w, c)

    def calibrate_f16(self, patches):
        """`dvsg_locnet_calibrate_f16`: measure the mean activation of every convolution input on the calibration windows
        `patches` [B,H,W,21] (a few frames of the clip about to be stabilised) and re-round the plain float16 weights of
        blocks 2-4 with error feedback, after which precision="f16" runs those blocks without the lo weight piece: 22 %
        less time, F_t within 1e-6 of the paired mode on frames of the calibration windows' statistics.  `None` undoes it."""
        if patches is None:
            ws, nbytes = self.workspace(1, 64, 64)
            _lib.call("dvsg_locnet_calibrate_f16", self.handle, 0, 0, 0, 0, ptr(ws), nbytes, stream())
            self.f16_calibrated = False
            return self
        t = as_dev(patches)
        B, H, W, C = t.shape
        if C != self.in_channels:
            raise ValueError("calibration windows must have %d channels, got %d" % (self.in_channels, C))
        ws, nbytes = self.workspace(B, H, W)
        _lib.call("dvsg_locnet_calibrate_f16", self.handle, ptr(t), B, H, W, ptr(ws), nbytes, stream())
        self.f16_calibrated = True
        return self

    def forward(self, patches, param_dim=25, precision="f32"):
        t = as_dev(patches)
        B, H, W, C = t.shape
        if C != self.in_channels:
            raise ValueError("localizationNet was loaded for %d input channels, got %d" % (self.in_channels, C))
        if param_dim != 25:
            raise ValueError("the reference's dense4 has 50 outputs: param_dim must be 25")
        ws, nbytes = self.workspace(B, H, W)
        F = empty((B, param_dim, 2), t)
        _lib.call(_entry("dvsg_locnet_forward", precision), self.handle, ptr(t), B, H, W, ptr(F), ptr(ws), nbytes,
                  stream())
        return F

    def tap(self, patches, stage, precision="f32"):
        """Parity hook: activation after `stage` (0 conv1, 1 pool1, 2..17 units, 18 pool5), float32."""
        t = as_dev(patches)
        B, H, W, C = t.shape
        ws, nbytes = self.workspace(B, H, W)
        h1, w1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        h, w = (h1 + 1) // 2, (w1 + 1) // 2
        cap = max(h1 * w1 * 64, h * w * 64)
        for base, units, last_stride in ((64, 3, 2), (128, 4, 2), (256, 6, 2), (512, 3, 1)):
            for u in range(1, units + 1):   # tiny frames: h*w*depth does not shrink monotonically
                s = last_stride if u == units else 1
                h, w = (h - 1) // s + 1, (w - 1) // s + 1
                cap = max(cap, h * w * base * 4)
        cap *= B
        buf = empty((cap,), t)
        dims = (ctypes.c_int * 3)()
        _lib.call(_entry("dvsg_locnet_forward_tap", precision), self.handle, ptr(t), B, H, W, int(stage), ptr(buf), cap * 4,
                  dims, ptr(ws), nbytes, stream())
        h, w, c = dims[0], dims[1], dims[2]
        return buf[:B * h * w * c].reshape(B, h, w, c)


def random_mask_plane(H_theta, h, w):
    """model.py:156-167 / eval_train.py:53-64: ProjectiveTransformer([h, w]).transform(ones, H_theta)[..., 0] for
    H_theta [B,8] (after the scale and identity offset of :162-163), float32 [B,h,w] on the device."""
    tt = as_dev(H_theta)
    if tt.dim() != 2 or tt.shape[1] != 8:
        raise ValueError("H must be [B,8], got %s" % (tuple(tt.shape),))
    B = int(tt.shape[0])
    out = empty((B, int(h), int(w)), tt)
    _lib.call("dvsg_random_mask_plane_f32", ptr(tt), B, int(h), int(w), ptr(out), stream())
    return out


def load_localizationNet(weights, scope="stabNet/localizationNet"):
    """Create (or replace) the network living under `scope` from checkpoint arrays keyed by
    the reference's variable names (ckpt_manager.py:33)."""
    _nets[scope] = LocNet(weights)
    return _nets[scope]


def localizationNet(input, param_dim, is_train=False, reuse=False, scope="stabNet/localizationNet"):
    """networks.py:30-46."""
    if is_train:
        raise NotImplementedError("training graph is out of scope; is_train must be False (model.py:99)")
    key = scope if isinstance(scope, str) else getattr(scope, "name", str(scope))
    if key not in _nets:
        raise _lib.DvsgError("no weights loaded for scope %r: call load_localizationNet(weights) first "
                             "(the reference would silently run on random weights, ckpt_manager.py:21-22)" % key)
    F = _nets[key].forward(input, param_dim)
    return like_input(F, input)
