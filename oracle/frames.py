"""CPU oracle for the frame formats either side of the hot path (eval.py:76-124).
TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py for who may import this.

PARITY STATUS: parity unpinned.  `cv2` is a third-party dependency of the reference's driver
(eval.py:3; no version pinned anywhere in the tree) and is not installed here, so
`resize_linear` restates OpenCV's published INTER_LINEAR algorithm for float64 images
(imgproc/resize.cpp: `resizeGeneric_` with `HResizeLinear<double,double,float>` /
`VResizeLinear<double,double,float>`) and is pinned only by the known-answer tests in
tests/test_oracle_kat.py (identity size, constant image, exact 2x case, edge clamping).
"""
import numpy as np


def read_frame(frame_bgr_u8, out_w, out_h):
    """eval.py:76-81: cv2.cvtColor(BGR2RGB), then cv2.resize(frame / 255., (out_w, out_h))."""
    rgb = np.asarray(frame_bgr_u8)[..., ::-1]
    return resize_linear(rgb / 255., out_w, out_h)


def _taps(n_dst, n_src):
    """Per destination index: left tap, right tap, float32 weight of the right tap."""
    scale = np.float64(n_src) / np.float64(n_dst)
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    s[lo] = 0
    f[lo] = 0.0
    hi = s >= n_src - 1
    s[hi] = n_src - 1
    f[hi] = 0.0
    return s, np.minimum(s + 1, n_src - 1), f


def resize_linear(src, out_w, out_h):
    """cv2.resize(src, (out_w, out_h)) for a float64 [H,W,C] image, default INTER_LINEAR."""
    src = np.asarray(src, dtype=np.float64)
    h, w = src.shape[:2]
    if (h, w) == (out_h, out_w):
        return src.copy()
    x0, x1, fx = _taps(out_w, w)
    y0, y1, fy = _taps(out_h, h)
    a1 = fx.astype(np.float64)[None, :, None]
    a0 = (np.float32(1.0) - fx).astype(np.float64)[None, :, None]
    rows = src[:, x0] * a0 + src[:, x1] * a1                     # horizontal pass
    b1 = fy.astype(np.float64)[:, None, None]
    b0 = (np.float32(1.0) - fy).astype(np.float64)[:, None, None]
    return rows[y0] * b0 + rows[y1] * b1                         # vertical pass


def to_uint8(x):
    """eval.py:112 np.uint8(x * 255.) -- float64 product, truncation."""
    return np.uint8(np.asarray(x, dtype=np.float64) * 255.)


def window_index_trace(n_frames, skip_length=(0, 16, 24, 28, 30, 31, 32)):
    """Which frame each window slot holds at each step of eval.py:93-124, found by running the
    reference's list manipulation on labels: ('u', i) = unstable input frame i, ('s', i) =
    stabilised frame i.  Returns a list (per step) of lists (per window slot)."""
    skip = np.array(skip_length)
    span = int(skip[-1] - skip[0])
    total = [('u', i) for i in range(n_frames)]
    for _ in range(span):
        total.insert(0, total[0])                                 # :93-94
    idx = skip.copy()
    trace = []
    for frame_idx in range(span, len(total)):                     # :101
        trace.append([total[j] for j in idx])                     # :103
        s = ('s', frame_idx - span)
        total[idx[-1]] = s                                        # :116
        if frame_idx == span:                                     # :118-120
            for i in range(span):
                total[i] = s
        idx = idx + 1                                             # :124
    return trace


def teacher_forced_trace(n_frames, skip_length=(0, 16, 24, 28, 30, 31, 32)):
    """Which frame each window slot holds at each step of eval_train.py:137-165, by running the
    reference's array manipulation on labels: ('u', i) unstable frame i, ('t', i) stable frame i."""
    skip = np.array(skip_length)
    span = int(skip[-1] - skip[0])
    unstab = [('u', i) for i in range(n_frames)]
    stab = [('t', i) for i in range(n_frames)]
    for i in range(span):                                          # :137-138
        unstab[i] = stab[i]
    idx = skip.copy()
    trace = []
    for frame_idx in range(span, n_frames):                        # :146
        trace.append([unstab[j] for j in idx])                     # :148
        unstab[idx[-1]] = stab[idx[-1]]                            # :162
        idx = idx + 1                                              # :165
    return trace
