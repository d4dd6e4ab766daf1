"""Seeded synthetic inputs shared by the parity tests, the golden-vector script and bench.py
(SURVEY.md section 8d).  Pure NumPy/SciPy: no oracle, no product imports."""
import numpy as np
from scipy import ndimage


def smooth_frames(seed, B, H, W, C=3, factor=8):
    """Band-limited frames in [0,1]: uniform noise at 1/factor resolution, cubic upsampling,
    clipped.  |grad I| stays ~0.1 per pixel, so a source-coordinate error of e px moves a
    warped pixel by ~0.1 e."""
    rng = np.random.default_rng(seed)
    h, w = -(-H // factor) + 3, -(-W // factor) + 3
    out = np.empty((B, H, W, C), dtype=np.float32)
    for b in range(B):
        for c in range(C):
            lo = rng.uniform(0.0, 1.0, (h, w))
            up = ndimage.zoom(lo, factor, order=3)[factor:factor + H, factor:factor + W]
            out[b, :, :, c] = np.clip(up, 0.0, 1.0)
    return out


def window_frames(seed, B, H, W, S=7):
    """[B,H,W,3S] windows: S views of one smooth scene with small integer translations
    (oldest -> newest on the channel axis, RGB inside each frame; eval.py:104)."""
    rng = np.random.default_rng(seed + 7919)
    base = smooth_frames(seed, B, H + 16, W + 16, 3)
    out = np.empty((B, H, W, 3 * S), dtype=np.float32)
    for b in range(B):
        for s in range(S):
            dy, dx = rng.integers(0, 17, 2)
            out[b, :, :, 3 * s:3 * s + 3] = base[b, dy:dy + H, dx:dx + W, :]
    return out


def smooth_flow(seed, B, H, W, sigma_px=4.0, oob_frac=0.01):
    """Flow ~ N(0, sigma) box-smoothed, plus a fraction of pixels pushed far out of bounds."""
    rng = np.random.default_rng(seed)
    f = rng.normal(0.0, sigma_px * 6.0, (B, H, W, 2))
    k = min(15, max(3, (min(H, W) // 4) | 1))
    f = ndimage.uniform_filter(f, size=(1, k, k, 1), mode="nearest")
    m = rng.uniform(size=(B, H, W)) < oob_frac
    f[m] += rng.choice([-1.0, 1.0], size=(int(m.sum()), 2)) * (max(H, W) + 5.0)
    return f.astype(np.float32)


def control_vectors(seed, B, P=25, scale=0.05):
    rng = np.random.default_rng(seed)
    return (scale * rng.standard_normal((B, P, 2))).astype(np.float32)


def v_src(B):
    lin = np.linspace(-1.0, 1.0, 5)
    pts = np.array([[x, y] for y in lin for x in lin], dtype=np.float32)
    return np.tile(pts[None], (B, 1, 1))


def mask_homographies(seed, n):
    """[n,8] homographies of eval_train.py:55-57 / model.py:161-163: uniform[-1,1) * scale + identity, float32."""
    u = np.random.default_rng(seed).uniform(-1.0, 1.0, (n, 8)).astype(np.float32)
    scale = np.array([0.1, 0.1, 0.5, 0.1, 0.1, 0.5, 0.1, 0.1], dtype=np.float32)
    ident = np.array([1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], dtype=np.float32)
    return ((u * scale).astype(np.float32) + ident).astype(np.float32)


def stable_unstable_clips(seed, N, H, W):
    """A (stable, unstable) pair of clips [N,H,W,3] for eval_train.py's teacher-forced loop."""
    stab = smooth_frames(seed, N, H, W)
    unstab = (np.roll(stab, 2, axis=2) * 0.9 + 0.05).astype(np.float32)
    return stab, unstab
