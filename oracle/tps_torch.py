"""Oracle (test infrastructure): ThinPlateSpline.py:4-170 on torch-CPU -- the same graph as
oracle/thin_plate_spline.py (the NumPy restatement of record), written with torch ops so that it runs on all host
cores: the warp leg of bench.py's `cpu_baseline` ("port").  TF-CPU executes the reference's graph with a
multi-threaded Eigen back end; a one-thread NumPy warp understated the CPU (round 3: 222 of 680 ms per 720p frame).
Checked against the NumPy restatement in tests/test_oracle_kat.py.  parity unpinned (see oracle/__init__.py).
"""
import numpy as np
import torch


@torch.no_grad()
def ThinPlateSpline(U, coord, vector, out_size):
    """Returns (output [B,h,w,C], x_s [B*h*w], y_s [B*h*w]) as NumPy float32, like ThinPlateSpline.py:170."""
    U = torch.as_tensor(np.asarray(U, dtype=np.float32))
    coord = torch.as_tensor(np.asarray(coord, dtype=np.float32))
    vector = torch.as_tensor(np.asarray(vector, dtype=np.float32))
    B, H, W, C = U.shape
    out_h, out_w = int(out_size[0]), int(out_size[1])
    P = coord.shape[1]
    # _solve_system (:143-166)
    p = torch.cat([torch.ones(B, P, 1), coord], 2)
    d2 = (p.reshape(B, P, 1, 3) - p.reshape(B, 1, P, 3)).square().sum(3)
    r = d2 * torch.log(d2 + 1e-6)
    W0 = torch.cat([p, r], 2)
    W1 = torch.cat([torch.zeros(B, 3, 3), p.transpose(1, 2)], 2)
    Winv = torch.linalg.inv(torch.cat([W0, W1], 1))
    tp = torch.cat([coord + vector, torch.zeros(B, 3, 2)], 1)
    T = torch.matmul(Winv, tp).transpose(1, 2)                                   # [B,2,P+3]
    # _meshgrid (:92-111): tf.linspace is start + step * i in float32
    x_t = (-1.0 + (2.0 / max(out_w - 1, 1)) * torch.arange(out_w, dtype=torch.float32)) if out_w > 1 else torch.tensor([-1.0])
    y_t = (-1.0 + (2.0 / max(out_h - 1, 1)) * torch.arange(out_h, dtype=torch.float32)) if out_h > 1 else torch.tensor([-1.0])
    x_t = x_t.repeat(out_h).reshape(1, 1, -1)
    y_t = y_t.repeat_interleave(out_w).reshape(1, 1, -1)
    dd = (x_t - coord[:, :, 0:1]).square() + (y_t - coord[:, :, 1:2]).square()    # [B,P,HW]
    rr = dd * torch.log(dd + 1e-6)
    grid = torch.cat([torch.ones(B, 1, out_h * out_w), x_t.expand(B, -1, -1), y_t.expand(B, -1, -1), rr], 1)
    Tg = torch.matmul(T, grid)                                                    # :129
    xs, ys = Tg[:, 0], Tg[:, 1]
    # _interpolate (:30-90), sampler A
    x = (xs + 1.0) * float(W) / 2.0
    y = (ys + 1.0) * float(H) / 2.0
    x0 = torch.floor(x).to(torch.int64)
    y0 = torch.floor(y).to(torch.int64)
    x1, y1 = x0 + 1, y0 + 1
    x0, x1 = x0.clamp(0, W - 1), x1.clamp(0, W - 1)
    y0, y1 = y0.clamp(0, H - 1), y1.clamp(0, H - 1)
    flat = U.reshape(B, H * W, C)

    def tap(yy, xx):
        return torch.gather(flat, 1, (yy * W + xx).unsqueeze(2).expand(-1, -1, C))
    x0f, x1f, y0f, y1f = x0.float(), x1.float(), y0.float(), y1.float()
    wa = ((x1f - x) * (y1f - y)).unsqueeze(2)
    wb = ((x1f - x) * (y - y0f)).unsqueeze(2)
    wc = ((x - x0f) * (y1f - y)).unsqueeze(2)
    wd = ((x - x0f) * (y - y0f)).unsqueeze(2)
    out = wa * tap(y0, x0) + wb * tap(y1, x0) + wc * tap(y0, x1) + wd * tap(y1, x1)
    return out.reshape(B, out_h, out_w, C).numpy(), xs.reshape(-1).numpy(), ys.reshape(-1).numpy()
