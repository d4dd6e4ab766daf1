"""Drop-in for the reference's ThinPlateSpline.py / ThinPlateSpline2.py on MI355X.

Same call surface as ThinPlateSpline.py:4 (`ThinPlateSpline(U, coord, vector, out_size)
-> (output, x_s_flat, y_s_flat)`) and ThinPlateSpline2.py:4 (`ThinPlateSpline2(U, source,
target, out_size)`), taking / returning torch tensors on the HIP device (NumPy in ->
NumPy out).  Compute: `dvsg_tps_solve_f32` + `dvsg_tps_warp_f32` (include/dvsg_amd.h).
"""
import torch

from . import _lib
from ._tensor import as_dev, empty, like_input, ptr, stream


def _solve(ct, rt, rhs_is_vector, B, P, T):
    """`_solve_system` with tf.matrix_inverse's error behaviour (ThinPlateSpline.py:159): repeated or
    collinear control points raise instead of warping with a garbage map.  Reading the 4-byte
    flag back synchronises, as a sess.run of these ops would; the fused evaluation path
    (`dvsg_stabilize_*`) inverts its constant V_src system once at load time and never comes here."""
    flag = torch.zeros(1, dtype=torch.int32, device=ct.device)
    _lib.call("dvsg_tps_solve_checked_f32", ptr(ct), ptr(rt), int(rhs_is_vector), B, P, ptr(T), ptr(flag), stream())
    bad = int(flag.item())
    if bad:
        raise _lib.DvsgError("ThinPlateSpline: the control-point system of %d of %d samples is not invertible "
                             "(repeated or collinear control points)" % (bad, B))


def _tps(U, coord, rhs, out_size, rhs_is_vector):
    Ut = as_dev(U, "U")
    ct = as_dev(coord, "coord")
    rt = as_dev(rhs, "vector/target")
    if Ut.dim() != 4:
        raise ValueError("U must be [num_batch, height, width, num_channels]")
    B, H, W, C = Ut.shape
    if ct.dim() != 3 or ct.shape[0] != B or ct.shape[2] != 2 or rt.shape != ct.shape:
        raise ValueError("coord / vector must be [num_batch, num_point, 2] with the batch of U")
    P = ct.shape[1]
    out_h, out_w = int(out_size[0]), int(out_size[1])
    T = empty((B, 2, P + 3), Ut)
    out = empty((B, out_h, out_w, C), Ut)
    xs = empty((B * out_h * out_w,), Ut)
    ys = empty((B * out_h * out_w,), Ut)
    s = stream()
    _solve(ct, rt, rhs_is_vector, B, P, T)
    _lib.call("dvsg_tps_warp_f32", ptr(Ut), ptr(ct), ptr(T), B, H, W, C, P, out_h, out_w,
              ptr(out), ptr(xs), ptr(ys), s)
    return like_input(out, U), like_input(xs, U), like_input(ys, U)


def ThinPlateSpline(U, coord, vector, out_size):
    """ThinPlateSpline.py:4-170.  U [B,H,W,C]; coord, vector [B,P,2]; out_size (h, w)."""
    return _tps(U, coord, vector, out_size, True)


def ThinPlateSpline2(U, source, target, out_size):
    """ThinPlateSpline2.py:4-170: as above with the right-hand side `target` (line 160)."""
    return _tps(U, source, target, out_size, False)


def solve_system(coord, rhs, rhs_is_vector=True):
    """`_solve_system` alone (ThinPlateSpline.py:143-166): returns T [B,2,P+3]."""
    ct = as_dev(coord)
    rt = as_dev(rhs)
    B, P, _ = ct.shape
    T = empty((B, 2, P + 3), ct)
    _solve(ct, rt, rhs_is_vector, B, P, T)
    return like_input(T, coord)
