"""Tensor plumbing shared by the facade modules: PyTorch-ROCm owns device memory and
streams; the compute goes through the C ABI (``_lib``)."""
import numpy as np
import torch

from . import _lib


def device():
    if not torch.cuda.is_available():
        raise _lib.DvsgError("coupe.dvsg_amd needs a HIP device (MI355X / gfx950); none is visible "
                             "and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def as_dev(x, name="tensor"):
    """float32, contiguous, on the current HIP device.  NumPy / list inputs are copied in
    (the reference feeds NumPy arrays through sess.run, eval.py:106-110)."""
    dev = device()
    if isinstance(x, torch.Tensor):
        t = x
        if t.device.type != "cuda":
            t = t.to(dev)
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.float32)).to(dev)
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t.contiguous()


def is_host(x):
    return not isinstance(x, torch.Tensor)


def like_input(t, ref):
    """Return `t` as NumPy when the caller passed NumPy (sess.run semantics), else as is."""
    return t.cpu().numpy() if is_host(ref) else t


def ptr(t):
    return 0 if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def empty(shape, like=None):
    return torch.empty(shape, dtype=torch.float32, device=like.device if like is not None else device())
