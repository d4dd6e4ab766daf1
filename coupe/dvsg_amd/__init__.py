"""coupe.dvsg_amd -- MI355X (gfx950) implementation of the coupe.DVSG per-frame inference
hot path: localizationNet CNN -> thin-plate-spline grid -> bilinear resampler, plus the
optical-flow warp and the affine / projective / elastic spatial transformers.

The modules mirror the reference's file names and call surfaces (model.py, networks.py,
ThinPlateSpline.py, ThinPlateSpline2.py, warp_with_optical_flow.py, spatial_transformer.py, and the
evaluation graph of eval_train.py);
the compute lives in libdvsg_amd.so (hand-written HIP, C ABI in include/dvsg_amd.h).
Importing this package does not need a GPU; calling any operator does, and fails loudly
without the library or the device.
"""
from . import _lib  # noqa: F401
from ._lib import DvsgError  # noqa: F401

__all__ = ["DvsgError", "model", "networks", "ThinPlateSpline", "ThinPlateSpline2",
           "warp_with_optical_flow", "spatial_transformer", "weights", "clip", "eval_train"]
