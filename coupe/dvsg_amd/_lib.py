"""ctypes binding of libdvsg_amd.so (include/dvsg_amd.h).

There is no fallback of any kind: if the shared library is missing, or was built for
another ABI version, importing a compute entry point raises.  Tensor plumbing (device
memory, streams) is PyTorch-ROCm; every compute call goes through the C ABI with raw
device pointers.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DVSG_AMD_LIB: A/B another build of the same ABI from tools/ (diagnostic; the product path is the in-tree library)
LIB_PATH = os.environ.get("DVSG_AMD_LIB") or os.path.join(_HERE, "libdvsg_amd.so")
ABI_VERSION = 1

c_float_p = ctypes.POINTER(ctypes.c_float)
_vp = ctypes.c_void_p
_i = ctypes.c_int

# name -> argtypes; every function returns int status except the three string/version queries.
SIGNATURES = {
    "dvsg_tps_solve_f32": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "dvsg_tps_solve_checked_f32": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "dvsg_tps_warp_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "dvsg_flow_warp_f32": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "dvsg_stn_sample_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "dvsg_grid_affine_f32": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "dvsg_grid_projective_f32": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "dvsg_elastic_constants_f32": [_i, _vp, _vp],
    "dvsg_grid_elastic_f32": [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "dvsg_scale_rgb_f32": [_vp, _i, _i, _i, _i, _vp, _vp],
    "dvsg_locnet_create": [_i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(_vp),
                           ctypes.POINTER(_i), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(_vp)],
    "dvsg_locnet_destroy": [_vp],
    "dvsg_locnet_in_channels": [_vp],
    "dvsg_locnet_workspace_bytes": [_vp, _i, _i, _i, ctypes.POINTER(ctypes.c_size_t)],
    "dvsg_locnet_forward_f32": [_vp, _vp, _i, _i, _i, _vp, _vp, ctypes.c_size_t, _vp],
    "dvsg_locnet_forward_tap_f32": [_vp, _vp, _i, _i, _i, _i, _vp, ctypes.c_size_t,
                                    ctypes.POINTER(_i), _vp, ctypes.c_size_t, _vp],
    "dvsg_stabilize_f32": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp],
    "dvsg_locnet_forward_f32s": [_vp, _vp, _i, _i, _i, _vp, _vp, ctypes.c_size_t, _vp],
    "dvsg_locnet_forward_tap_f32s": [_vp, _vp, _i, _i, _i, _i, _vp, ctypes.c_size_t,
                                     ctypes.POINTER(_i), _vp, ctypes.c_size_t, _vp],
    "dvsg_stabilize_f32s": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp],
    "dvsg_conv_gemm_f32s": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, ctypes.c_size_t, _vp],
    "dvsg_locnet_forward_f32x3": [_vp, _vp, _i, _i, _i, _vp, _vp, ctypes.c_size_t, _vp],
    "dvsg_locnet_forward_tap_f32x3": [_vp, _vp, _i, _i, _i, _i, _vp, ctypes.c_size_t,
                                      ctypes.POINTER(_i), _vp, ctypes.c_size_t, _vp],
    "dvsg_stabilize_f32x3": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp],
    "dvsg_conv_gemm_f32x3": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, ctypes.c_size_t, _vp],
    "dvsg_pack_weights_f32x3": [_vp, _vp, _i, _i, _vp],
    "dvsg_conv3x3_1x1_f32x3": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dvsg_f32_to_pieces": [_vp, _vp, ctypes.c_size_t, _vp],
    "dvsg_pieces_to_f32": [_vp, _vp, ctypes.c_size_t, _vp],
    "dvsg_locnet_forward_f16": [_vp, _vp, _i, _i, _i, _vp, _vp, ctypes.c_size_t, _vp],
    "dvsg_locnet_forward_tap_f16": [_vp, _vp, _i, _i, _i, _i, _vp, ctypes.c_size_t,
                                    ctypes.POINTER(_i), _vp, ctypes.c_size_t, _vp],
    "dvsg_stabilize_f16": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp],
    "dvsg_conv_gemm_f16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, ctypes.c_size_t, _vp],
    "dvsg_conv_gemm_f16s": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, ctypes.c_size_t, _vp],
    "dvsg_conv_gemm_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, ctypes.c_size_t, _vp],
    "dvsg_conv3x3_1x1_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dvsg_frames_u8_to_f32": [_vp, ctypes.c_size_t, _i, _vp, _vp],
    "dvsg_frames_resize_u8_f32": [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp],
    "dvsg_window_gather_f32": [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp],
    "dvsg_stabilize_ring_f32": [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp],
    "dvsg_stabilize_ring_u8": [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp],
    "dvsg_locnet_forward_ring": [_vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp, ctypes.c_size_t, ctypes.POINTER(_i), _vp,
                                 ctypes.c_size_t, _vp],
    "dvsg_random_mask_plane_f32": [_vp, _i, _i, _i, _vp, _vp],
    "dvsg_stabilize_masked_f32": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp],
    "dvsg_stabilize_ring_masked_f32": [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp],
    "dvsg_stabilize_ring_masked_u8": [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp],
    "dvsg_locnet_forward_masked": [_vp, _i, _vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, ctypes.c_size_t, ctypes.POINTER(_i),
                                   _vp, ctypes.c_size_t, _vp],
    "dvsg_frames_f32_to_u8": [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp],
    "dvsg_frames_f64_to_u8": [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp],
    "dvsg_debug_set_option": [ctypes.c_char_p, _i],
    "dvsg_debug_calibrate_f16_weights": [_vp, _vp, _i, _i, _i, _i, _vp, ctypes.c_size_t, _vp],
    "dvsg_locnet_calibrate_f16": [_vp, _vp, _i, _i, _i, _vp, ctypes.c_size_t, _vp],
    "dvsg_markers_enabled": [],
    "dvsg_prof_begin": [_i],
    "dvsg_prof_end": [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i), ctypes.POINTER(ctypes.c_double),
                      ctypes.POINTER(ctypes.c_double)],
}
QUERIES = ("dvsg_abi_version", "dvsg_last_error_string", "dvsg_target_arch")

_lib = None


class DvsgError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library.  Raises if it is absent or mismatched."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise DvsgError(
            "libdvsg_amd.so not found at %s: build it with ./build.sh (hipcc --offload-arch=gfx950). "
            "coupe.dvsg_amd has no CPU or PyTorch fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.dvsg_abi_version.restype = ctypes.c_int
    lib.dvsg_last_error_string.restype = ctypes.c_char_p
    lib.dvsg_target_arch.restype = ctypes.c_char_p
    if lib.dvsg_abi_version() != ABI_VERSION:
        raise DvsgError("libdvsg_amd.so ABI %d != expected %d" % (lib.dvsg_abi_version(), ABI_VERSION))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    _lib = lib
    # diagnostic A/B switches for kernel experiments (see dvsg_debug_set_option in the header): they
    # change which kernels production calls select, so they only apply under an explicit DVSG_DEBUG=1
    for env, opt in (("DVSG_CONV_VARIANT", b"conv_variant"), ("DVSG_FUSE_CONV", b"fuse_conv"),
                     ("DVSG_F16_SPLIT", b"f16_split"), ("DVSG_FUSE_SHORTCUT", b"fuse_shortcut"),
                     ("DVSG_CONV1_VARIANT", b"conv1_variant"), ("DVSG_F16_PAIR_MASK", b"f16_pair_mask"),
                     ("DVSG_WIDE16_PACKED", b"wide16_packed"), ("DVSG_WIDE16_AROWS", b"wide16_arows"),
                     ("DVSG_WIDE16_HREUSE", b"wide16_hreuse"), ("DVSG_FUSED_HREUSE", b"fused_hreuse"),
                     ("DVSG_CONCAT_SC", b"concat_sc"), ("DVSG_FLOW_TILED", b"flow_tiled")):
        if os.environ.get(env) and os.environ.get("DVSG_DEBUG") == "1":
            check(lib.dvsg_debug_set_option(opt, int(os.environ[env])), "dvsg_debug_set_option")
    return lib


def check(status, what):
    if status != 0:
        msg = load().dvsg_last_error_string().decode("utf-8", "replace")
        raise DvsgError("%s failed (status %d): %s" % (what, status, msg))


def call(name, *args):
    check(getattr(load(), name)(*args), name)
