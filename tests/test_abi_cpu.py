"""No-GPU checks of the C-ABI boundary: the shared library loads, exports every symbol that
include/dvsg_amd.h declares, rejects bad arguments with a status code (never aborts), and its
host-only entry point agrees with the oracle.  No kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dvsg_amd.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dvsg_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from coupe.dvsg_amd import _lib
    return _lib.load()


def test_header_declares_the_survey_export_set():
    syms = declared_symbols()
    for s in ["dvsg_tps_solve_f32", "dvsg_tps_warp_f32", "dvsg_flow_warp_f32", "dvsg_stn_sample_f32",
              "dvsg_grid_projective_f32", "dvsg_grid_affine_f32", "dvsg_grid_elastic_f32", "dvsg_locnet_create",
              "dvsg_locnet_destroy", "dvsg_locnet_forward_f32", "dvsg_stabilize_f32", "dvsg_last_error_string"]:
        assert s in syms


def test_library_exports_every_declared_symbol(lib):
    for s in declared_symbols():
        assert hasattr(lib, s), "libdvsg_amd.so does not export %s" % s
    assert lib.dvsg_abi_version() == 1
    assert lib.dvsg_target_arch() == b"gfx950"


def test_python_binding_covers_every_declared_symbol():
    from coupe.dvsg_amd import _lib
    bound = set(_lib.SIGNATURES) | set(_lib.QUERIES)
    assert set(declared_symbols()) == bound


def test_bad_arguments_return_status_not_abort(lib):
    from coupe.dvsg_amd import DvsgError, _lib
    assert lib.dvsg_tps_solve_f32(None, None, 1, 1, 25, None, None) == -1
    assert b"NULL" in lib.dvsg_last_error_string()
    assert lib.dvsg_tps_solve_f32(8, 8, 1, 1, 99, 8, None) == -1
    assert b"P=99" in lib.dvsg_last_error_string()
    assert lib.dvsg_flow_warp_f32(8, 8, 0, 4, 4, 3, 8, None) == -1
    assert lib.dvsg_scale_rgb_f32(8, 1, 4, 4, 4, 8, None) == -1          # C not a multiple of 3
    assert lib.dvsg_grid_elastic_f32(8, 8, 8, 100, None, 1, 4, 4, 3, 4, 4, None, 8, 8, None) == -1
    assert lib.dvsg_prof_end(None, None, None, None) == -1               # not armed
    with pytest.raises(DvsgError, match="dvsg_locnet_workspace_bytes"):
        _lib.call("dvsg_locnet_workspace_bytes", None, 1, 8, 8, None)


def test_locnet_create_rejects_incomplete_checkpoint(lib):
    """Runs before any device work: a missing array is DVSG_ERR_WEIGHTS (-4)."""
    name = (ctypes.c_char_p * 1)(b"stabNet/localizationNet/df/dense1/b:0")
    arr = np.zeros(2048, np.float32)
    data = (ctypes.c_void_p * 1)(arr.ctypes.data)
    nd = (ctypes.c_int * 1)(1)
    dims = (ctypes.c_int64 * 4)(2048, 1, 1, 1)
    handle = ctypes.c_void_p()
    rc = lib.dvsg_locnet_create(1, name, data, nd, dims, ctypes.byref(handle))
    assert rc == -4 and b"conv1/weights" in lib.dvsg_last_error_string()
    assert not handle.value


@pytest.mark.parametrize("g", [4, 3, 5])
def test_elastic_constants_match_oracle(lib, g):
    from oracle.spatial_transformer import ElasticTransformer
    n = g * g
    src = np.empty((2, n), np.float32)
    linv = np.empty((n, n + 3), np.float32)
    assert lib.dvsg_elastic_constants_f32(g, src.ctypes.data, linv.ctypes.data) == 0
    o = ElasticTransformer((8, 8), param_dim=2 * n, param_dim_per_side=g)
    assert np.array_equal(src, o.source_points)
    assert np.abs(linv - o.L_inv).max() < 5e-4     # float64 vs float32 (LAPACK) inverse


def test_no_cpu_fallback_without_device():
    """The product must fail loudly, not compute on the CPU, when no HIP device is visible."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from coupe.dvsg_amd import DvsgError
    from coupe.dvsg_amd.warp_with_optical_flow import tf_warp
    with pytest.raises(DvsgError, match="no CPU fallback"):
        tf_warp(np.zeros((1, 4, 4, 3), np.float32), np.zeros((1, 4, 4, 2), np.float32), 4, 4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "coupe")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f


def _build_c_demo(tmpdir):
    """examples/c_abi_demo.c with gcc as C99: the header is C (not only C++) and every entry point the demo uses links."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("gcc / ROCm headers not available")
    exe = os.path.join(str(tmpdir), "c_abi_demo")
    libdir = os.path.join(ROOT, "coupe", "dvsg_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L" + libdir, "-ldvsg_amd",
           "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    return exe


def test_header_is_c99_and_the_c_demo_links(tmp_path):
    _build_c_demo(tmp_path)


@pytest.mark.gpu
def test_c_demo_runs_the_known_answer_tests(tmp_path):
    import subprocess
    exe = _build_c_demo(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "all checks passed" in out.stdout


def test_roctx_ranges_are_off_by_default_and_load_on_request():
    """DVSG_ROCTX=1 makes every stage / bottleneck unit open a roctx range; the roctx
    library is found with dlopen at run time and the switch is read once per process."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from coupe.dvsg_amd import _lib; "
            "print(_lib.load().dvsg_markers_enabled())" % ROOT)
    env = {k: v for k, v in os.environ.items() if k != "DVSG_ROCTX"}
    off = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True)
    on = subprocess.run([sys.executable, "-c", code], env=dict(env, DVSG_ROCTX="1"), capture_output=True, text=True, check=True)
    assert off.stdout.strip() == "0" and on.stdout.strip() == "1", (off.stdout, on.stdout, on.stderr)
