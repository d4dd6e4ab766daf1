#!/usr/bin/env python3
"""Benchmark of the DVSG hot path on MI355X: stabilised 1280x720 frames per second.

One "step" = one pass of the evaluation graph of model.py:98-123 (localizationNet CNN ->
TPS solve -> TPS grid + sampler A) over one batch of 16 synthetic 7-frame windows
(BASELINE.json configs[1]) through `dvsg_stabilize_f32`, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work: started bare with --gpus N > 1 (no WORLD_SIZE in the environment) this script
spawns the N ranks itself -- before anything in the parent touches the GPU -- relays rank 0's JSON
line and exits non-zero if any rank does.

Multi-GPU (N > 1) is BASELINE.json configs[3]: every rank owns a contiguous shard of 64 independent
720p windows (512 at N = 8), one step stabilises the shard in four `dvsg_stabilize_f32`
calls of 16 windows -- the very call configs[1] times at N = 1, so per-GPU work per call is the same
at every N (weak scaling) -- with no data-path collective, and the stabilised frames of every step
are gathered to rank 0 over RCCL (one `gather` of [64,720,1280,3] per rank and step, issued
asynchronously so that it overlaps the next step's kernels; the last one is waited for inside the
timed region).

Rank 0 prints ONE JSON line.  `roofline` is measured live with hipEvents around every launch
of the dominant kernel class inside the timed region; `cpu_baseline` times the CPU oracle
(torch-CPU CNN + NumPy TPS restatement, the "port") on the host cores at N=1.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 chip peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense f16 / bf16 MFMA peak
PEAK_HBM_GBS = 8000.0
MFMA_CLASSES = (0, 1, 2, 8)
KERNEL_CLASSES = {0: "conv1_kernel", 1: "conv_gemm_kernel<*,3,*>", 2: "conv_gemm_kernel<*,1,*>",
                  3: "maxpool_kernel", 4: "head", 5: "tps_solve_kernel", 6: "tps_warp_kernel", 7: "flow_warp_strip_kernel / stn_kernel",
                  8: "conv3x3_1x1_kernel"}
# the float16 mode's kernels of the same classes (big launches)
KERNEL_CLASSES_F16 = {0: "conv1_f16_march_kernel", 1: "conv_wide16h_kernel / conv_wide16a_kernel<3,*>", 2: "conv_wide16a_kernel<1,*> / conv_wide16_kernel<1,*>",
                      3: "maxpool_h8_kernel", 8: "conv3x3_1x1_f16h_kernel"}


def gpu_windows(B, H, W, seed, dev, S=7):
    """Band-limited synthetic windows [B,H,W,3S] in [0,1] generated on the device (same recipe
    as tests/inputs.py: uniform noise at 1/8 resolution, bicubic upsampling, small shifts)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    lo = torch.rand((B, 3, (H + 16) // 8 + 3, (W + 16) // 8 + 3), generator=g, device=dev)
    up = F.interpolate(lo, scale_factor=8, mode="bicubic", align_corners=False)[:, :, 8:8 + H + 16, 8:8 + W + 16]
    up = up.clamp_(0.0, 1.0)
    shifts = torch.randint(0, 17, (S, 2), generator=g, device=dev).tolist()
    views = [up[:, :, dy:dy + H, dx:dx + W] for dy, dx in shifts]
    x = torch.cat(views, dim=1).permute(0, 2, 3, 1).contiguous()
    return x


def pmc_traffic(cls, B, H, W, precision="f32", kind="stabilize", calibrated=False):
    """HBM-side bytes per launch of kernel class `cls` from the committed rocprofv3 PMC passes
    (profiles/rNN*_traffic.json, written by tools/summarize_profiles.py from separate FETCH_SIZE /
    WRITE_SIZE runs of this same command with the gfx950 corrections of MI355X_MICROARCH.md).
    Counters cannot be read from inside the timed run, so this is the last profiled value of the SAME
    workload (batch, size, precision recorded in the file); None for any other shape."""
    import glob
    want = {"batch": B, "height": H, "width": W, "precision": precision, "kind": kind, "calibrated": bool(calibrated)}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            have = dict(d.get("workload", {"batch": 16, "height": 720, "width": 1280, "precision": "f32"}))
            have.setdefault("kind", "stabilize")
            have.setdefault("calibrated", False)
            if have != want:
                continue
            c = d["classes"].get(str(cls))
            if c:
                return c["bytes_per_launch"], os.path.relpath(path, ROOT)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def usable_cores():
    """Host cores this process may actually use: min(affinity, cgroup quota), and never more
    than the 16-core share a one-GPU box grants (oversubscribing torch-CPU is slower)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, int(os.environ.get("DVSG_BENCH_CPU_THREADS", "16")))


def cpu_model():
    """CPU model string of the box (SURVEY.md 8d asks for core count AND model)."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine() or "unknown"


def cpu_baseline(weights, H, W, budget_s=20.0):
    """CPU oracle ("port": torch-CPU CNN + torch-CPU TPS warp, both on every usable core -- TF-CPU's Eigen back end is
    multi-threaded too; round 3's one-thread NumPy warp understated the CPU) on one 720p window at a time.  The CNN and
    the warp are also timed apart; tools/cpu_baseline_full.py adds the B=16 pass, too long for a default bench run
    (profiles/r04_cpu_baseline.json)."""
    from oracle.cnn_torch import TorchLocNet
    from oracle.tps_torch import ThinPlateSpline as o_tps
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import inputs
    cores = usable_cores()
    torch.set_num_threads(cores)
    net = TorchLocNet(weights)
    x = inputs.window_frames(1234, 1, H, W)
    vsrc = inputs.v_src(1)
    times, cnn_t, warp_t = [], [], []
    t_start = time.perf_counter()
    runs = 0
    while True:
        t0 = time.perf_counter()
        Ft = net.forward(x)
        t1 = time.perf_counter()
        o_tps(x[..., 18:], vsrc, Ft, (H, W))
        t2 = time.perf_counter()
        runs += 1
        if runs > 1:           # first run is warm-up
            times.append(t2 - t0)
            cnn_t.append(t1 - t0)
            warp_t.append(t2 - t1)
        if (time.perf_counter() - t_start > budget_s and len(times) >= 2) or len(times) >= 5:
            break
    med = float(np.median(times))
    return {"value": 1.0 / med, "unit": "frames/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "cnn_ms_per_frame": 1e3 * float(np.median(cnn_t)), "warp_ms_per_frame": 1e3 * float(np.median(warp_t)),
            "sample": "%d single-window (B=1, %dx%d) passes of the torch-CPU CNN + torch-CPU TPS warp oracle (%d threads "
                      "each) after 1 warm-up; median" % (len(times), W, H, cores)}


def cpu_baseline_flow(H, W, budget_s=10.0):
    """CPU oracle of tf_warp (NumPy, one thread) on one frame at a time."""
    from oracle.warp_with_optical_flow import tf_warp as o_tf_warp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import inputs
    im = inputs.smooth_frames(1234, 1, H, W)
    flow = inputs.smooth_flow(4321, 1, H, W)
    times = []
    t_start = time.perf_counter()
    while len(times) < 5 and (time.perf_counter() - t_start < budget_s or len(times) < 2):
        t0 = time.perf_counter()
        o_tf_warp(im, flow, H, W)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times[1:] if len(times) > 2 else times))
    return {"value": 1.0 / med, "unit": "frames/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
            "sample": "%d single-frame (%dx%d) passes of the NumPy tf_warp oracle; median" % (len(times), W, H)}


CNN_GFLOP_PER_FRAME = {(720, 1280): 154.5, (288, 512): 24.6}


def latency_mode(net, dev, sizes=((720, 1280), (288, 512)), n_frames=48, precision="f32"):
    """The reference's actual operating point (eval.py:93-124, config.py:12-13,123): ONE clip, batch 1, every frame's
    window reading the stabilised frames before it -- a lag-1 recurrence, nothing to batch.  `clip.stabilize_clip` on a
    synthetic float32 clip resident in HBM (one dvsg_stabilize_ring_f32 call per frame), steady state after a 4-frame
    warm-up clip; 720p and the reference's own 512x288.  Outside the timed region of the line of record."""
    from coupe.dvsg_amd.clip import stabilize_clip
    from coupe.dvsg_amd.model import StabNet
    out = {"mode": "eval.py clip loop: batch 1, exact autoregressive history, clip resident in HBM, %s"
                   % ("float32" if precision == "f32" else precision),
           "frames_per_clip": n_frames}
    for H, W in sizes:
        model = StabNet(H, W)
        model.locnet = net
        model.precision = precision
        frames = gpu_windows(n_frames, H, W, 4321, dev, S=1)
        stabilize_clip(model, None, frames[:4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stab = stabilize_clip(model, None, frames)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_frames
        assert stab.shape == frames.shape
        out["%dx%d" % (W, H)] = {"ms_per_frame": 1e3 * dt, "frames_per_s": 1.0 / dt,
                                 "achieved_tflops": CNN_GFLOP_PER_FRAME[(H, W)] / dt / 1e3,
                                 "frac_of_f32_mfma_peak": CNN_GFLOP_PER_FRAME[(H, W)] / dt / 1e3 / PEAK_F32_MFMA_TFLOPS}
        del frames, stab
    return out


def roofline_object(cls, precision, prof, CB, H, W, kind="stabilize", bound=None, calibrated=False):
    """The `roofline` object of one kernel class from dvsg_prof_end's figures (summed hipEvent durations of its launches,
    their count, algorithmic FLOPs and bytes) and the committed PMC traffic of the same workload."""
    total_ms, launches, flops, nbytes = prof
    if bound is None:
        bound = "mfma" if cls in MFMA_CLASSES else "hbm"
    if bound == "mfma":
        achieved = flops / (total_ms * 1e-3) / 1e12 if total_ms > 0 else 0.0
        # f32s: three float16 MFMAs per float32-equivalent product; f16: two (hi / lo weights), priced on algorithmic FLOPs
        # f32x3: six bfloat16 MFMAs per float32 product (the bf16 dense peak equals the f16 one)
        peak = (PEAK_F32_MFMA_TFLOPS if precision == "f32" else
                PEAK_F16_MFMA_TFLOPS / 3.0 if precision == "f32s" else
                PEAK_F16_MFMA_TFLOPS / 6.0 if precision == "f32x3" else PEAK_F16_MFMA_TFLOPS)
        roofline = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak}
    else:
        achieved = nbytes / (total_ms * 1e-3) / 1e9 if total_ms > 0 else 0.0
        roofline = {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": achieved / PEAK_HBM_GBS}
    traffic, traffic_src = pmc_traffic(cls, CB, H, W, precision, kind, calibrated)
    roofline.update({"traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                     "kernel": (KERNEL_CLASSES_F16.get(cls, KERNEL_CLASSES[cls]) if precision == "f16" else KERNEL_CLASSES[cls]),
                     "launches": launches, "avg_launch_ms": total_ms / max(launches, 1),
                     "algorithmic_per_launch": (flops if bound == "mfma" else nbytes) / max(launches, 1),
                     "algorithmic_bytes_per_launch": nbytes / max(launches, 1)})
    return roofline


def prof_region(cls, fn, steps):
    """Time `steps` calls of fn() (fenced) with the library's hipEvent hook armed for kernel class `cls`."""
    from coupe.dvsg_amd import _lib
    torch.cuda.synchronize()
    _lib.call("dvsg_prof_begin", cls)
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, n, fl, by = ctypes.c_double(), ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
    _lib.call("dvsg_prof_end", ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by))
    return dt, (ms.value, n.value, fl.value, by.value)


def free_gib(dev):
    free, _ = torch.cuda.mem_get_info(dev)
    return free / 2.0 ** 30


def make_flow_inputs(B, H, W, seed, dev):
    """SURVEY.md 8d cfg 3: frames as cfg 2; flow ~ N(0, 4 px) smoothed by a 15-px box, 1 % of the pixels out of bounds."""
    u_t = gpu_windows(B, H, W, 1234 + seed, dev, S=1)
    g = torch.Generator(device=dev).manual_seed(4321 + seed)
    flow = 4.0 * 15.0 * torch.randn((B, 2, H, W), generator=g, device=dev)
    flow = F.avg_pool2d(flow, 15, stride=1, padding=7).permute(0, 2, 3, 1).contiguous()
    oob = torch.rand((B, H, W, 1), generator=g, device=dev) < 0.01
    flow = torch.where(oob, flow + (max(H, W) + 5.0), flow).contiguous()
    return u_t, flow


def cfg2_tf_warp(dev, steps, warmup):
    """BASELINE.json configs[2] beside the line of record: B=64 1280x720 frames through `dvsg_flow_warp_f32`
    (warp_with_optical_flow.py:96-176), cfg-3 flow, inputs resident in HBM.  Outside the timed region of the headline."""
    from coupe.dvsg_amd import _lib
    B, H, W = 64, 720, 1280
    need = 4.0 * B * H * W * (3 + 2 + 3 + 3) * 3 / 2.0 ** 30      # frames, flow, two outputs; x3 for the generator's temporaries
    if free_gib(dev) < need:
        return {"skipped": "%.1f GiB free, %.1f needed" % (free_gib(dev), need)}
    u_t, flow = make_flow_inputs(B, H, W, 0, dev)
    outs = [torch.empty((B, H, W, 3), device=dev) for _ in range(2)]
    st = torch.cuda.current_stream().cuda_stream

    def run(i):
        _lib.call("dvsg_flow_warp_f32", u_t.data_ptr(), flow.data_ptr(), B, H, W, 3, outs[i & 1].data_ptr(), st)
    for i in range(max(warmup, 2)):
        run(i)
    steps = max(steps, 20)
    dt, prof = prof_region(7, run, steps)
    return {"workload": "configs[2]: batch=64 1280x720 frames, optical-flow warp (warp_with_optical_flow.tf_warp), flow ~ N(0, 4 px) "
                        "box-smoothed (15 px), 1 % of the pixels out of bounds",
            "metric": "tf_warp frames/sec (1280x720 RGB)", "value": B * steps / dt, "unit": "frames/s", "dtype": "f32",
            "steps": steps, "ms_per_step": 1e3 * dt / steps,
            "roofline": roofline_object(7, "f32", prof, B, H, W, kind="tf_warp")}


def cfg4_f16_4k(net, dev, steps, warmup):
    """BASELINE.json configs[4] beside the line of record: B=32 3840x2160 windows, float16 mode (float16 activations,
    hi / lo float16 weight pairs, float32 accumulation, float32 TPS / warp), ONE dvsg_stabilize_f16 call per step.
    `roofline` is the class with the most time in this mode -- the 1x1 convolutions, HBM-bound."""
    B, H, W = 32, 2160, 3840
    ws_bytes = ctypes.c_size_t()
    from coupe.dvsg_amd import _lib
    _lib.call("dvsg_locnet_workspace_bytes", net.handle, B, H, W, ctypes.byref(ws_bytes))
    inputs_gib = 4.0 * B * H * W * (21 + 3 + 3 + 3) / 2.0 ** 30
    need = ws_bytes.value / 2.0 ** 30 + inputs_gib + 4.0 * 8 * H * W * 21 * 2 / 2.0 ** 30 + 4.0   # + one chunk of generator temporaries
    if free_gib(dev) < need:
        return {"skipped": "%.1f GiB free, %.1f needed" % (free_gib(dev), need)}
    patches = torch.cat([gpu_windows(8, H, W, 400 + i, dev) for i in range(B // 8)], 0)
    u_t = patches[..., 18:].contiguous()
    outs = [torch.empty((B, H, W, 3), device=dev) for _ in range(2)]
    F_t = torch.empty((B, 25, 2), device=dev)

    def run(i):
        net.stabilize(patches, u_t, outs[i & 1], F_t, precision="f16")
    steps = max(2, min(steps, 4))
    # (a) hi / lo weight pairs in every layer: what precision = "f16" does with nothing but a checkpoint
    for i in range(max(1, min(warmup, 2))):
        run(i)
    dt_pairs, prof_pairs = prof_region(2, run, steps)
    # (b) calibrated (dvsg_locnet_calibrate_f16): the mean activation of every convolution input measured on ONE other
    # 3840x2160 window, the plain float16 weights of blocks 2-4 re-rounded with error feedback against it; those blocks
    # then run without the lo weight piece.  Same parity bounds (tests/test_gpu_configs.py::test_cfg4_b32_4k_f16[True]).
    calib = gpu_windows(1, H, W, 499, dev)
    net.calibrate_f16(calib)
    del calib
    try:
        for i in range(max(1, min(warmup, 2))):
            run(i)
        dt, prof = prof_region(2, run, steps)
    finally:
        net.calibrate_f16(None)
    res = {"workload": "configs[4]: batch=32 3840x2160 7-frame windows, float16 mode (float16 activations and float16 MFMA convs, "
                       "float32 accumulation, float32 TPS / warp), full CNN+TPS+bilinear warp, one call per step",
           "metric": "stabilized frames/sec (3840x2160 RGB)", "value": B * steps / dt, "unit": "frames/s", "dtype": "f16",
           "steps": steps, "ms_per_step": 1e3 * dt / steps, "workspace_gib": ws_bytes.value / 2.0 ** 30,
           "weights": "calibrated: block 1 on hi / lo float16 weight pairs, blocks 2-4 on plain float16 weights re-rounded with "
                      "error feedback against channel means measured on one other 4K window (dvsg_locnet_calibrate_f16)",
           "tolerance": "F_t < 1e-5 and warped pixels < 1e-3 against a float64 evaluation of the reference's definition on the "
                        "synthetic checkpoint (tests/test_gpu_configs.py: 1.4e-6 / 5.6e-4; pairs everywhere 9.6e-7 / 3.9e-4); NOT "
                        "met by the float16 mode on a stress checkpoint, with or without pairs (tests/test_gpu_stress.py)",
           "roofline": roofline_object(2, "f16", prof, B, H, W, bound="hbm", calibrated=True),
           "pairs_everywhere": {"value": B * steps / dt_pairs, "ms_per_step": 1e3 * dt_pairs / steps,
                                "roofline": roofline_object(2, "f16", prof_pairs, B, H, W, bound="hbm")}}
    del patches, u_t, outs
    net._ws = None          # give the 4K workspace back
    torch.cuda.empty_cache()
    return res


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes of THIS
    process, which never touches the GPU itself (no exec of a GPU-initialised process, no HIP call
    in the parent), relay rank 0's stdout (the JSON line) and fail if any rank fails."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = b""
    rc = 0
    try:
        # rank 0's stdout is small (one line); a failing peer must not leave the others waiting in a
        # collective forever, so poll everybody and stop the rest once one rank has failed
        import selectors
        sel = selectors.DefaultSelector()
        sel.register(procs[0].stdout, selectors.EVENT_READ)
        eof = False
        while True:
            if not eof:
                for key, _ in sel.select(timeout=0.5):
                    chunk = os.read(key.fileobj.fileno(), 65536)
                    if chunk:
                        out0 += chunk
                    else:
                        eof = True
                        sel.unregister(key.fileobj)
            else:
                time.sleep(0.2)
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0] if bad[0] > 0 else 1
                break
            if all(c == 0 for c in codes) and eof:
                break
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    sys.stdout.write(out0.decode("utf-8", "replace"))
    sys.stdout.flush()
    if rc:
        raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="stabilize", choices=["stabilize", "tf_warp"],
                    help="stabilize: the headline path (CNN + TPS + sampler, configs[1]); tf_warp: the optical-flow "
                         "warp alone (configs[2], batch 64)")
    ap.add_argument("--batch", type=int, default=None,
                    help="windows per GPU and step; default 16 (stabilize, 1 GPU: configs[1]), 64 (stabilize, N > 1 "
                         "GPUs: configs[3]'s shard) / 64 (tf_warp)")
    ap.add_argument("--call-batch", type=int, default=None,
                    help="windows per dvsg_stabilize call; default: the whole batch at 1 GPU, 16 at N > 1 (a rank's "
                         "64-window shard runs as four calls of the configs[1] size)")
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--prof-class", type=int, default=None,
                    help="kernel class timed for the roofline object (default: 1 = 3x3 convs, or 7 for tf_warp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the f32s measurement reported beside the f32 line")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 clip-loop measurement reported beside the line")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the cfg2_tf_warp / cfg4_f16_4k objects (BASELINE configs[2] and configs[4]) reported beside the line")
    ap.add_argument("--precision", default="f32", choices=["f32", "f32s", "f32x3", "f16"],
                    help="f32 (default, the reference's arithmetic: the headline number); f32s: float32 storage / "
                         "accumulation with products from two float16 pieces per operand (dtype f32x2f16); f16: float16 "
                         "activations, hi / lo float16 weight pairs")
    ap.add_argument("--calibrate", action="store_true",
                    help="--precision f16 only: dvsg_locnet_calibrate_f16 on one window of another seed before the run (blocks 2-4 on "
                         "error-feedback-rounded plain float16 weights instead of hi / lo pairs)")
    ap.add_argument("--source", default="window", choices=["window", "ring_f32", "ring_u8"],
                    help="window (default): the [B,H,W,21] float32 window tensor the reference feeds (eval.py:106-110), "
                         "dvsg_stabilize_*; ring_f32 / ring_u8: a pool of 7 B RGB frames (float32, or raw uint8 with the / 255. "
                         "fused) + the [B,7] index table, windows assembled inside conv1's load stage: dvsg_stabilize_ring_*")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the batch of one step is split over (LocNet.stabilize); 2 is <1 %% faster "
                         "but concurrent launches make the per-kernel hipEvent durations of `roofline` meaningless")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    # Rehearsal knobs for a one-GPU box (the real multi-GPU run uses neither): DVSG_BENCH_BACKEND=gloo
    # stages the gather through host memory, DVSG_BENCH_SHARE_DEVICE=1 puts every rank on cuda:0.
    if os.environ.get("DVSG_BENCH_FAIL_RANK") == str(rank) and world > 1:   # test hook: a rank that dies early
        raise SystemExit("rank %d: failing on request (DVSG_BENCH_FAIL_RANK)" % rank)
    backend = os.environ.get("DVSG_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("DVSG_BENCH_SHARE_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # DVSG_BENCH_FORCE_DIST=1: a one-rank process group anyway, so that the RCCL calls of the N > 1 path (async gather,
    # device barrier) run on a one-GPU box
    if world > 1 or os.environ.get("DVSG_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        # "nccl" is RCCL on ROCm; rendezvous comes from the launcher's MASTER_ADDR / MASTER_PORT.  A rank that cannot join
        # (ncclSystemError, rendezvous timeout) says who and where it is and exits non-zero: self_launch / torchrun then
        # stop the others.  Nothing is retried in this process -- it has touched the GPU.
        import datetime
        try:
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=int(os.environ.get("DVSG_BENCH_INIT_TIMEOUT_S", "300"))))
            if backend == "nccl":   # create the communicator now, with every rank present, and fail here if it cannot be
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError("communicator sees %d ranks, expected %d" % (int(probe.item()), world))
        except Exception as exc:   # noqa: BLE001
            env = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                   "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "HIP_VISIBLE_DEVICES")}
            sys.stderr.write("bench.py: rank %d of %d could not join the %s process group on device %d: %s: %s\n  env: %s\n"
                             % (rank, world, backend, dev_index, type(exc).__name__, exc, env))
            sys.stderr.flush()
            raise SystemExit(3)
    on_host = backend != "nccl"

    from coupe.dvsg_amd import _lib
    from coupe.dvsg_amd.networks import LocNet
    from coupe.dvsg_amd.weights import make_synthetic_weights

    flow_mode = args.workload == "tf_warp"
    if args.batch is None:
        args.batch = 64 if (flow_mode or world > 1) else 16
    if args.prof_class is None:
        args.prof_class = 7 if flow_mode else 1
    B, H, W = args.batch, args.height, args.width
    CB = args.call_batch or (B if world == 1 else 16)
    weights = net = patches = u_t = flow = None
    if flow_mode:
        u_t, flow = make_flow_inputs(B, H, W, rank, dev)
    else:
        weights = make_synthetic_weights(seed=0)
        net = LocNet(weights)
        patches = gpu_windows(B, H, W, 1234 + rank, dev)
        u_t = patches[..., 18:].contiguous()
        if args.calibrate:
            if args.precision != "f16":
                raise SystemExit("--calibrate applies to --precision f16")
            net.calibrate_f16(gpu_windows(1, H, W, 499, dev))
        ring_pool = ring_table = None
        if args.source != "window":
            # the same windows as a frame ring: frame 7 b + s of the pool is slot s of window b
            ring_pool = patches.reshape(B, H, W, 7, 3).permute(0, 3, 1, 2, 4).reshape(7 * B, H, W, 3).contiguous()
            if args.source == "ring_u8":
                ring_pool = (ring_pool * 255.0).round_().to(torch.uint8)
            ring_table = torch.arange(7 * B, dtype=torch.int32, device=dev).reshape(B, 7).contiguous()
            patches = u_t = None
    outs = [torch.empty((B, H, W, 3), device=dev) for _ in range(2)]
    F_t = torch.empty((B, 25, 2), device=dev)
    gather_bufs = None
    if dist is not None and not args.no_gather and rank == 0:
        gather_bufs = [[torch.empty((B, H, W, 3), device="cpu" if on_host else dev) for _ in range(world)]
                       for _ in range(2)]

    pending = [None, None]   # in-flight gather per output buffer

    def step(i):
        slot = i & 1
        if pending[slot] is not None:
            # the gather of step i-2 read this buffer: the compute stream must not overwrite it
            # before that gather is done (the gather of step i-1 keeps overlapping this step)
            pending[slot].wait()
            pending[slot] = None
        out = outs[slot]
        if flow_mode:
            _lib.call("dvsg_flow_warp_f32", u_t.data_ptr(), flow.data_ptr(), B, H, W, 3, out.data_ptr(),
                      torch.cuda.current_stream().cuda_stream)
        else:
            for b0 in range(0, B, CB):   # dvsg_stabilize_*, CB windows per call
                b1 = min(B, b0 + CB)
                if ring_pool is not None:
                    net.stabilize_ring(ring_pool, ring_table[b0:b1], out[b0:b1], F_t[b0:b1], precision=args.precision)
                else:
                    net.stabilize(patches[b0:b1], u_t[b0:b1], out[b0:b1], F_t[b0:b1], n_streams=args.streams,
                                  precision=args.precision)
        if dist is not None and not args.no_gather:
            if on_host:   # rehearsal path: synchronous, through host memory
                dist.gather(out.cpu(), gather_bufs[slot] if rank == 0 else None, dst=0)
            else:
                pending[slot] = dist.gather(out, gather_bufs[slot] if rank == 0 else None, dst=0, async_op=True)

    def drain():
        for slot in (0, 1):
            if pending[slot] is not None:
                pending[slot].wait()
                pending[slot] = None

    def barrier():
        if on_host:
            dist.barrier()
        else:
            dist.barrier(device_ids=[dev_index])

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    drain()
    fence()
    if rank == 0:
        _lib.call("dvsg_prof_begin", args.prof_class)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    prof = None
    if rank == 0:
        ms, n, fl, by = ctypes.c_double(), ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        _lib.call("dvsg_prof_end", ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by))
        prof = (ms.value, n.value, fl.value, by.value)
    rank_ms = None
    if dist is not None:
        # the line reports the slowest rank (MAX); the fastest one beside it makes a straggler visible
        t = torch.tensor([elapsed], device="cpu" if on_host else dev, dtype=torch.float64)
        tmin = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        rank_ms = {"min": 1e3 * float(tmin.item()) / args.steps, "max": 1e3 * float(t.item()) / args.steps,
                   "this_rank": 1e3 * elapsed / args.steps}
        elapsed = float(t.item())

    if rank == 0:
        frames = world * B * args.steps
        ms_per_step = 1e3 * elapsed / args.steps
        cls = args.prof_class
        # the profiled unit is one dvsg_stabilize call: CB windows (= B at one GPU, 16 of a rank's 64 at N > 1)
        roofline = roofline_object(cls, args.precision, prof, CB, H, W, kind="tf_warp" if flow_mode else "stabilize",
                                   calibrated=args.calibrate)
        if flow_mode:
            which = "configs[2]" if (B, H, W) == (64, 720, 1280) else "non-BASELINE shape"
        elif (B, H, W) == (16, 720, 1280):
            which = "configs[1]"
        elif (B, H, W, args.precision) == (32, 2160, 3840, "f16"):
            which = "configs[4]"
        elif world > 1 and (B * world, H, W) == (512, 720, 1280):
            which = "configs[3]"
        elif world > 1 and (B, H, W) == (64, 720, 1280):
            which = "configs[3] shard (64 windows per GPU) on %d of its 8 GPUs" % world
        else:
            which = "non-BASELINE shape"
        line = {
            "metric": ("tf_warp frames/sec (%dx%d RGB)" if flow_mode else "stabilized frames/sec (%dx%d RGB)") % (W, H),
            "value": frames / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if flow_mode else {"f32s": "f32x2f16", "f32x3": "f32x3bf16"}.get(args.precision, args.precision),
            "data": "synthetic",
            "config": {"workload": ("%s: batch=%d %dx%d frames, optical-flow warp (warp_with_optical_flow.tf_warp) per GPU"
                                    if flow_mode else
                                    "%s: batch=%d %dx%d 7-frame windows, full CNN+TPS+bilinear warp per GPU") % (which, B, W, H),
                       "batch_per_gpu": B, "windows_per_call": CB, "windows_per_step": B * world,
                       "height": H, "width": W, "parallelism": "window-sharded x%d" % world,
                       "ranks_seen": dist.get_world_size() if dist is not None else 1,
                       "backend": ("rccl" if backend == "nccl" else backend) if dist is not None else None,
                       "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version())
                                        if dist is not None and backend == "nccl" else None),
                       "ms_per_step_over_ranks": rank_ms,
                       "gather": bool(dist is not None and not args.no_gather), "streams_per_gpu": args.streams,
                       "source": "n/a" if flow_mode else args.source, "f16_calibrated": bool(args.calibrate),
                       "weights": "n/a" if flow_mode else "synthetic seed 0 (reference ships no checkpoint)"},
            "roofline": roofline,
        }
        if world == 1 and not flow_mode and args.precision == "f32" and not args.no_secondary and args.source == "window":
            # Beside the line of record (exact float32 matrix cores), the same workload in the "f32s" precision:
            # float32 accumulation, products from two float16 pieces per operand (22 significant bits).  It passes the
            # float32 path's own parity bounds (tests/test_gpu_f32s.py) but is not the reference's arithmetic, so it is
            # reported here, outside the timed region above, and never as `value`.
            try:   # (a secondary figure must never take the line of record down with it)
                for i in range(args.warmup):
                    net.stabilize(patches, u_t, outs[0], F_t, precision="f32s")
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(args.steps):
                    net.stabilize(patches, u_t, outs[i & 1], F_t, precision="f32s")
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                line["secondary"] = {"precision": "f32s", "dtype": "f32x2f16", "value": B * args.steps / dt, "unit": "frames/s",
                                     "ms_per_step": 1e3 * dt / args.steps,
                                     "note": "same workload; float32 storage width and accumulation, every product from two float16 "
                                             "pieces per operand on the f16 matrix cores; F_t within 2e-7 of the exact path"}
            except Exception as exc:   # noqa: BLE001
                line["secondary"] = {"precision": "f32s", "error": str(exc)}
        if world == 1 and not flow_mode and args.precision == "f32" and not args.no_secondary and args.source == "window":
            # ... and in the "f32x3" precision: float32 tensors and accumulation exactly as the line of record, the conv products
            # from three bfloat16 pieces per operand -- all 24 significant bits of both float32 operands at any magnitude, six
            # bfloat16 MFMAs per product.  Against float64 evaluations it is as close as the exact path is, layer by layer and
            # end to end (tests/test_gpu_f32x3.py); its matrix instructions are not float32 ones, so it is reported here,
            # beside the line of record, with the roofline of its own dominant class.
            try:
                def run_x3(i):
                    net.stabilize(patches, u_t, outs[i & 1], F_t, precision="f32x3")
                for i in range(args.warmup):
                    run_x3(i)
                dt, prof_x3 = prof_region(args.prof_class, run_x3, args.steps)
                line["f32x3"] = {"precision": "f32x3", "dtype": "f32 tensors and accumulation; conv products from 3 bf16 pieces per operand",
                                 "value": B * args.steps / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / args.steps,
                                 "roofline": roofline_object(args.prof_class, "f32x3", prof_x3, CB, H, W),
                                 "parity": "tests/test_gpu_f32x3.py: every layer within 1.25 x the exact kernel's own error against float64 "
                                           "math (measured 0.8-1.0 x), pooled features and F_t against the float64 arbiter as close as the "
                                           "exact path (rms 3.1e-8 vs 3.3e-8 relative), the float32 path's oracle bounds unchanged"}
                if not args.no_latency:
                    line["f32x3"]["latency"] = latency_mode(net, dev, precision="f32x3")
            except Exception as exc:   # noqa: BLE001
                line["f32x3"] = {"precision": "f32x3", "error": "%s: %s" % (type(exc).__name__, exc)}
        if world == 1 and not flow_mode and args.precision == "f32" and not args.no_latency:
            try:
                line["latency"] = latency_mode(net, dev)
            except Exception as exc:   # noqa: BLE001
                line["latency"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if (world == 1 and not flow_mode and args.precision == "f32" and not args.no_configs and args.source == "window"
                and (B, H, W) == (16, 720, 1280)):
            # the other single-GPU BASELINE configs in the driver's record, each in its own try, outside the timed region
            # above and after its tensors are released; `value` / `config` / `dtype` of the line are configs[1]'s alone
            del patches, u_t, outs
            torch.cuda.empty_cache()
            for key, leg in (("cfg2_tf_warp", lambda: cfg2_tf_warp(dev, args.steps, args.warmup)),
                             ("cfg4_f16_4k", lambda: cfg4_f16_4k(net, dev, args.steps, args.warmup))):
                try:
                    line[key] = leg()
                except Exception as exc:   # noqa: BLE001
                    line[key] = {"error": "%s: %s" % (type(exc).__name__, exc)}
                torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_flow(H, W) if flow_mode else cpu_baseline(weights, H, W)
        print(json.dumps(line), flush=True)
    if dist is not None:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
