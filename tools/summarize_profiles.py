#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile_round.sh into the committed summaries:
    python tools/summarize_profiles.py gpurun_out/prof_r01 profiles r01
writes profiles/<tag>_kernel_stats.csv (rocprofv3 --stats, verbatim), <tag>_pmc_{sq,fetch,write}.csv
(per kernel symbol sums over the run) and <tag>_traffic.json (HBM-side bytes per launch per kernel
class, corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are in KiB, and on gfx950
FETCH_SIZE reads exactly half of a wide coalesced stream, so it is doubled)."""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

CLASSES = {"conv1_kernel": 0, "conv1_x3_kernel": 0, "conv1_f16_kernel": 0, "conv1_f16_pair_kernel": 0, "conv1_f16_march_kernel": 0, "conv1_split_kernel": 0, "maxpool_p_kernel": 3, "maxpool_h8_kernel": 3, "avgpool_partial_p_kernel": 4, "conv_gemm": None, "maxpool_kernel": 3, "tps_solve_kernel": 5, "tps_warp_kernel": 6,
           "stn_kernel": 7, "flow_warp_strip_kernel": 7, "mask_plane_kernel": 7, "dense_kernel": 4, "avgpool_partial_kernel": 4, "dense_finalize_kernel": 4}


def short(n):
    n = re.sub(r"dvsg::\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", n).replace("void ", "")


def kclass(name):
    s = short(name)
    m = re.search(r"_GLOBAL__N_1(\d+)", s)   # rocprofv3 leaves names with _Float16 in them mangled: <length><name>
    if s.startswith("_ZN") and m:
        base = s[m.end():m.end() + int(m.group(1))]
        if base == "conv_gemm_kernel":
            li = re.findall(r"Li(\d+)E", s)      # <T, BN, WM, WN, KS, ...>
            return 1 if len(li) >= 4 and li[3] == "3" else 2
        if base in ("conv_wide16_kernel", "conv_wide16a_kernel"):   # the float16 mode's 256 x 128 tiles: <KS, RELU, RES>
            li = re.findall(r"Li(\d+)E", s)
            return 1 if li and li[0] == "3" else 2
        if base == "conv_wide16h_kernel":        # ... their 3x3 stride-1 form
            return 1
        s = base
    if s.startswith("conv_wide16h_kernel"):
        return 1
    if s.startswith(("conv_wide16_kernel", "conv_wide16a_kernel")):
        m = re.match(r"conv_wide16a?_kernel<(\d+)", s)
        return 1 if m and m.group(1) == "3" else 2
    if s.startswith("conv3x3_1x1"):          # block 1's fused conv2 + conv3 (float32 / pieces / float16 kernels)
        return 8
    if s.startswith("conv_gemm"):
        m = re.match(r"conv_gemm_kernel<\w+, (\d+), (\d+), (\d+), (\d+)", s)   # <T, BN, WM, WN, KS, ...>
        return 1 if m and m.group(4) == "3" else 2
    for k, v in CLASSES.items():
        if s.startswith(k):
            return v
    return None


def kprec(name):
    """Precision a kernel symbol belongs to: "f32" (exact float32 matrix cores), "f32s" (float32 from two float16
    pieces: SPLIT conv_gemm with T = float, conv1_split_kernel, the *_p_kernel family, conv3x3_1x1_kernel<*, true>),
    "f16" (float16 activations), or None for the kernels every mode shares (head, TPS, samplers, tickets).  A traffic
    file holds ONE precision: round 2's mixed the f32 launches with the f32s launches of bench.py's `secondary` leg."""
    s = short(name)
    if s.startswith("_ZN"):
        m = re.search(r"_GLOBAL__N_1(\d+)", s)
        base = s[m.end():m.end() + int(m.group(1))] if m else s
        if base == "conv1_split_kernel":
            return "f32s"
        if base == "conv1_x3_kernel":
            return "f32x3"
        if base == "conv_gemm_kernel":   # <T, BN, WM, WN, KS, RELU, RES, MODE, SPLIT, X3>
            if "IDF16_" in s:
                return "f16"
            lb = re.findall(r"Lb([01])E", s.split("EEv")[0] + "E")
            return "f32x3" if lb[-1:] == ["1"] else ("f32s" if lb[-2:-1] == ["1"] else "f32")
        return "f16" if "DF16_" in s else None
    if s.startswith(("conv_wide16", "conv3x3_1x1_f16_kernel", "conv1_f16_kernel", "maxpool_h8_kernel")) or "_Float16" in s:
        return "f16"
    if s.startswith(("conv1_split_kernel", "maxpool_p_kernel", "avgpool_partial_p_kernel")):
        return "f32s"
    m = re.match(r"conv3x3_1x1_kernel<\d+, (true|false)>", s)
    if m:
        return "f32s" if m.group(1) == "true" else "f32"
    m = re.match(r"conv_gemm_kernel<float(?:, \w+)*, (true|false), (true|false)>", s)   # ..., SPLIT, X3>
    if m:
        return "f32x3" if m.group(2) == "true" else ("f32s" if m.group(1) == "true" else "f32")
    if s.startswith("conv1_x3_kernel"):
        return "f32x3"
    if s.startswith(("conv1_kernel", "maxpool_kernel<float>", "avgpool_partial_kernel<float>")):
        return "f32"
    return None


def newest(pattern):
    """gpurun merges every call's files into the same directory: take the latest run's."""
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def counters(d):
    cc = newest(d + "/**/*_counter_collection.csv")
    kt = newest(d + "/**/*_kernel_trace.csv")
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt))}
    per = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for r in csv.DictReader(open(cc)):
        k = r["Kernel_Name"]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return per, disp, dur


def write_pmc(d, path, precision, drop_foreign=False):
    per, disp, dur = counters(d)
    # (the f32x3 precision runs the pools, the head and block 1's opening unit on the float32 kernels themselves)
    allowed = (None, precision, "f32") if precision == "f32x3" else (None, precision)
    foreign = sorted({short(k) for k in per if kclass(k) is not None and kprec(k) not in allowed})
    if foreign and drop_foreign:    # bench.py --calibrate: ONE float32 pass over one window before the float16 run; left out
        for k in [k for k in per if kclass(k) is not None and kprec(k) not in allowed]:
            del per[k]
            del disp[k]
        foreign = []
    if foreign:
        raise SystemExit("%s: launches of another precision than %s in this pass (profile with --no-secondary): %s"
                         % (d, precision, foreign[:4]))
    names = sorted({c for k in per for c in per[k]})
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "total_us"] + names)
        for k in sorted(per, key=lambda k: -sum(dur.get(i, 0) for i in disp[k])):
            if kclass(k) is None:
                continue
            w.writerow([short(k), len(disp[k]), "%.1f" % sum(dur.get(i, 0) for i in disp[k])] + ["%.6g" % per[k][n] for n in names])
    return per, disp


def workload_of(flags):
    """The profiled bench.py workload (batch, height, width, precision) from its command-line flags."""
    import shlex
    a = shlex.split(flags or "")
    get = lambda k, d: a[a.index(k) + 1] if k in a else d
    kind = get("--workload", "stabilize")
    w = {"batch": int(get("--batch", 64 if kind == "tf_warp" else 16)), "height": int(get("--height", 720)),
         "width": int(get("--width", 1280)), "precision": get("--precision", "f32")}
    if kind != "stabilize":
        w["kind"] = kind          # bench.py --workload tf_warp (BASELINE configs[2]): class 7 only
    if "--calibrate" in a:
        w["calibrated"] = True    # float16 mode after dvsg_locnet_calibrate_f16: blocks 2-4 without the lo weight piece
    return w


def main(src, dst, tag, flags=""):
    stats = newest(src + "/stats/**/*_kernel_stats.csv")
    shutil.copy(stats, "%s/%s_kernel_stats.csv" % (dst, tag))
    prec = workload_of(flags)["precision"]
    cal = "--calibrate" in (flags or "")
    write_pmc(src + "/pmc_sq", "%s/%s_pmc_sq.csv" % (dst, tag), prec, cal)
    fetch, fdisp = write_pmc(src + "/pmc_fetch", "%s/%s_pmc_fetch.csv" % (dst, tag), prec, cal)
    write, wdisp = write_pmc(src + "/pmc_write", "%s/%s_pmc_write.csv" % (dst, tag), prec, cal)
    traffic = {}
    for cls in range(9):
        fb = sum(fetch[k]["FETCH_SIZE"] for k in fetch if kclass(k) == cls) * 1024.0 * 2.0
        fn = sum(len(fdisp[k]) for k in fetch if kclass(k) == cls)
        wb = sum(write[k]["WRITE_SIZE"] for k in write if kclass(k) == cls) * 1024.0
        wn = sum(len(wdisp[k]) for k in write if kclass(k) == cls)
        if fn and wn:
            traffic[str(cls)] = {"fetch_bytes_per_launch": fb / fn, "write_bytes_per_launch": wb / wn,
                                 "bytes_per_launch": fb / fn + wb / wn, "launches_profiled": fn}
    json.dump({"note": "FETCH_SIZE x1024 x2 (gfx950 half-count correction) + WRITE_SIZE x1024, separate --pmc passes; "
                       "memory-side (fabric) requests, Infinity-Cache hits included",
               "workload": workload_of(flags), "precision_checked": "every profiled conv / pool launch is a %s kernel" % prec,
               "classes": traffic},
              open("%s/%s_traffic.json" % (dst, tag), "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
