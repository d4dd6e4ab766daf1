#!/bin/bash
# SQ counter passes over tools/x3_once.py (run ON THE GPU BOX from the repo root): matrix-pipe busy, issue / wait split, LDS.
# Usage: tools/pmc_x3.sh TAG [f32|f32x3]
set -euo pipefail
TAG=${1:-x3}
PREC=${2:-f32x3}
OUT=gpurun_out/pmc_$TAG
export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d "$OUT/$1" -- python3 tools/x3_once.py $PREC > "$OUT/$1.log" 2>&1 || tail -3 "$OUT/$1.log"; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
run sq2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
run sq3 "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/*/")):
    cc = glob.glob(d + "**/*_counter_collection.csv", recursive=True)
    kt = glob.glob(d + "**/*_kernel_trace.csv", recursive=True)
    if not cc:
        print(d, "no counters"); continue
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0]))}
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(cc[0])):
        if "conv_gemm_kernel" in r["Kernel_Name"]:
            key = (r["Grid_Size"], r["Counter_Name"])
            acc[key] += float(r["Counter_Value"]); n[key].add(r["Dispatch_Id"])
    for k in sorted(acc):
        ids = n[k]
        print("grid %-9s %-34s %16.5g per launch   (%d launches, %.1f us each)" % (k[0], k[1], acc[k] / len(ids), len(ids), sum(dur[i] for i in ids) / len(ids) / 1e3))
PY
