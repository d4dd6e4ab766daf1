#!/bin/bash
# PMC passes over tools/flow_once.py (run ON THE GPU BOX from the repo root): SQ wait / issue split, the texture
# addresser / vector L1 / L2 request counters of the tf_warp kernel.  Usage: tools/pmc_flow.sh TAG [cfg3|const|noise]
set -euo pipefail
TAG=${1:-flow}
KIND=${2:-cfg3}
OUT=gpurun_out/pmc_$TAG
export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d "$OUT/$1" -- python3 tools/flow_once.py $KIND > "$OUT/$1.log" 2>&1 || tail -3 "$OUT/$1.log"; }
run sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_BUSY_CU_CYCLES"
run ta "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
run tcp1 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
run tcp2 "TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_EA0_RDREQ_sum"
run tcc2 "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_BUSY_sum"
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
        if "stn_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
    for k in acc:
        ids = n[k]
        print("%-40s %16.4g per launch   (%d launches, %.1f us each)" % (k, acc[k] / len(ids), len(ids), sum(dur[i] for i in ids) / len(ids) / 1e3))
PY
