#!/bin/bash
# rocprofv3 kernel stats of the float16 step at configs[4] (or "$@" bench flags): per-kernel ms per step
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
STEPS=4
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_f16_step -o f16 --output-format csv -- python3 bench.py --precision f16 --batch 32 --height 2160 --width 3840 --steps $STEPS --warmup 1 --no-cpu-baseline --no-secondary --no-latency --no-configs "$@" > gpurun_out/prof_f16_step.log 2>&1
f=$(find gpurun_out/prof_f16_step -name "*kernel_stats.csv" | head -1)
python3 - "$f" $STEPS <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=int(sys.argv[2])+1
tot=0
for r in rows:
    n=r['Name'].replace('dvsg::(anonymous namespace)::','')
    if 'at::' in n or 'rocclr' in n: continue
    ms=float(r['TotalDurationNs'])/1e6/steps
    tot+=ms
    if ms>0.3: print("%-100s %5.1f calls/step avg %8.1f us  %6.2f ms/step"%(n[:100], int(r['Calls'])/steps, float(r['AverageNs'])/1e3, ms))
print("sum %.1f ms/step"%tot)
PY
grep -o '"value": [0-9.]*, "unit"' gpurun_out/prof_f16_step.log | head -1
