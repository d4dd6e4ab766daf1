#!/bin/bash
# Run ON THE GPU BOX from the repo root: the bench line of every BASELINE config, kernel class and precision in one
# session -> gpurun_out/bench_lines.jsonl (copied to profiles/rNN_bench_lines.jsonl).
set -euo pipefail
OUT=gpurun_out/bench_lines.jsonl
mkdir -p gpurun_out
: > "$OUT"
run() { python3 bench.py --no-cpu-baseline "$@" | tail -1 >> "$OUT"; echo "done: $*"; }
python3 bench.py | tail -1 >> "$OUT"; echo "done: default (with cpu_baseline and secondary)"
run --workload tf_warp --steps 20 --warmup 5                                                   # configs[2]
run --precision f16 --batch 32 --height 2160 --width 3840 --steps 4 --warmup 1            # configs[4]
run --precision f16 --steps 20 --warmup 5 --no-secondary
for c in 0 2 6 8; do run --steps 20 --warmup 5 --prof-class $c --no-secondary; done
for c in 1 0 2 8; do run --precision f32s --steps 20 --warmup 5 --prof-class $c; done
# configs[3] rehearsal: 2 of its 8 shards on this one GPU (gloo, host gather)
DVSG_BENCH_BACKEND=gloo DVSG_BENCH_SHARE_DEVICE=1 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline | tail -1 >> "$OUT"
echo "wrote $OUT"
