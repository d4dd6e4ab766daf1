#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: kernel-trace stats and the three PMC passes
# (own runs, --kernel-trace only: gpurun refuses --pmc combined with other trace domains) of one
# short bench run.  Usage: tools/profile_round.sh TAG [bench.py flags of the profiled workload]
#   tools/profile_round.sh r02                                              (configs[1]: B=16 720p f32)
#   tools/profile_round.sh r02_f16 --precision f16 --batch 32 --height 2160 --width 3840   (configs[4])
set -euo pipefail
TAG=${1:-r02}
shift || true
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
rm -rf "$OUT"
mkdir -p "$OUT"
# --no-secondary / --no-latency: the f32s leg and the batch-1 clip loop bench.py reports beside the f32 line must not be in
# the profile (round 2 mixed the f32s launches into the f32 traffic file)
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-latency --no-configs $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/pmc_sq.log" 2>&1
echo "sq pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1
tail -1 "$OUT/stats.log"
python3 tools/summarize_profiles.py "$OUT" "$OUT" "$TAG" "$*" > "$OUT/summary.log" 2>&1 || tail -5 "$OUT/summary.log"
echo "profiles written under $OUT (summaries: $OUT/${TAG}_*.csv / .json -- copy them into profiles/)"
