#!/bin/bash
# Round-4 evidence run (on the GPU box): the default bench line, the profile set of the headline workload, bench lines of
# every config / precision / class, batch-1 marker tables, the CPU baseline in full.  (The GPU suite and the tf_warp /
# float16 profile sets have their own calls: tools/profile_round.sh r04_flow --workload tf_warp, ... r04_f16 ... --calibrate.)
set -uo pipefail
export TMPDIR=/tmp
O=gpurun_out
python bench.py > $O/r04_bench_default.json 2> $O/r04_bench_default.err; tail -c 400 $O/r04_bench_default.json; echo
bash tools/profile_round.sh r04 > $O/r04_prof.log 2>&1; tail -2 $O/r04_prof.log
bash tools/profile_round.sh r04_x3 --precision f32x3 > $O/r04_prof_x3.log 2>&1; tail -2 $O/r04_prof_x3.log
rm -f $O/r04_bench_lines.jsonl
for args in "" "--precision f16 --batch 32 --height 2160 --width 3840 --steps 4 --warmup 1" \
            "--precision f16 --batch 32 --height 2160 --width 3840 --steps 4 --warmup 1 --calibrate" "--precision f16" "--precision f16 --calibrate" \
            "--precision f32s" "--precision f32x3" "--precision f32x3 --prof-class 2" "--precision f32x3 --prof-class 0" "--precision f32x3 --batch 1 --steps 50 --warmup 10" "--workload tf_warp --steps 30 --warmup 5" "--batch 1 --steps 50 --warmup 10" "--batch 64 --steps 5 --warmup 2" \
            "--prof-class 2" "--prof-class 0" "--prof-class 8" "--prof-class 6" "--prof-class 3" "--source ring_f32" "--source ring_u8"; do
  python bench.py --no-cpu-baseline --no-secondary --no-latency --no-configs $args 2>/dev/null >> $O/r04_bench_lines.jsonl
  echo "line: $args"
done
wc -l $O/r04_bench_lines.jsonl
bash tools/b1_markers.sh r04 720 1280 > $O/r04_b1_720.log 2>&1; bash tools/b1_markers.sh r04 288 512 > $O/r04_b1_288.log 2>&1
python tools/clip_latency.py > $O/r04_clip_latency.log 2>&1; python tools/clip_latency.py 288 512 >> $O/r04_clip_latency.log 2>&1; cat $O/r04_clip_latency.log
python tools/cpu_baseline_full.py > $O/r04_cpu_baseline.json 2> $O/r04_cpu_baseline.err; tail -12 $O/r04_cpu_baseline.json
