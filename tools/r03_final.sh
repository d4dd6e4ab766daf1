#!/bin/bash
# Round-3 evidence run (on the GPU box): full GPU suite, the default bench line, the three profile sets, bench lines of every config.
set -uo pipefail
export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -m gpu -q -s > $O/r03_pytest_final.log 2>&1; echo "pytest rc $?" | tee -a $O/r03_pytest_final.log; tail -3 $O/r03_pytest_final.log
python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err; tail -c 600 $O/r03_bench_default.json; echo
bash tools/profile_round.sh r03 > $O/r03_prof.log 2>&1; tail -2 $O/r03_prof.log
bash tools/profile_round.sh r03_f16 --precision f16 --batch 32 --height 2160 --width 3840 > $O/r03_prof_f16.log 2>&1; tail -2 $O/r03_prof_f16.log
rm -f $O/r03_bench_lines.jsonl
for args in "" "--precision f16 --batch 32 --height 2160 --width 3840 --steps 4 --warmup 1" "--precision f16" "--precision f32s" \
            "--workload tf_warp" "--batch 1 --steps 50 --warmup 10" "--batch 64 --steps 5 --warmup 2" "--prof-class 2" "--prof-class 0" "--prof-class 8" "--prof-class 6" "--prof-class 3"; do
  python bench.py --no-cpu-baseline --no-secondary --no-latency $args 2>/dev/null >> $O/r03_bench_lines.jsonl
done
wc -l $O/r03_bench_lines.jsonl
python tools/clip_latency.py > $O/r03_clip_latency.log 2>&1; python tools/clip_latency.py 288 512 >> $O/r03_clip_latency.log 2>&1; cat $O/r03_clip_latency.log
