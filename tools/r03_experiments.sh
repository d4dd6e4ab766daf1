#!/bin/bash
# Round-3 one-off measurements (run on the GPU box): stream-K A/B, float16 4K with 1 / 2 streams, roctx marker trace.
set -uo pipefail
export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_gpu_bench.py -q > $O/r03_pytest_bench.log 2>&1; echo "bench tests rc $?" | tee -a $O/r03_pytest_bench.log
for v in 0 6 0 6; do
  DVSG_DEBUG=1 DVSG_CONV_VARIANT=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-latency 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('conv_variant $v: %.1f frames/s %.3f ms/step; 3x3 class %.1f TFLOP/s avg launch %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms']))" | tee -a $O/r03_streamk_ab.log
done
for st in 1 2 1 2; do
  python bench.py --precision f16 --batch 32 --height 2160 --width 3840 --steps 4 --warmup 1 --no-cpu-baseline --streams $st 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f16 4K B=32 streams $st: %.1f frames/s %.2f ms/step' % (d['value'], d['ms_per_step']))" | tee -a $O/r03_f16_streams.log
done
rm -rf $O/prof_markers
DVSG_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $O/prof_markers -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-latency > $O/prof_markers.log 2>&1
ls $O/prof_markers/*/ | head -20
