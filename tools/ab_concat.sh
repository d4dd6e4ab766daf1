for v in 0 1 0 1; do
  DVSG_DEBUG=1 DVSG_CONCAT_SC=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-latency --no-configs 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('concat_sc $v: %.1f frames/s %.3f ms/step' % (d['value'], d['ms_per_step']))"
done
for v in 0 1 0 1; do
  DVSG_DEBUG=1 DVSG_CONCAT_SC=$v python bench.py --precision f32s --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-latency --no-configs 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32s concat_sc $v: %.1f frames/s %.3f ms/step' % (d['value'], d['ms_per_step']))"
done
