set -uo pipefail
python -m pytest tests/test_gpu_f16.py tests/test_gpu_ring.py tests/test_gpu_cnn.py tests/test_gpu_f32s.py tests/test_gpu_golden.py -m gpu -q -x > gpurun_out/r03_pytest_pair.log 2>&1; echo "rc $?"; tail -3 gpurun_out/r03_pytest_pair.log
for v in 3 0 3 0; do
  DVSG_DEBUG=1 DVSG_CONV1_VARIANT=$v python bench.py --precision f16 --batch 32 --height 2160 --width 3840 --steps 4 --warmup 1 --no-cpu-baseline --prof-class 0 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('conv1_variant $v: %.1f fps %.2f ms/step; conv1 avg %.3f ms' % (d['value'], d['ms_per_step'], r['avg_launch_ms']))"
done
for p in f32 f32 f16 f32s; do
  python bench.py --precision $p --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-latency --prof-class 0 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('720p B=16 $p: %.1f fps %.3f ms/step; conv1 avg %.4f ms (%.1f TFLOP/s)' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['achieved']))"
done
