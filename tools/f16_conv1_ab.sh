set -uo pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_f16.py tests/test_gpu_ring.py -m gpu -q -x > gpurun_out/r03_pytest_march.log 2>&1; rc=$?; echo "rc $rc"; tail -5 gpurun_out/r03_pytest_march.log
[ $rc -eq 0 ] || exit 1
DVSG_AMD_LIB=build/lib_stamps.so timeout -k 10 200 python tools/stamp_probe_march.py 8 2160 3840
for v in 4 0 4 0; do
  DVSG_DEBUG=1 DVSG_CONV1_VARIANT=$v timeout -k 10 200 python bench.py --precision f16 --batch 32 --height 2160 --width 3840 --steps 4 --warmup 1 --no-cpu-baseline --prof-class 0 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('conv1_variant $v: %.1f fps %.2f ms/step; conv1 avg %.3f ms' % (d['value'], d['ms_per_step'], r['avg_launch_ms']))"
done
for v in 4 0; do
  DVSG_DEBUG=1 DVSG_CONV1_VARIANT=$v timeout -k 10 100 python bench.py --precision f16 --steps 20 --warmup 5 --no-cpu-baseline --prof-class 0 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('720p conv1_variant $v: %.1f fps %.3f ms/step; conv1 avg %.4f ms' % (d['value'], d['ms_per_step'], r['avg_launch_ms']))"
done
