P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["dtype"], round(d["value"],1), round(d["ms_per_step"],3), r["kernel"], round(r["avg_launch_ms"],3), round(r["achieved"],1))'
python -m pytest tests/test_gpu_f16.py tests/test_gpu_configs.py -q -x 2>&1 | tail -3
for v in 5 0 5 0; do echo "variant $v (5 = no 256x128 tiles)"; DVSG_DEBUG=1 DVSG_CONV_VARIANT=$v python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 --precision f16 | python -c "$P"; done
for v in 5 0; do echo "4K variant $v"; DVSG_DEBUG=1 DVSG_CONV_VARIANT=$v python bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 1 --precision f16 --batch 32 --height 2160 --width 3840 | python -c "$P"; done
