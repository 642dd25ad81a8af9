timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_path.py tests/test_gpu_fuzz.py tests/test_gpu_frames.py -q -x -m gpu -p no:cacheprovider > gpurun_out/t_leaf.log 2>&1 || { tail -30 gpurun_out/t_leaf.log; exit 1; }
tail -2 gpurun_out/t_leaf.log
for i in 1 2; do python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print(b['ms_per_step'], b['roofline']['kernel_ms'])"; done
timeout -k 10 200 python tools/srch_probe.py 2>&1 | grep -E "level" | sed 's/k_assemble.*k_order/k_order/'
