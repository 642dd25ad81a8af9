timeout -k 10 1000 python -m pytest tests -q -x -m gpu -p no:cacheprovider > gpurun_out/t_full.log 2>&1 || { tail -30 gpurun_out/t_full.log; exit 1; }
tail -2 gpurun_out/t_full.log
python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/b_lb5.json; python - <<'PY'
import json
b=json.load(open('gpurun_out/b_lb5.json'))
print(b['ms_per_step'], b['roofline']['kernel_ms'])
for o in b['other_configs']: print(o['workload'][:40], o['ms_per_step'], o['kernel_ms'])
print(b['host_path'])
PY
