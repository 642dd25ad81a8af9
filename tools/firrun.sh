timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vbs_dev.py tests/test_gpu_ref_path.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r03_t10.log 2>&1; tail -5 gpurun_out/r03_t10.log
for mm in 0 1; do
if [ $mm = 0 ]; then export FHIP_NO_MM=1; else unset FHIP_NO_MM; fi
python - <<PY
import sys, json
sys.path.insert(0, '.')
import bench, flake_amd
P = flake_amd.level_params
rows = [bench.subframe_case(0, "configs[2]", P(5, bits_per_sample=24, sample_rate=96000, order_method=flake_amd.OM_SEARCH, max_prediction_order=32, max_partition_order=8), 4096, 10, cpu=False),
        bench.subframe_case(0, "search12 16-bit", P(10, variable_block_size=0, allow_vbs=0), 4096, 10, cpu=False),
        bench.vbs_case(0, 10, 1024, 10, cpu=False), bench.vbs_case(0, 12, 1024, 10, cpu=False)]
for r in rows: print("mm=$mm", r["workload"][:22], r["ms_per_step"], r.get("kernel_ms") or r.get("kernel_ms_serial_sums"))
PY
done
