for lib in libflakehip.so libflakehip_w4.so; do
FHIP_LIB=$PWD/flake_amd/lib/$lib python - <<PY
import sys, json
sys.path.insert(0, '.')
import bench, flake_amd
P = flake_amd.level_params
rows = [bench.subframe_case(0, "configs[2]", P(5, bits_per_sample=24, sample_rate=96000, order_method=flake_amd.OM_SEARCH, max_prediction_order=32, max_partition_order=8), 4096, 10, cpu=False),
        bench.vbs_case(0, 10, 1024, 10, cpu=False), bench.vbs_case(0, 12, 1024, 10, cpu=False)]
for r in rows: print("$lib", r["workload"][:22], r["ms_per_step"], r.get("kernel_ms") or r.get("kernel_ms_serial_sums"))
PY
done
