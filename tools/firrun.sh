for g in "" "18,64"; do
FHIP_K3_GEOM=$g python - <<PY
import sys, json
sys.path.insert(0, '.')
import bench, flake_amd
P = flake_amd.level_params
rows = [bench.subframe_case(0, "level 2", P(2), 4096 * 4096 // 1152, 20, cpu=False),
        bench.subframe_case(0, "level 0", P(0), 4096 * 4096 // 1152, 20, cpu=False)]
for r in rows: print("geom '$g'", r["workload"][:22], r["ms_per_step"], r.get("kernel_ms"))
PY
done
FHIP_K3_GEOM="18,128" python - <<PY
import sys, json
sys.path.insert(0, '.')
import bench, flake_amd
P = flake_amd.level_params
r = bench.subframe_case(0, "level 2 n=2304", P(2, block_size=2304), 4096 * 4096 // 2304, 20, cpu=False); print("18,128", r["ms_per_step"], r["kernel_ms"])
PY
python - <<PY
import sys, json
sys.path.insert(0, '.')
import bench, flake_amd
P = flake_amd.level_params
r = bench.subframe_case(0, "level 2 n=2304", P(2, block_size=2304), 4096 * 4096 // 2304, 20, cpu=False); print("default 2304", r["ms_per_step"], r["kernel_ms"])
PY
