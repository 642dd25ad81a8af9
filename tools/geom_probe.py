import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); 
import bench, flake_amd
P = flake_amd.level_params
for n in (3072, 3584, 2560, 6144):
    p = P(10, block_size=n, variable_block_size=0)
    r = bench.subframe_case(0, f"n={n}", p, 1024 * 4096 // n, 10, cpu=False)
    print(n, os.environ.get("FHIP_K3_GEOM"), r["ms_per_step"], r["kernel_ms"], flush=True)
