"""Timing probe (not part of the product): one other_configs row of bench.py by name fragment, e.g.
python tools/case_ms.py "configs[2]" "level 8"   (settled clocks, hipEvent per kernel)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, flake_amd
P = flake_amd.level_params
cases = {
    "configs[2]": (P(5, bits_per_sample=24, sample_rate=96000, order_method=flake_amd.OM_SEARCH, max_prediction_order=32, max_partition_order=8), 4096),
    "configs[3]": (P(5, channels=8, bits_per_sample=24, sample_rate=192000, order_method=flake_amd.OM_MAX, max_prediction_order=12), 4096),
    "configs[0]": (P(2, channels=1, block_size=4096), 8192),
    "level 8": (P(8), 4096),
    "level 7": (P(7), 4096),
    "level 2": (P(2), 4096 * 4096 // 1152),
    "search12": (P(5, order_method=flake_amd.OM_SEARCH, max_prediction_order=12, max_partition_order=8), 4096),
    "search32-16": (P(5, order_method=flake_amd.OM_SEARCH, max_prediction_order=32, max_partition_order=8), 4096),
}
for name in sys.argv[1:]:
    p, nfr = cases[name]
    r = bench.subframe_case(0, name, p, nfr, 20, cpu=False)
    print(name, r["ms_per_step"], r["kernel_ms"], flush=True)
