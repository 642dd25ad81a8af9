"""Timing probe (not part of the product): per-kernel times of configs[2] and configs[3]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flake_amd
from ablate import run
P = flake_amd.level_params
tag = os.environ.get("FHIP_LIB", "")[-8:]
run("c4 8ch24 lpc12 " + tag, P(5, channels=8, bits_per_sample=24, sample_rate=192000, order_method=flake_amd.OM_MAX, max_prediction_order=12), nframes=4096, steps=10)
run("lvl5 24bit " + tag, P(5, bits_per_sample=24, order_method=flake_amd.OM_MAX), nframes=4096, steps=10)
run("c3 " + tag, P(5, bits_per_sample=24, order_method=flake_amd.OM_SEARCH, max_prediction_order=32, max_partition_order=8), nframes=4096, steps=3)
