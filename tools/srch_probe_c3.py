"""Timing probe (not part of the product): k_order_search on configs[2] only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flake_amd
from ablate import run
P = flake_amd.level_params
run("search32 24bit", P(5, bits_per_sample=24, order_method=flake_amd.OM_SEARCH, max_prediction_order=32, max_partition_order=8), nframes=4096, steps=3)
