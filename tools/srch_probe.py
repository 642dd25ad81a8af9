"""Timing probe (not part of the product): k_order_search on configs[2] and level 8.
FHIP_LIB=<variant .so> python tools/srch_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flake_amd
from ablate import run
P = flake_amd.level_params
run("search32 24bit " + os.environ.get("FHIP_LIB", "")[-12:], P(5, bits_per_sample=24, order_method=flake_amd.OM_SEARCH, max_prediction_order=32, max_partition_order=8), nframes=4096, steps=5)
run("level8 " + os.environ.get("FHIP_LIB", "")[-12:], P(8), nframes=4096, steps=10)
run("level7 " + os.environ.get("FHIP_LIB", "")[-12:], P(7), nframes=4096, steps=10)
