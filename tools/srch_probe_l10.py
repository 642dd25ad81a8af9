"""Timing probe (not part of the product): k_order_search, SEARCH over small maximum orders."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flake_amd
from ablate import run
P = flake_amd.level_params
tag = os.environ.get("FHIP_LIB", "")[-12:]
for bps in (16, 24):
    for mo in (8, 12, 16, 24):
        run(f"{tag} search {mo} {bps}-bit", P(5, bits_per_sample=bps, order_method=flake_amd.OM_SEARCH, max_prediction_order=mo, max_partition_order=8), nframes=4096, steps=3)
