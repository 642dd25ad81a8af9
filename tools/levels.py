"""Timing probe (not part of the product): per-kernel times of every compression
level preset on 33.5 M stereo 16-bit samples.  python tools/levels.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flake_amd
from ablate import run
for lvl in range(0, 9):
    p = flake_amd.level_params(lvl)
    run(f"level {lvl} n={p.block_size}", p, nframes=4096 * 4096 // p.block_size, steps=5)
