import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import flake_amd
from ablate import run
P = flake_amd.level_params
which = sys.argv[1]
if which == "1152":
    run("lvl2 n=1152", P(2), nframes=4096 * 4096 // 1152, steps=5)
    run("lvl5max n=1152", P(5, block_size=1152, order_method=flake_amd.OM_MAX), nframes=4096 * 4096 // 1152, steps=5)
else:
    run("lvl5max n=4608", P(5, block_size=4608, order_method=flake_amd.OM_MAX), nframes=4096 * 4096 // 4608, steps=5)
    run("lvl2 n=4608", P(2, block_size=4608), nframes=4096 * 4096 // 4608, steps=5)
