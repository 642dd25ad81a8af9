"""Measurement probe (not part of the product): what the order search and K3 cost per subframe at each piece length of a
ragged level-10 / level-12 batch, each length as a uniform batch of its own (same samples per batch).
python tools/piece_cost.py [level]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, flake_amd

level = int(sys.argv[1]) if len(sys.argv) > 1 else 10
blk = flake_amd.level_params(level).block_size
for k in range(1, 9):
    n = blk * k // 8
    p = flake_amd.level_params(level, variable_block_size=0, block_size=n)
    nframes = (1 << 23) // n
    r = bench.subframe_case(0, f"n {n}", p, nframes, 10, cpu=False)
    km = r["kernel_ms"]
    nsub = 2 * nframes
    print(json.dumps({"n": n, "subframes": nsub, "ms": r["ms_per_step"],
                      "ns_per_sub": {q: round(v * 1e6 / nsub, 1) for q, v in km.items()},
                      "ns_per_ksample": {q: round(v * 1e6 / nsub / n * 1000, 1) for q, v in km.items()}}), flush=True)
