"""Debug probe: first differing words of the Rice sections, device vs oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, flake_amd
from oraclelib import Oracle
o = Oracle()
p = flake_amd.level_params(5, order_method=flake_amd.OM_MAX)
n = p.block_size
pcm = flake_amd.synth_pcm(2, n, 2, 16, first_frame=17)
with flake_amd.Encoder(p, max_frames=2) as enc:
    got = enc.encode_subframes(pcm, n)
exp = o.encode_subframes_batch(p, pcm, n, slot_bytes=got["slot_bytes"])
for s in range(2):
    gi, ei = got["info"][s], exp["info"][s]
    print(s, "order", gi["order"], ei["order"], "porder", gi["porder"], ei["porder"], "nbits", gi["rice_nbits"], ei["rice_nbits"],
          "k0", gi["rparams"][:4], ei["rparams"][:4])
    print("  resid equal:", np.array_equal(got["residual"][s], exp["residual"][s]),
          np.nonzero(got["residual"][s] != exp["residual"][s])[0][:10])
    nb = (int(ei["rice_nbits"]) + 7) // 8
    g = got["rice_bits"][s][:nb]; e = exp["rice_bits"][s][:nb]
    bad = np.nonzero(g != e)[0]
    print("  bytes differing:", bad.size, bad[:16])
    for b in bad[:6]:
        print("   byte", b, format(int(g[b]), "08b"), format(int(e[b]), "08b"))
