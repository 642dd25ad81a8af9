import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, flake_amd
from test_gpu_ahead import _run
p = flake_amd.level_params(5, channels=6, bits_per_sample=16)
n, nfr, ch = p.block_size, 70, 6
dev = torch.device("cuda", 0)
a = torch.from_numpy(flake_amd.synth_pcm(nfr, n, ch, 16, first_frame=0)).to(dev)
b = torch.from_numpy(flake_amd.synth_pcm(nfr, n, ch, 16, first_frame=1000)).to(dev)
torch.cuda.synchronize()
with flake_amd.Encoder(p, max_frames=nfr) as enc:
    enc.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    for tag, t, h in (("plain a", a, None), ("plain b", b, None), ("plain a", a, None), ("hint a", a, a), ("hint b", b, b), ("hint a", a, a), ("plain b", b, None)):
        i, _ = _run(enc, t, nfr, n, p, hint=h)
        print(tag, i["obits"][:8], i["wasted"][:4], i["ch_mode"][:4], i["type"][:4], int(i["rice_nbits"].sum()))
