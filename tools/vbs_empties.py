"""Measurement probe (not part of the product): a ragged (variable-block-size) batch whose blocks all stay whole -- white
noise of one level -- so that the kernels of bins 0..6 launch their capacity of workgroups and find nothing: what an
empty workgroup costs.   rocprofv3 --kernel-trace --stats -- python3 tools/vbs_empties.py [blocks] [level]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, flake_amd

nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
level = int(sys.argv[2]) if len(sys.argv) > 2 else 10
p = flake_amd.level_params(level)
n = p.block_size
rng = np.random.default_rng(7)
pcm = rng.integers(-3000, 3000, size=(nblk, n, 2), dtype=np.int32)
dev = torch.device("cuda", 0)
d_pcm = torch.from_numpy(pcm).to(dev)
cap = pcm.size * 5
packed = torch.zeros(cap, dtype=torch.uint8, device=dev)
totals = torch.zeros(4, dtype=torch.int64, device=dev)
enc = flake_amd.Encoder(p, max_frames=8 * nblk)
st = torch.cuda.Stream(dev)
torch.cuda.synchronize(dev)
enc.set_stream(st.cuda_stream)
for _ in range(6):
    enc.encode_blocks_vbs_dev(d_pcm, nblk, n, packed, cap, totals)
enc.sync()
print("frames", int(totals.cpu()[0]), "of", nblk, "blocks")
enc.close()
