"""Timing probe (not part of the product): ms/step of consecutive groups of 20
steps from a cold start -- how long the clocks take to settle.  python tools/ramp.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, flake_amd
p = flake_amd.level_params(5, order_method=flake_amd.OM_MAX)
n, nfr = 4096, 4096
dev = torch.device("cuda", 0)
nsub = nfr * 2
slot = flake_amd.rice_slot_bytes(p, n)
pcm = torch.from_numpy(flake_amd.synth_pcm(nfr, n, 2, 16)).to(dev)
info = torch.zeros(nsub * flake_amd.INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
bits = torch.zeros(nsub * slot, dtype=torch.uint8, device=dev)
enc = flake_amd.Encoder(p, max_frames=nfr)
st = torch.cuda.Stream(); enc.set_stream(st.cuda_stream)
enc.encode_subframes_dev(pcm, nfr, n, info, rice_bits=bits, slot_bytes=slot)
torch.cuda.synchronize()
time.sleep(1.0)
out = []
for g in range(40):
    t0 = time.perf_counter()
    for _ in range(20):
        enc.encode_subframes_dev(pcm, nfr, n, info, rice_bits=bits, slot_bytes=slot)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 20 * 1e3)
print("ms/step per group of 20:", " ".join(f"{x:.4f}" for x in out))
