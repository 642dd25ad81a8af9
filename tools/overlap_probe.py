"""Diagnostic (not part of the product): do consecutive batches overlap when two
encoders alternate on two streams?  Settled clocks, same total work as bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, flake_amd as fa
dev = torch.device("cuda", 0)
p = fa.level_params(5, channels=2, bits_per_sample=16, sample_rate=44100, order_method=fa.OM_MAX)
n = p.block_size
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NE = int(sys.argv[2]) if len(sys.argv) > 2 else 2
slot = fa.rice_slot_bytes(p, n)
pcm = torch.from_numpy(fa.synth_pcm(nfr, n, 2, 16)).to(dev)
ib = fa.INFO_DTYPE.itemsize
encs = []
for i in range(NE):
    e = fa.Encoder(p, max_frames=nfr, device=0)
    st = torch.cuda.Stream(dev)
    e.set_stream(st.cuda_stream)
    info = torch.zeros(nfr * 2 * ib, dtype=torch.uint8, device=dev)
    bits = torch.zeros(nfr * 2 * slot, dtype=torch.uint8, device=dev)
    encs.append((e, st, info, bits))
torch.cuda.synchronize()
def step(i):
    e, st, info, bits = encs[i % NE]
    e.encode_subframes_dev(pcm, nfr, n, info, rice_bits=bits, slot_bytes=slot)
def run(steps, ne):
    for i in range(steps):
        e, st, info, bits = encs[i % ne]
        e.encode_subframes_dev(pcm, nfr, n, info, rice_bits=bits, slot_bytes=slot)
    torch.cuda.synchronize()
for ne in (1, NE, 1, NE):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.06:
        run(10, ne)
    run(150, ne)
    t0 = time.perf_counter(); run(400, ne); dt = time.perf_counter() - t0
    print(f"encoders/streams={ne}: {dt / 400 * 1e3:.4f} ms per batch of {nfr} frames, {nfr * n * 2 * 400 / dt / 1e9:.1f} Gsamples/s")
