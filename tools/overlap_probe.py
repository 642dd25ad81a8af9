"""Probe (not product): do two independent batches on two streams overlap K1 with K0/K3?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, flake_amd
p = flake_amd.level_params(5, order_method=flake_amd.OM_MAX)
n, nframes = 4096, 4096
dev = torch.device("cuda", 0)
pcm = torch.from_numpy(flake_amd.synth_pcm(nframes, n, 2, 16)).to(dev)
nsub = nframes * 2
slot = flake_amd.rice_slot_bytes(p, n)
def mk():
    e = flake_amd.Encoder(p, max_frames=nframes)
    st = torch.cuda.Stream(dev)
    e.set_stream(st.cuda_stream)
    info = torch.zeros(nsub * flake_amd.INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    bits = torch.zeros(nsub * slot, dtype=torch.uint8, device=dev)
    return e, st, info, bits
for nstreams in (1, 2, 3):
    ctxs = [mk() for _ in range(nstreams)]
    def step(i):
        e, st, info, bits = ctxs[i % nstreams]
        e.encode_subframes_dev(pcm, nframes, n, info, rice_bits=bits, slot_bytes=slot)
    for i in range(6): step(i)
    torch.cuda.synchronize()
    K = 60
    t0 = time.perf_counter()
    for i in range(K): step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(nstreams, "streams:", round(dt / K * 1e6, 1), "us/step", round(nframes * n * 2 * K / dt / 1e9, 1), "Gsamples/s", flush=True)
    for c in ctxs: c[0].close()
