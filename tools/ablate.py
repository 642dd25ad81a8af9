"""Timing probe (not part of the product): per-kernel times of the C2 workload
with outputs switched on/off.  python tools/ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, flake_amd

def run(tag, p, nframes=4096, bits=True, resid=False, steps=10):
    n = p.block_size
    dev = torch.device("cuda", 0)
    pcm = torch.from_numpy(flake_amd.synth_pcm(nframes, n, p.channels, p.bits_per_sample)).to(dev)
    nsub = nframes * p.channels
    slot = flake_amd.rice_slot_bytes(p, n)
    info = torch.zeros(nsub * flake_amd.INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    bb = torch.zeros(nsub * slot, dtype=torch.uint8, device=dev) if bits else None
    rr = torch.zeros((nsub, n), dtype=torch.int32, device=dev) if resid else None
    enc = flake_amd.Encoder(p, max_frames=nframes)
    enc.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        enc.encode_subframes_dev(pcm, nframes, n, info, residual=rr, rice_bits=bb, slot_bytes=slot)
    enc.sync()
    enc.set_profiling(True); enc.kernel_times(reset=True)
    for _ in range(steps):
        enc.encode_subframes_dev(pcm, nframes, n, info, residual=rr, rice_bits=bb, slot_bytes=slot)
    enc.sync()
    kt = enc.kernel_times()
    print(tag, {k: round(ms / max(c, 1) * 1e3, 1) for k, (ms, c) in kt.items()}, flush=True)
    enc.close()

if __name__ == "__main__":
    P = flake_amd.level_params
    c2 = P(5, order_method=flake_amd.OM_MAX)
    run("c2 bits", c2)
    if "--quick" in sys.argv:
        sys.exit(0)
    run("c2 nobits", c2, bits=False)
    run("c2 bits+resid", c2, resid=True)
    for bs in (1152, 4608, 576, 256):
        run(f"lvl5 n={bs}", P(5, block_size=bs, order_method=flake_amd.OM_MAX), nframes=4096 * 4096 // bs)
    run("c2 fixed", P(2, block_size=4096))
    run("c2 est", P(5))
    run("lvl8 log12", P(8), nframes=1024)
    run("search32 24bit", P(5, bits_per_sample=24, order_method=flake_amd.OM_SEARCH, max_prediction_order=32, max_partition_order=8), nframes=512)
    run("8ch lpc12", P(5, channels=8, bits_per_sample=24, order_method=flake_amd.OM_MAX, max_prediction_order=12), nframes=1024)
