"""Measurement probe (not part of the product): a few steps of one named workload, for a rocprofv3 --pmc
pass over its kernels.  python tools/sq_case.py c0|l2|l8|c2|c3"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, flake_amd

def steps(p, nframes, k=3):
    n = p.block_size
    dev = torch.device("cuda", 0)
    pcm = torch.from_numpy(flake_amd.synth_pcm(nframes, n, p.channels, p.bits_per_sample)).to(dev)
    nsub = nframes * p.channels
    slot = flake_amd.rice_slot_bytes(p, n)
    info = torch.zeros(nsub * flake_amd.INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    bb = torch.zeros(nsub * slot, dtype=torch.uint8, device=dev)
    enc = flake_amd.Encoder(p, max_frames=nframes)
    enc.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(k):
        enc.encode_subframes_dev(pcm, nframes, n, info, rice_bits=bb, slot_bytes=slot)
    enc.sync()
    enc.close()

P = flake_amd.level_params
case = sys.argv[1]
if case == "c0":
    steps(P(2, block_size=4096, channels=1), 8192)
elif case == "l2":
    p = P(2); steps(p, 4096 * 4096 // p.block_size)
elif case == "l8":
    steps(P(8), 4096)
elif case == "c2":
    steps(P(5, bits_per_sample=24, order_method=flake_amd.OM_SEARCH, max_prediction_order=32, max_partition_order=8), 4096)
elif case == "c3":
    steps(P(5, channels=8, bits_per_sample=24, order_method=flake_amd.OM_MAX, max_prediction_order=12), 4096)
else:
    raise SystemExit("case?")
