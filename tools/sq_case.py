"""Measurement probe (not part of the product): a few steps of one named workload, for a rocprofv3
pass (--kernel-trace --stats, or --pmc) over its kernels alone.
python tools/sq_case.py c0|c1|l2|l8|c2|c3|l10|l12|l10x8|l12x8|ahead [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, flake_amd

K = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def steps(p, nframes, k=K, ahead=False):
    n = p.block_size
    dev = torch.device("cuda", 0)
    pcms = [torch.from_numpy(flake_amd.synth_pcm(nframes, n, p.channels, p.bits_per_sample, first_frame=q * nframes)).to(dev)
            for q in range(2 if ahead else 1)]
    nsub = nframes * p.channels
    slot = flake_amd.rice_slot_bytes(p, n)
    info = torch.zeros(nsub * flake_amd.INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    bb = torch.zeros(nsub * slot, dtype=torch.uint8, device=dev)
    enc = flake_amd.Encoder(p, max_frames=nframes)
    st = torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    enc.set_stream(st.cuda_stream)
    for i in range(k):
        enc.encode_subframes_dev(pcms[i % len(pcms)], nframes, n, info, rice_bits=bb, slot_bytes=slot)
        if ahead:
            enc.prepare_ahead(pcms[(i + 1) % 2], nframes, n)
    enc.sync()
    enc.close()


def vbs(level, nblk, k=K):
    p = flake_amd.level_params(level)
    n = p.block_size
    dev = torch.device("cuda", 0)
    pcm = flake_amd.synth_pcm(nblk, n, 2, 16)
    pcm[::3, n // 2:, :] //= 16          # a transient in every third block: something to split
    d_pcm = torch.from_numpy(pcm).to(dev)
    cap = pcm.size * 5
    packed = torch.zeros(cap, dtype=torch.uint8, device=dev)
    totals = torch.zeros(4, dtype=torch.int64, device=dev)
    enc = flake_amd.Encoder(p, max_frames=8 * nblk)
    st = torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    enc.set_stream(st.cuda_stream)
    for _ in range(k):
        enc.encode_blocks_vbs_dev(d_pcm, nblk, n, packed, cap, totals)
    enc.sync()
    enc.close()


P = flake_amd.level_params
case = sys.argv[1]
if case == "c0":
    steps(P(2, block_size=4096, channels=1), 8192)
elif case == "c1":
    steps(P(5, order_method=flake_amd.OM_MAX), 4096)
elif case == "ahead":
    steps(P(5, order_method=flake_amd.OM_MAX), 4096, ahead=True)
elif case == "l2":
    p = P(2); steps(p, 4096 * 4096 // p.block_size)
elif case == "l8":
    steps(P(8), 4096)
elif case == "c2":
    steps(P(5, bits_per_sample=24, sample_rate=96000, order_method=flake_amd.OM_SEARCH, max_prediction_order=32, max_partition_order=8), 4096)
elif case == "c3":
    steps(P(5, channels=8, bits_per_sample=24, sample_rate=192000, order_method=flake_amd.OM_MAX, max_prediction_order=12), 4096)
elif case in ("l10", "l12"):
    vbs(int(case[1:]), 1024)
elif case in ("l10x8", "l12x8"):
    vbs(int(case[1:3]), 8192)
else:
    raise SystemExit("case?")
