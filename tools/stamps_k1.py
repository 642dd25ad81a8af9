"""Diagnostic (not part of the product): where the first wave pair of k_autocorr_wt
spends its cycles.  Needs the stamped build:
  python -c "from flake_amd.build import build_hip; build_hip(True, ['-DFHIP_STAMPS'], 'libflakehip_dbg.so')"
  FHIP_LIB=flake_amd/lib/libflakehip_dbg.so python tools/stamps_k1.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, flake_amd as fa
p = fa.level_params(5, order_method=fa.OM_MAX)
n = 4096; nfr = 4096
dev = torch.device("cuda", 0)
pcm = torch.from_numpy(fa.synth_pcm(nfr, n, 2, 16)).to(dev)
info = torch.zeros(nfr * 2 * fa.INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
enc = fa.Encoder(p, max_frames=nfr)
enc.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    enc.encode_subframes_dev(pcm, nfr, n, info)
enc.sync()
st = (C.c_longlong * 64)()
fa.load_library().fhip_debug_read_stamps(st)
names = {40: "consumer: barrier wait", 41: "consumer: walk (tiles 1..)", 42: "consumer: tile 0 (head + walk)",
         44: "producer: wait loads + window + LDS writes", 45: "producer: issue loads", 46: "producer: barrier wait"}
for k in sorted(names):
    print(f"{names[k]:46s} {st[k]:9d} cycles")
