"""Diagnostic (not part of the product): the host path (flake_amd_encode_frames, 4096 stereo frames, MD5 off)
with the caller's buffers pageable / page-locked in place, over chunk sizes.  python tools/host_pin_probe.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, flake_amd
nfr, n = 4096, 4096
pcm = flake_amd.synth_pcm(nfr, n, 2, 16)
flat = np.ascontiguousarray(pcm, dtype=np.int32).reshape(-1, 2)
cap = 64 + pcm.size * 5
out = np.ones(cap, dtype=np.uint8)
sizes = np.zeros(nfr, dtype=np.int32)
os.environ["FLAKE_AMD_MD5"] = "0"
os.environ["FLAKE_AMD_BATCH"] = str(nfr)
for chunk in (1024, 512, 2048, 0):
    os.environ["FLAKE_AMD_CHUNK"] = str(chunk)
    for pin in (False, True):
        enc = flake_amd.HostEncoder(level=5, channels=2, bits_per_sample=16, sample_rate=44100, block_size=n, order_method=flake_amd.OM_MAX)
        if pin:
            enc.lib.flake_amd_pin_buffers(C.byref(enc.ctx), flat.ctypes.data, flat.nbytes, out.ctypes.data, flat.nbytes // 2)
        ts = []
        for call in range(6):
            t0 = time.perf_counter()
            w = enc.lib.flake_amd_encode_frames(C.byref(enc.ctx), flat.ctypes.data, nfr, n, 0, out.ctypes.data, cap, sizes.ctypes.data)
            ts.append((time.perf_counter() - t0) * 1e3)
            assert w > 0
        enc.close()
        print(f"chunk {chunk:5d} pinned {pin!s:5s}: " + " ".join(f"{t:.2f}" for t in ts), flush=True)
