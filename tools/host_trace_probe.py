import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, flake_amd
nfr, n = 4096, 4096
pcm = flake_amd.synth_pcm(nfr, n, 2, 16)
flat = np.ascontiguousarray(pcm, dtype=np.int32).reshape(-1, 2)
cap = 64 + pcm.size * 5
out = np.ones(cap, dtype=np.uint8)
sizes = np.zeros(nfr, dtype=np.int32)
os.environ["FLAKE_AMD_MD5"] = "0"; os.environ["FLAKE_AMD_BATCH"] = str(nfr); os.environ["FLAKE_AMD_TRACE"] = "1"
enc = flake_amd.HostEncoder(level=5, channels=2, bits_per_sample=16, sample_rate=44100, block_size=n, order_method=flake_amd.OM_MAX)
for call in range(4):
    print("call", call, file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    w = enc.lib.flake_amd_encode_frames(C.byref(enc.ctx), flat.ctypes.data, nfr, n, 0, out.ctypes.data, cap, sizes.ctypes.data)
    print("  total %.3f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
enc.close()
