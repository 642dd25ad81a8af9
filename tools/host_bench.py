"""End-to-end rate of the host C layer (not the roofline number): PCM in host memory
-> complete FLAC frames in host memory, through flake_amd_encode_frames: H2D, the
five kernels, D2H of the assembled frames, frame copy-out and the stream MD5.
    python tools/host_bench.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, flake_amd

nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = 4096
pcm = flake_amd.synth_pcm(nfr, n, 2, 16)
for batch in (1024, 4096):
    os.environ["FLAKE_AMD_BATCH"] = str(batch)
    enc = flake_amd.HostEncoder(level=5, channels=2, bits_per_sample=16, sample_rate=44100, block_size=n, order_method=flake_amd.OM_MAX)
    # the first call pays the one-time costs (device buffers, code objects, first touch of
    # the host staging pages); a stream of batches runs at the rate of the later calls
    import ctypes as C
    ch = 2
    cap = 64 + pcm.size * 5 + 64 * (nfr + 1) * 8
    out = np.ones(cap, dtype=np.uint8)                   # touched: no page faults in the timed calls
    sizes = np.zeros(nfr, dtype=np.int32)
    flat = np.ascontiguousarray(pcm, dtype=np.int32).reshape(-1, ch)
    for call in range(4):
        t0 = time.perf_counter()
        w = enc.lib.flake_amd_encode_frames(C.byref(enc.ctx), flat.ctypes.data, nfr, n, 0, out.ctypes.data, cap, sizes.ctypes.data)
        dt = time.perf_counter() - t0
        assert w > 0
        print(f"batch {batch} call {call}: {nfr} frames, {w} bytes, {dt * 1e3:.1f} ms, {nfr * n * 2 / dt / 1e6:.0f} Msamples/s "
              f"(ratio {w / (nfr * n * 4):.3f})", flush=True)
    enc.close()
