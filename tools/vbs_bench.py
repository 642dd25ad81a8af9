"""End-to-end rate of the host C layer on the VBS presets (levels 9-12): blocks in host memory ->
FLAC frames in host memory (split, ragged batches, frames).  python tools/vbs_bench.py [blocks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, flake_amd

nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
os.environ["FLAKE_AMD_BATCH"] = str(nblk)
os.environ.setdefault("FLAKE_AMD_MD5", "0")
for level in (9, 10, 12):
    enc = flake_amd.HostEncoder(level=level, channels=2, bits_per_sample=16, sample_rate=44100)
    n = enc.params().block_size
    pcm = flake_amd.synth_pcm(nblk, n, 2, 16)
    # a transient in every third block so that the splitter has something to split
    pcm[::3, n // 2:, :] //= 16
    flat = np.ascontiguousarray(pcm, dtype=np.int32).reshape(-1, 2)
    cap = 64 + pcm.size * 5 + 64 * (nblk + 1) * 8
    out = np.ones(cap, dtype=np.uint8)
    sizes = np.zeros(nblk, dtype=np.int32)
    for call in range(3):
        t0 = time.perf_counter()
        w = enc.lib.flake_amd_encode_frames(C.byref(enc.ctx), flat.ctypes.data, nblk, n, 0, out.ctypes.data, cap, sizes.ctypes.data)
        dt = time.perf_counter() - t0
        assert w > 0
        print(f"level {level} n={n} call {call}: {nblk} blocks, {w} bytes, {dt * 1e3:.1f} ms, {nblk * n * 2 / dt / 1e6:.0f} Msamples/s", flush=True)
    enc.close()
