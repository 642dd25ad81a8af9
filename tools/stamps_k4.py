"""Diagnostic (not part of the product): phase stamps of k_assemble, workgroup 0, inside a level-10 ragged batch.
Build: python -c "from flake_amd.build import build_hip; build_hip(True, ['-DFHIP_STAMPS'], 'libflakehip_dbg.so')"; run on the GPU box."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import flake_amd as fa
os.environ["FHIP_LIB"] = os.path.join(fa.LIB_DIR, "libflakehip_dbg.so")
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
p = fa.level_params(10)
n = p.block_size
dev = torch.device("cuda", 0)
pcm = fa.synth_pcm(nblk, n, 2, 16)
pcm[::3, n // 2:, :] //= 16
d_pcm = torch.from_numpy(pcm).to(dev)
cap = pcm.size * 5
packed = torch.zeros(cap, dtype=torch.uint8, device=dev)
totals = torch.zeros(4, dtype=torch.int64, device=dev)
enc = fa.Encoder(p, max_frames=8 * nblk)
for _ in range(3):
    enc.encode_blocks_vbs_dev(d_pcm, nblk, n, packed, cap, totals)
enc.sync()
st = (C.c_longlong * 64)()
rc = fa.load_library().fhip_debug_read_stamps_k4(st)
v = np.array(st[:7], dtype=np.int64)
names = ["loads issued", "tables", "sums, header", "prefixes", "quads", "-", "crc to the end, bytes"]
print("rc", rc, "(s_memtime ticks: 100 MHz)")
prev = v[0]
for i in (1, 2, 3, 4, 6):
    print(f"{i} {names[i]:22s} +{int(v[i]-prev):7d}   (abs {int(v[i]-v[0])})")
    prev = v[i]
