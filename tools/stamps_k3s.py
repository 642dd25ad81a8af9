"""Diagnostic (not part of the product): phase stamps of k_order_search, workgroup 0 (wave 0), per round.
Build: python -c "from flake_amd.build import build_hip; build_hip(True, ['-DFHIP_STAMPS'], 'libflakehip_dbg.so')"; run on the GPU box.
python tools/stamps_k3s.py [level] [block size]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flake_amd as fa
os.environ["FHIP_LIB"] = os.path.join(fa.LIB_DIR, "libflakehip_dbg.so")
level = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = fa.level_params(level, variable_block_size=0) if level >= 9 else fa.level_params(level)
if len(sys.argv) > 2:
    p = fa.level_params(level, variable_block_size=0, block_size=int(sys.argv[2]))
nfr = 2048 * max(1, 4096 // p.block_size)
pcm = fa.synth_pcm(nfr, p.block_size, 2, 16)
enc = fa.Encoder(p, max_frames=nfr)
for _ in range(3):
    enc.encode_subframes(pcm, p.block_size, want_residual=False)
st = (C.c_longlong * 64)()
lib = fa.load_library()
lib.fhip_debug_read_stamps_k3s.argtypes = [C.POINTER(C.c_longlong)]
print("rc", lib.fhip_debug_read_stamps_k3s(st))
v = np.array(st[:32], dtype=np.int64)
print("prologue (load, stage, flags):", v[1] - v[0])
names = ["plan", "stage rows+barrier", "FIR (this wave)", "barrier (all FIRs)", "Rice (this wave)", "barrier (round end)"]
prev = v[1]
for r in range(4):
    b = 2 + 6 * r
    if v[b] <= prev: break
    seq = [v[b], v[b + 1], v[b + 2], v[b + 3], v[b + 4], v[b + 5]]
    print(f"round {r}: since previous {seq[0] - prev:6d} |", "  ".join(f"{n} {int(seq[i + 1] - seq[i])}" for i, n in enumerate(names[1:])))
    prev = seq[5]
print("replay+exit:", v[30] - prev, " winner write:", v[31] - v[30], " total:", v[31] - v[0])
