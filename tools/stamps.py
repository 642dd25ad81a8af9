"""Diagnostic (not part of the product): phase stamps of k_encode_pow2, workgroup 0.
Build: python -c "from flake_amd.build import build_hip; build_hip(True, ['-DFHIP_STAMPS'], 'libflakehip_dbg.so')"; run on the GPU box."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, flake_amd
flake_amd.LIB_DIR = flake_amd.LIB_DIR  # same dir
import flake_amd as fa
path = os.path.join(fa.LIB_DIR, "libflakehip_dbg.so")
# swap the library the package loads
orig = os.path.join(fa.LIB_DIR, "libflakehip.so")
fa._lib = None
_real_exists = os.path.exists
lib = C.CDLL(path)
import types
def _load():
    return lib
# bind signatures by loading through the package with the path patched
os.environ["FHIP_LIB"] = path
fa_load = fa.load_library
def patched():
    if fa._lib is None:
        import shutil
        fa._lib = None
    return fa_load()
p = fa.level_params(5, order_method=fa.OM_MAX)
if len(sys.argv) > 1 and sys.argv[1] == "level7":
    p = fa.level_params(7)
if len(sys.argv) > 1 and sys.argv[1] == "search":
    p = fa.level_params(5, bits_per_sample=24, order_method=fa.OM_SEARCH, max_prediction_order=32, max_partition_order=8)
nch = 2
if len(sys.argv) > 1 and sys.argv[1] == "fixed":          # configs[0] on the GPU: mono, fixed orders 0..4
    p = fa.level_params(2, block_size=4096, channels=1); nch = 1
n = 4096
if len(sys.argv) > 1 and sys.argv[1] == "level2":         # stereo, n 1152, fixed orders 0..4
    p = fa.level_params(2); n = 1152
pcm = fa.synth_pcm(2048, n, nch, p.bits_per_sample)
enc = fa.Encoder(p, max_frames=2048)
for _ in range(3):
    out = enc.encode_subframes(pcm, n, want_residual=False)
st = (C.c_longlong * 64)()
rc = fa.load_library().fhip_debug_read_stamps(st)
v = np.array(st[:13], dtype=np.int64)
names = ["load+const", "->coef", "fir", "pyramid(shfl)", "upper levels", "k-search", "select", "ret", "->emit", "lens+scan", "emit ORs", "flush+end"]
print("rc", rc)
prev = v[0]
for i in range(1, 13):
    print(f"{i:2d} {names[i-1]:16s} +{int(v[i]-prev):7d}   (abs {int(v[i]-v[0])})")
    prev = v[i]
