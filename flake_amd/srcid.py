"""Identity of the kernel sources: a SHA-1 over flake_amd/csrc and include/flakehip.h.  The
PMC summary under profiles/ records the one it was measured with, bench.py reports whether
the library it times still is that code (the traffic figure must not go stale
silently)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha1() -> str:
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, "flake_amd", "csrc", "*")))
    files.append(os.path.join(ROOT, "include", "flakehip.h"))
    for p in files:
        if os.path.isfile(p):
            h.update(os.path.basename(p).encode())
            h.update(open(p, "rb").read())
    return h.hexdigest()
