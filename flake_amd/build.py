"""Build recipes for the native parts (run here without a GPU: hipcc cross-compiles).

    python -m flake_amd.build            # everything
    python -m flake_amd.build hip host   # selected targets

Targets
  hip   flake_amd/lib/libflakehip.so   gfx950 kernels + C ABI (include/flakehip.h)
  host  flake_amd/lib/libflake_amd.so  host C layer (include/flake_amd.h) on top of it
  probes  the K1 timing-probe libraries tools/k1_sq.sh and tools/k1run.sh load (libflakehip_noprod.so,
          _nowalk.so, _nob.so: -DFHIP_PROBE_*; their RESULTS ARE WRONG, measurements only -- never
          built by default, never loaded by the package unless FHIP_LIB points at one)
"""
from __future__ import annotations

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "flake_amd")
LIB = os.path.join(PKG, "lib")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

HIP_SRCS = ["csrc/k0_prepare.hip", "csrc/k1_autocorr.hip", "csrc/k2_lpc.hip", "csrc/k3_encode.hip", "csrc/k3s_search.hip",
            "csrc/k4_assemble.hip", "csrc/api.hip"]
HIP_HDRS = ["csrc/kernels.h", "csrc/device_util.h", "csrc/lpc_reg.h", "csrc/k3_common.h", "../include/flakehip.h"]
HIP_DEPS = HIP_SRCS + HIP_HDRS
HOST_SRCS = ["host/flake_host.c", "host/synth.c", "host/md5.c"]
HOST_DEPS = HOST_SRCS + ["host/host_internal.h", "../include/flakehip.h", "../include/flake_amd.h"]

# -ffp-contract=off is load-bearing: the fp64 LPC stages must round after every
# multiply and add to reproduce the reference bit for bit.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared",
             "-std=c++17", "-fvisibility=hidden", "-fgpu-rdc" if False else ""]
HIP_FLAGS = [f for f in HIP_FLAGS if f]


def _stale(out: str, deps: list[str]) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.exists(os.path.join(PKG, d)) and os.path.getmtime(os.path.join(PKG, d)) > t
               for d in deps)


def _run(cmd: list[str]) -> None:
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build_hip(force: bool = False, extra: list[str] | None = None, out_name: str = "libflakehip.so") -> str:
    """One object per kernel file, compiled side by side, then one link."""
    out = os.path.join(LIB, out_name)
    if force or _stale(out, HIP_DEPS):
        os.makedirs(LIB, exist_ok=True)
        objdir = os.path.join(PKG, "build", os.path.splitext(out_name)[0])
        os.makedirs(objdir, exist_ok=True)
        cflags = [f for f in HIP_FLAGS if f != "-shared"]
        hdr_t = max(os.path.getmtime(os.path.join(PKG, h)) for h in HIP_HDRS)
        jobs = []
        for s in HIP_SRCS:
            src = os.path.join(PKG, s)
            obj = os.path.join(objdir, os.path.basename(s) + ".o")
            if (force or extra or not os.path.exists(obj)
                    or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t)):
                cmd = [HIPCC, *cflags, *(extra or []), "-I", os.path.join(ROOT, "include"),
                       "-I", os.path.join(PKG, "csrc"), "-c", src, "-o", obj]
                print("+", " ".join(cmd), flush=True)
                jobs.append((cmd, subprocess.Popen(cmd)))
        for cmd, proc in jobs:
            if proc.wait() != 0:
                raise subprocess.CalledProcessError(proc.returncode, cmd)
        objs = [os.path.join(objdir, os.path.basename(s) + ".o") for s in HIP_SRCS]
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
    return out


def build_host(force: bool = False) -> str:
    out = os.path.join(LIB, "libflake_amd.so")
    srcs = [os.path.join(PKG, s) for s in HOST_SRCS if os.path.exists(os.path.join(PKG, s))]
    if not srcs:
        return ""
    if force or _stale(out, HOST_DEPS):
        os.makedirs(LIB, exist_ok=True)
        _run(["gcc", "-std=gnu99", "-O2", "-Wall", "-fPIC", "-shared", "-fvisibility=hidden",
              "-I", os.path.join(ROOT, "include"), *srcs, "-o", out,
              "-L", LIB, "-lflakehip", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"])
    return out


def build_flake_names(force: bool = False) -> str:
    """libflake.so: the same host layer exporting libflake's own symbol names
    (flake_set_defaults ... flake_encode_close), a link-level drop-in."""
    out = os.path.join(LIB, "libflake.so")
    srcs = [os.path.join(PKG, s) for s in HOST_SRCS if os.path.exists(os.path.join(PKG, s))]
    if force or _stale(out, HOST_DEPS):
        _run(["gcc", "-std=gnu99", "-O2", "-Wall", "-fPIC", "-shared", "-fvisibility=hidden",
              "-DFLAKE_AMD_EXPORT_FLAKE_NAMES", "-I", os.path.join(ROOT, "include"), *srcs, "-o", out,
              "-L", LIB, "-lflakehip", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"])
    return out


def build_cli(force: bool = False) -> str:
    """The C command-line harness on the host API (flake/flake.c's block loop)."""
    out = os.path.join(LIB, "flake_amd_cli")
    src = os.path.join(PKG, "host/flake_amd_cli.c")
    if not os.path.exists(src):
        return ""
    if force or _stale(out, ["host/flake_amd_cli.c", "../include/flake_amd.h"]):
        _run(["gcc", "-std=gnu99", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", out,
              "-L", LIB, "-lflake_amd", "-lflakehip", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"])
    return out


def build_all(force: bool = False) -> None:
    build_hip(force)
    build_host(force)
    build_flake_names(force)
    build_cli(force)


PROBE_LIBS = {"libflakehip_noprod.so": "-DFHIP_PROBE_NOPROD", "libflakehip_nowalk.so": "-DFHIP_PROBE_NOWALK",
              "libflakehip_nob.so": "-DFHIP_PROBE_NOB"}


def build_probes() -> None:
    for name, flag in PROBE_LIBS.items():
        build_hip(force=True, extra=[flag], out_name=name)


if __name__ == "__main__":
    targets = sys.argv[1:] or ["hip", "host"]
    if "probes" in targets:
        build_probes()
    if "hip" in targets:
        build_hip(force=True, extra=["-Rpass-analysis=kernel-resource-usage"] if os.environ.get("FHIP_REMARKS") else None)
    if "host" in targets:
        build_host(force=True)
        build_flake_names(force=True)
        build_cli(force=True)
