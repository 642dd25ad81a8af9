#!/usr/bin/env python3
"""bench.py -- Msamples/s of the FLAC prediction/entropy hot path on MI355X.

    python bench.py --gpus N --steps 400 --warmup 150          # any N: for N > 1 the
        # parent (which never touches the GPU) starts N child ranks and relays rank 0's line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

One *step* is one pass of the hot path (K0 prepare -> K1 autocorrelation ->
K2 Levinson/quantise -> K3 residual + Rice search + Rice bit emit) over one
batch of synthetic PCM that is already resident in HBM: BASELINE.json
configs[1] -- stereo 16-bit 44.1 kHz, block size 4096, LPC order 8 (level-5
parameters with the MAX order method), 4096 frames per GPU.  Frames are
independent, so with N ranks each rank encodes its own shard; the only collective
is the final all-reduce of {frames, residual bits} (RCCL), once per job, inside
the timed region.

  --scaling weak    (default, what the driver runs) 4096 frames per rank: a 4096*N-frame job
  --scaling strong  ONE job of --frames frames cut into contiguous shards (flake_amd.shard.
                    shard_range): the per-GPU batch shrinks as N grows -- BASELINE configs[3]'s
                    "one batch sharded over 1/2/4/8 GPUs"

Prints ONE JSON line on rank 0 (see the keys in main()).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
SIMDS = 1024                     # 256 CUs x 4
CLOCK_HZ = 2.4e9                 # max clock: issue bounds below are the least time the work can take
# A gfx950 SIMD is 32 lanes wide (MI355X_MICROARCH.md): with several waves resident it issues a wave64 vector instruction
# of the simple classes (v_add_u32, v_sub_u32, v_ashrrev, and / or / xor, mov, fp32) every 2 cycles and one of the
# half-rate classes (three-operand integer forms, v_lshlrev, min / max, compares, cndmask, carries, multiplies, dot
# products, alignbyte / perm, DPP forms, lane reads, every fp64 instruction) every 4 -- measured with
# tools/ubench_valu.hip, profiles/r04_ubench_valu.txt.  A kernel's price per vector instruction is its own static mix of
# the two (tools/valu_mix.py -> profiles/valu_mix.json); without one it is the optimistic 2.  (Rounds 1-3 priced every
# vector instruction at 4 cycles: their valu_issue fractions read 20-35 % too high.)
ISSUE_CYCLES_FULL = 2
ISSUE_CYCLES = 4                 # the half-rate classes; fp64 (K1, K2) is all of this class


def load_valu_mix():
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "valu_mix.json")))
        from flake_amd.srcid import kernel_sources_sha1
        tj["_current"] = (tj.get("_src_sha1") == kernel_sources_sha1())
        return tj
    except Exception:
        return {}


_VALU_MIX = None


def valu_cycles(symbol_fragment):
    """Average issue cycles per vector instruction of the kernel instance whose name contains the fragment."""
    global _VALU_MIX
    if _VALU_MIX is None:
        _VALU_MIX = load_valu_mix()
    for k, v in _VALU_MIX.items():
        if not k.startswith("_") and symbol_fragment and k.startswith(symbol_fragment):
            return float(v["avg_cycles"])
    return float(ISSUE_CYCLES_FULL)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # Defaults: the GPU's clocks need ~100 steps (20 ms) of this load to settle
    # (tools/ramp.py: 0.206 ms/step in the first 20 steps, 0.179 from step ~120 on),
    # so the warm-up covers that and the timed region is long enough to average.
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=150)
    ap.add_argument("--settle-ms", type=float, default=40.0,
                    help="untimed steps before the warm-up until this much wall time has passed: "
                         "the clocks ramp for ~20 ms under this load whatever --warmup says")
    ap.add_argument("--frames", type=int, default=4096,
                    help="frames per GPU per step (weak scaling) or of the whole job (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--workload", choices=["configs[1]", "configs[3]"], default="configs[1]",
                    help="configs[1] is the one BASELINE.json's metric is quoted on (the default "
                         "and the only one the driver runs); configs[3] (8-channel 24-bit LPC-12, "
                         "the config BASELINE.json shards over 1/2/4/8 GPUs) runs through the same "
                         "timed region and sharding for a scaling run of that shape")
    ap.add_argument("--ahead", action="store_true",
                    help="hint the next batch's feeder stage ahead (fhip_prepare_ahead) so that it "
                         "runs beside the kernels in flight; measured SLOWER on MI355X (DESIGN.md: "
                         "the kernels contend, 0.163-0.194 vs 0.151 ms/step), hence opt-in")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="rough budget of CPU work for the cpu_baseline leg")
    ap.add_argument("--profile-steps", type=int, default=200,
                    help="extra steps with per-kernel hipEvent timing for the roofline object")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the per-kernel timings of the other BASELINE configs, the small-batch "
                         "steps and the host path that rank 0 adds at N = 1")
    ap.add_argument("--other-steps", type=int, default=20)
    ap.add_argument("--with-residual", action="store_true",
                    help="also write the int32 residual (stage A of SURVEY 8d)")
    ap.add_argument("--host-path-only", action="store_true",
                    help="print host_path()'s JSON and exit (the main run starts this as a child process)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------
def workload(name):
    import flake_amd
    if name == "configs[3]":
        p = flake_amd.level_params(5, channels=8, bits_per_sample=24, sample_rate=192000,
                                   order_method=flake_amd.OM_MAX, max_prediction_order=12)
        return p, "Msamples/s encoded, 8-channel 24-bit 192k blocksize 4096 LPC-12", \
            ("configs[3]: 8-channel 24-bit 192 kHz, blocksize 4096, LPC-12 (level-5 params, "
             "order method MAX, max order 12, partition orders 0-5), synthetic resonator PCM "
             "resident in HBM")
    p = flake_amd.level_params(5, channels=2, bits_per_sample=16, sample_rate=44100,
                               order_method=flake_amd.OM_MAX)
    return p, "Msamples/s encoded, 16-bit stereo 44.1k blocksize 4096 LPC-8", \
        ("configs[1]: stereo 16-bit 44.1 kHz, blocksize 4096, LPC max order 8 "
         "(level-5 params, order method MAX, partition orders 0-5, stereo "
         "estimate), synthetic resonator PCM resident in HBM")


def load_pmc():
    """profiles/pmc_traffic.json: HBM bytes and SQ instruction counts per launch of the headline
    kernels (rocprofv3 --pmc passes, tools/prof_round.sh); counters cannot be read from inside
    this process, so the figures belong to the code state named there (`traffic_current`)."""
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        tj = json.load(open(tp))
        from flake_amd.srcid import kernel_sources_sha1
        tj["_current"] = (tj.get("_src_sha1") == kernel_sources_sha1())
        return tj
    except Exception:
        return None


def load_pmc_cases():
    """profiles/pmc_cases.json: SQ counters per wave of every kernel instance of the other_configs workloads
    (tools/r04_measure.sh -> tools/r04_collect.py); `_current` says whether they belong to this source tree."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_cases.json")))
        from flake_amd.srcid import kernel_sources_sha1
        tj["_current"] = (tj.get("_src_sha1") == kernel_sources_sha1())
        return tj
    except Exception:
        return None


MFMA_ISSUE_CYCLES = 8            # a 16x16x64 int8 MFMA holds its SIMD's vector issue for 8 of its 16 cycles (MI355X_MICROARCH.md)


def issue_bound(short, case_ent, ms_per_step):
    """A kernel priced by its instruction issue (the search and encode kernels are not HBM problems):
    bound time = sum over its instances of launches x waves / 1024 SIMDs x (the instance's cycles per vector
    instruction, valu_cycles() above: 2 .. 4, + 8 per matrix instruction) at 2.4 GHz; frac = bound time / measured time.  `parked` = share of wave cycles spent
    at barriers / waits (SQ_WAIT_ANY), `issue_stalled` = SQ_WAIT_INST_ANY's share."""
    inst = {k: v for k, v in case_ent.items() if short in k}
    if not inst:
        return None
    tb = vi = mi = wc = wa = wi = wv = vcyc = 0.0
    for name, v in inst.items():
        w = v["launches_per_step"] * v["waves_per_launch"]
        vi += w * v["valu_per_wave"]
        vcyc += w * v["valu_per_wave"] * valu_cycles(name)
        mi += w * v["mfma_per_wave"]
        wc += w * v["wave_cycles"]
        wa += w * v["wait_any_cycles"]
        wi += w * v["wait_inst_cycles"]
        wv += w
    tb = (vcyc + mi * MFMA_ISSUE_CYCLES) / SIMDS / CLOCK_HZ
    t = ms_per_step * 1e-3
    parked = wa / wc if wc else None
    frac = tb / t if t > 0 else None
    out = {"ms": round(ms_per_step, 4),
           "bound": "valu_issue" if (frac or 0) >= 0.5 or (parked or 0) < 0.4 else "barrier",
           "frac": round(frac, 4) if frac is not None else None,
           "valu_per_wave": round(vi / wv, 1), "cycles_per_valu": round(vcyc / vi, 3) if vi else None,
           "mfma_per_wave": round(mi / wv, 1), "waves_per_step": int(wv),
           "mfma_pipe_frac": round(mi * 16 / SIMDS / CLOCK_HZ / t, 4) if mi else 0.0,
           "parked": round(parked, 3) if parked is not None else None,
           "issue_stalled": round(wi / wc, 3) if wc else None,
           "instances": len(inst)}
    return out


def per_kernel_bounds(p, n, nframes, kernel_ms, rice_bytes, pmc, with_residual=False, case=None, cases=None):
    """What binds each kernel of the step, and how close to that bound it runs.

    hbm         least bytes the kernel must move / 8 TB/s
    fp64_issue  its un-fused fp64 operations (a multiply and an add per product: the reference's
                rounding sequence) as wave instructions x 4 cycles over 1024 SIMDs at 2.4 GHz
    valu_issue  its vector instructions per wave (SQ_INSTS_VALU / SQ_WAVES, PMC) x waves per SIMD x its
                cycles per vector instruction (2 .. 4: valu_cycles()) at 2.4 GHz
    frac = bound time / measured time."""
    import flake_amd
    ch = p.channels
    nsub = nframes * ch
    samples = nsub * n
    lags = p.max_prediction_order + 1
    out = {}
    for k, ms in kernel_ms.items():
        t = ms * 1e-3
        if k == "k_prepare":
            row_bytes = 2 if (ch == 2 and p.bits_per_sample <= 16) else 4
            b = samples * 4 + samples * row_bytes + nsub * 16
            out[k] = {"ms": round(ms, 4), "bound": "hbm", "bytes": b,
                      "frac": round(b / t / 1e9 / HBM_PEAK_GBPS, 4)}
        elif k == "k_autocorr":
            ops = samples * (2 * lags + 1)              # window multiply + a multiply and an add per lag
            tb = ops / 64 * ISSUE_CYCLES / SIMDS / CLOCK_HZ
            out[k] = {"ms": round(ms, 4), "bound": "fp64_issue", "fp64_ops": ops, "frac": round(tb / t, 4)}
        elif k == "k_lpc":
            o = p.max_prediction_order
            ops = nsub * (2 * o * o + 8 * o)
            tb = ops / 64 * ISSUE_CYCLES / SIMDS / CLOCK_HZ
            out[k] = {"ms": round(ms, 4), "bound": "fp64_issue", "fp64_ops": ops, "frac": round(tb / t, 4),
                      "note": "one lane per subframe: latency, not issue, sets its time"}
        elif cases and case and isinstance(cases.get(case), dict) and issue_bound(k, cases[case], ms):
            out[k] = issue_bound(k, cases[case], ms)
            out[k]["source"] = f"profiles/pmc_cases.json [{case}], tag {cases.get('_tag')}"
            out[k]["current"] = cases.get("_current")
        else:
            ent = (pmc or {}).get(k, {}) if pmc else {}
            vpw, waves = ent.get("valu_per_wave"), ent.get("waves_per_launch")
            if vpw and waves and ent.get("frames") == nframes and ent.get("workload") == "configs[1]" \
                    and ch == 2 and n == 4096 and p.bits_per_sample == 16:
                cyc = valu_cycles(ent.get("symbol", "") + ("<16, 256, 0>" if ent.get("symbol") == "k_encode_pow2" else ""))
                tb = vpw * (waves / SIMDS) * cyc / CLOCK_HZ
                out[k] = {"ms": round(ms, 4), "bound": "valu_issue", "valu_per_wave": vpw, "cycles_per_valu": cyc,
                          "waves_per_launch": waves, "frac": round(tb / t, 4),
                          "source": f"profiles/pmc_traffic.json, tag {pmc.get('_tag')}",
                          "current": pmc.get("_current")}
            else:
                b = samples * 4 + rice_bytes + nsub * flake_amd.INFO_DTYPE.itemsize \
                    + (samples * 4 if with_residual else 0)
                out[k] = {"ms": round(ms, 4), "bound": "hbm", "bytes": b,
                          "frac": round(b / t / 1e9 / HBM_PEAK_GBPS, 4),
                          "note": "no instruction counts on record for this shape: algorithmic bytes"}
    return out


# ---------------------------------------------------------------------------------------------
# CPU legs
# ---------------------------------------------------------------------------------------------
def _cpu_worker(job):
    """One host process of the all-cores leg: the oracle on its own copy of the sample."""
    frames, n, reps, pvals = job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib
    import flake_amd
    p = flake_amd.Params()
    for (k, _), v in zip(p._fields_, pvals):
        setattr(p, k, int(v))
    orc = oraclelib.Oracle()
    pcm = flake_amd.synth_pcm(frames, n, p.channels, p.bits_per_sample)
    slot = flake_amd.rice_slot_bytes(p, n)
    orc.encode_subframes_batch(p, pcm, n, want_residual=False, slot_bytes=slot)     # page in
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.encode_subframes_batch(p, pcm, n, want_residual=False, slot_bytes=slot)
    return time.perf_counter() - t0


def host_cores():
    """(cores used, cores available): the scheduler affinity cut to the cgroup's CPU quota where
    one is set (a GPU box hands each GPU a share of the host: 16 cores for one GPU of this pool);
    the all-cores leg uses at most BENCH_CPU_CORES (default 16) of them."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    cap = int(os.environ.get("BENCH_CPU_CORES", "16"))
    return max(1, min(n, cap)), n


REF_BENCH = os.path.join(ROOT, "build", "ref", "ref_bench")     # oracle/Makefile `refbench`: the reference's own CMake build
REF_CLI = os.path.join(ROOT, "build", "ref", "flake")


def reference_leg(params, n, frames, seconds=1.0, transient=False):
    """TIMING ONLY: the reference's own flake_encode_frame() loop (flake/flake.c:624-663) over the same
    synthetic frames, as a child process -- build/ref/ref_bench, linked against the libflake_static.a
    the reference's CMake build makes (oracle/Makefile target `refbench`; git-ignored, built where
    /root/reference exists and shipped as a binary).  Never a parity witness: nothing compares its
    bytes.  Its calls include the stream MD5 (encode.c:1006), as every libflake caller's do."""
    import subprocess
    if not os.path.exists(REF_BENCH):
        return None
    args = [REF_BENCH, params.channels, params.bits_per_sample, params.sample_rate, n, params.order_method,
            params.stereo_method, params.prediction_type, params.min_prediction_order, params.max_prediction_order,
            params.min_partition_order, params.max_partition_order, params.variable_block_size, params.allow_vbs,
            frames, seconds, 1 if transient else 0]
    try:
        r = subprocess.run([str(a) for a in args], capture_output=True, text=True, timeout=120 + 20 * seconds)
        j = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:
        return {"error": repr(e)}
    return {"value": round(j["samples"] / j["seconds"] / 1e6, 3), "unit": "Msamples/s", "cores": 1,
            "kind": "reference (timing only)",
            "sample": f"{j['frames']} calls of the reference's flake_encode_frame() on this workload's frames "
                      f"({j['samples'] / 1e6:.1f} Msamples, {j['seconds']:.2f} s; stream MD5 included, as in every call "
                      "of libflake), build/ref/ref_bench = oracle/ref_bench.c + the reference's CMake-built libflake_static.a"}


def reference_cli_leg(nframes=16384, n=4096):
    """TIMING ONLY, BASELINE configs[0] as it is defined: the reference's `flake` CLI on a mono 16-bit WAV in
    tmpfs (`flake -q -2 -b 4096 in.wav -o out.flac`), wall time of the child process (WAV parsing, encode,
    MD5, file output included)."""
    import subprocess
    import tempfile
    import wave
    import flake_amd
    if not os.path.exists(REF_CLI):
        return None
    tmp = tempfile.mkdtemp(prefix="flake_ref_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    wav, out = os.path.join(tmp, "in.wav"), os.path.join(tmp, "out.flac")
    try:
        pcm = flake_amd.synth_pcm(nframes, n, 1, 16).astype(np.int16)
        with wave.open(wav, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(44100)
            w.writeframes(pcm.tobytes())
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            r = subprocess.run([REF_CLI, "-q", "-2", "-b", str(n), wav, "-o", out], capture_output=True, timeout=300)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": f"flake CLI rc {r.returncode}: {r.stderr[-200:]!r}"}
            best = dt if best is None else min(best, dt)
        samples = nframes * n
        return {"value": round(samples / best / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                "kind": "reference (timing only)",
                "sample": f"`flake -q -2 -b {n} in.wav -o out.flac` on a {samples / 1e6:.1f} Msample mono 16-bit WAV in "
                          f"tmpfs, best of 2 runs ({best:.2f} s wall: WAV parsing, encode, MD5 and file output included), "
                          f"{os.path.getsize(out)} bytes out; build/ref/flake = the reference's CMake build"}
    except Exception as e:
        return {"error": repr(e)}
    finally:
        for f in (wav, out):
            try:
                os.remove(f)
            except OSError:
                pass
        try:
            os.rmdir(tmp)
        except OSError:
            pass


def small_cpu_baseline(params, n, frames, pcm=None, budget_s=1.0, transient=False):
    """The CPU figures every `other_configs` row carries, each over AT LEAST `budget_s` seconds of CPU work
    (round 3 timed 0.00-0.02 s here): the CPU restatement (kind "port", one thread; whole frames through the
    oracle's flake_encode_frame() -- encode_block, the VBS driver included -- when the row's outputs are frames)
    repeated over a small sample until the budget is spent, and under "reference" the reference's own loop
    (reference_leg: timing only)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib
    import flake_amd
    orc = oraclelib.Oracle()
    ch = params.channels
    if pcm is None:
        pcm = flake_amd.synth_pcm(frames, n, ch, params.bits_per_sample)
    frames = pcm.shape[0]
    t0 = time.perf_counter()
    done = 0
    passes = 0
    while True:
        if params.variable_block_size:
            fc = 0
            for b in range(frames):
                rc, _, fc = orc.encode_block(params, fc, pcm[b], n, 8 * n * ch * 4 + 4096)
                if rc <= 0:
                    raise RuntimeError("oracle encode_block failed")
                done += 1
                if passes and time.perf_counter() - t0 > budget_s:
                    break
        else:
            slot = flake_amd.rice_slot_bytes(params, n)
            step = max(1, frames // 8)
            for f0 in range(0, frames, step):
                orc.encode_subframes_batch(params, pcm[f0:f0 + step], n, want_residual=False, slot_bytes=slot)
                done += min(step, frames - f0)
                if passes and time.perf_counter() - t0 > budget_s:
                    break
        passes += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    samples = done * n * ch
    out = {"value": round(samples / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": f"{done} frames ({frames} distinct, repeated) of this workload ({samples / 1e6:.2f} Msamples, "
                     f"{dt:.2f} s), oracle/flake_oracle.c"}
    ref = reference_leg(params, n, frames, budget_s, transient)
    if ref is not None:
        out["reference"] = ref
    return out


def cpu_baseline(params, n, budget_s, gpu_bits_per_frame=None):
    """CPU legs, timed on this box's host cores.  Reported, not the target.

    value            the CPU restatement (oracle/flake_oracle.c, kind "port"), one thread, on a
                     bounded sample of the same workload: prepare + encode_residual + Rice emit.
    all_cores        the same on every host core this process may use, one process per core,
                     each on its own frames (frames are independent: the frame-parallel bound).
    reference_stages the stages of the path that compile from the reference's own sources
                     (oracle/_ref: lpc.c, rice.c, bitio.h, crc.c) timed as the reference's code,
                     one thread, on prepared subframes of the same sample; the integer FIR
                     (optimize.c, not buildable here -- DESIGN.md 4) is a plain C loop beside them.
    reference        TIMING ONLY (round 4): the reference's own flake_encode_frame() loop on the same frames
                     (reference_leg above: build/ref/ref_bench on the libflake_static.a of the reference's CMake
                     build; a child process; never a parity witness -- parity is pinned by oracle/_ref, which is
                     built without the reference's build system).  Includes the stream MD5."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib
    import flake_amd

    orc = oraclelib.Oracle()
    frames = 1024
    pcm = flake_amd.synth_pcm(frames, n, params.channels, params.bits_per_sample)
    slot = flake_amd.rice_slot_bytes(params, n)
    t0 = time.perf_counter()
    orc.encode_subframes_batch(params, pcm, n, want_residual=False, slot_bytes=slot)
    dt1 = time.perf_counter() - t0
    reps = max(1, min(400, int(budget_s / max(dt1, 1e-3)) - 1))
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.encode_subframes_batch(params, pcm, n, want_residual=False, slot_bytes=slot)
    dt = time.perf_counter() - t0
    samples = reps * frames * n * params.channels
    out = {
        "value": round(samples / dt / 1e6, 3),
        "unit": "Msamples/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{reps} x {frames} frames of the same synthetic workload "
                  f"({samples / 1e6:.1f} Msamples, {dt:.1f} s), oracle/flake_oracle.c, "
                  "prepare + encode_residual + Rice emit",
    }

    ref = reference_leg(params, n, frames, max(3.0, budget_s * 0.4))
    if ref is not None:
        out["reference"] = ref
        try:                                   # the same on every host core this process may use, one process each
            import subprocess
            ncores, navail = host_cores()
            args = [REF_BENCH, params.channels, params.bits_per_sample, params.sample_rate, n, params.order_method,
                    params.stereo_method, params.prediction_type, params.min_prediction_order,
                    params.max_prediction_order, params.min_partition_order, params.max_partition_order,
                    params.variable_block_size, params.allow_vbs, 256, 3.0, 0]
            procs = [subprocess.Popen([str(a) for a in args], stdout=subprocess.PIPE, text=True) for _ in range(ncores)]
            rates = []
            for pr in procs:
                o, _ = pr.communicate(timeout=120)
                j = json.loads(o.strip().splitlines()[-1])
                rates.append(j["samples"] / j["seconds"] / 1e6)
            out["reference_all_cores"] = {"value": round(sum(rates), 1), "unit": "Msamples/s", "cores": ncores,
                                          "cores_available": navail, "kind": "reference (timing only)",
                                          "sample": f"{ncores} processes of build/ref/ref_bench side by side, 256 frames "
                                                    "repeated for 3 s each; sum of their rates"}
        except Exception as e:
            out["reference_all_cores"] = {"error": repr(e)}

    # ---- every host core, one process each (spawned: this process holds the GPU)
    try:
        import multiprocessing as mp
        ncores, navail = host_cores()
        wf = 256
        wreps = max(1, int(reps * frames / wf * 0.5))            # ~ half the one-thread budget each
        pvals = [getattr(params, k) for k, _ in params._fields_]
        with mp.get_context("spawn").Pool(ncores) as pool:
            t0 = time.perf_counter()
            times = pool.map(_cpu_worker, [(wf, n, wreps, pvals)] * ncores)
            wall = time.perf_counter() - t0
        tot = ncores * wreps * wf * n * params.channels
        out["all_cores"] = {"value": round(tot / max(times) / 1e6, 1), "unit": "Msamples/s",
                            "cores": ncores, "cores_available": navail, "kind": "port",
                            "sample": f"{ncores} processes x {wreps} x {wf} frames, slowest worker "
                                      f"{max(times):.1f} s (pool wall {wall:.1f} s incl. start-up)"}
    except Exception as e:                                        # a reported extra, never fatal
        out["all_cores"] = {"error": repr(e)}

    # ---- the compiled reference stages on prepared subframes
    if oraclelib.Ref.available() and hasattr(oraclelib.Ref().L, "ref_time_hotpath") \
            and params.order_method == flake_amd.OM_MAX:
        ref = oraclelib.Ref()
        rf = 256
        smp = np.zeros((rf * params.channels, n), np.int32)
        obits = np.zeros(rf * params.channels, np.int32)
        for f in range(rf):
            _, s, sf = orc.prepare_frame(params, pcm[f], n)
            smp[f * params.channels:(f + 1) * params.channels] = s
            obits[f * params.channels:(f + 1) * params.channels] = sf["obits"]
        rreps = max(1, int(budget_s * 0.4 / max(dt1 * rf / frames, 1e-3)))
        tt = np.zeros(5)
        bits = 0
        for _ in range(rreps):
            bits, t = ref.time_hotpath(smp, obits, params.max_prediction_order, params.lpc_precision,
                                       params.min_partition_order, params.max_partition_order)
            tt += t
        rs = rreps * rf * n * params.channels
        names = ("lpc_calc_coefs [lpc.c]", "integer FIR [harness loop]",
                 "calc_rice_params_lpc [rice.c]", "output_residual via BitWriter [bitio.h]",
                 "calc_crc16 [crc.c]")
        out["reference_stages"] = {
            "value": round(rs / tt.sum() / 1e6, 3), "unit": "Msamples/s", "cores": 1,
            "kind": "reference functions compiled from /root/reference (oracle/_ref) + a C loop "
                    "for the FIR; feeders (prepare) excluded",
            "ns_per_sample": {k: round(float(v) / rs * 1e9, 3) for k, v in zip(names, tt)},
            "sample": f"{rreps} x {rf} frames ({rs / 1e6:.1f} Msamples, {tt.sum():.1f} s)",
            "residual_bits_match_hip": (None if gpu_bits_per_frame is None
                                        else bool(bits == int(gpu_bits_per_frame[:rf].sum()))),
        }
    return out


# ---------------------------------------------------------------------------------------------
# GPU side extras (rank 0, N = 1)
# ---------------------------------------------------------------------------------------------
def copy_bandwidth(dev, achieved_gbps, nbytes=1 << 30, reps=10):
    """Device-to-device copy of `nbytes` on the current stream: bytes read + bytes written per
    second, the rate a kernel that only moves data reaches on this box."""
    import torch
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    src.zero_()
    for _ in range(3):
        dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize(dev)
    gbps = 2 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    return {"copy_GBps_measured": round(gbps, 1), "frac_of_copy": round(achieved_gbps / gbps, 4)}


def settled_ms(step, sync, steps, settle_ms=40.0):
    """ms per step at settled clocks: untimed steps for `settle_ms`, then `steps` timed ones."""
    for _ in range(3):
        step()
    sync()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < settle_ms:
        for _ in range(5):
            step()
        sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    return (time.perf_counter() - t0) / steps * 1e3


def kernel_profile(enc, step, steps, per_step=False):
    """hipEvent time per kernel (pairs on the launch stream, fhip_set_profiling): the average
    launch, or -- per_step: a step launches a kernel several times (a ragged batch's bins) --
    the sum over one step."""
    enc.set_profiling(True)
    enc.kernel_times(reset=True)
    for _ in range(steps):
        step()
    enc.sync()
    per = {k: tms / (steps if per_step else c) for k, (tms, c) in enc.kernel_times(reset=True).items() if c}
    enc.set_profiling(False)
    return per


def dominant_bound(per_kernel, dom):
    """The dominant kernel against ITS bound (the row's answer to "how close to speed of light")."""
    e = per_kernel.get(dom) or {}
    return {"kernel": dom, "bound": e.get("bound"), "frac": e.get("frac"), "current": e.get("current")}


def subframe_case(dev_index, tag, p, nframes, steps, with_residual=False, cpu=True, cpu_frames=48, case=None, cases=None):
    """One uniform-batch workload through fhip_encode_subframes_dev: ms per step at settled
    clocks, hipEvent time per kernel, what binds each, and a small CPU figure beside it."""
    import torch
    import flake_amd
    dev = torch.device("cuda", dev_index)
    n = p.block_size
    nsub = nframes * p.channels
    slot = flake_amd.rice_slot_bytes(p, n)
    pcm = torch.from_numpy(flake_amd.synth_pcm(nframes, n, p.channels, p.bits_per_sample)).to(dev)
    info = torch.zeros(nsub * flake_amd.INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    bits = torch.zeros(nsub * slot, dtype=torch.uint8, device=dev)
    resid = torch.zeros((nframes, p.channels, n), dtype=torch.int32, device=dev) if with_residual else None
    enc = flake_amd.Encoder(p, max_frames=nframes, device=dev_index)
    enc.set_stream(torch.cuda.current_stream(dev).cuda_stream)

    def step():
        enc.encode_subframes_dev(pcm, nframes, n, info, residual=resid, rice_bits=bits, slot_bytes=slot)
    ms = settled_ms(step, lambda: torch.cuda.synchronize(dev), steps)
    per = kernel_profile(enc, step, max(3, steps // 2))
    info_np = np.frombuffer(info.cpu().numpy().tobytes(), dtype=flake_amd.INFO_DTYPE)
    samples = nframes * n * p.channels
    rice_bytes = int(((info_np["rice_nbits"].clip(min=0) + 31) // 32 * 4).sum())
    alg = samples * 4 + rice_bytes + nsub * flake_amd.INFO_DTYPE.itemsize + (samples * 4 if with_residual else 0)
    dom = max(per, key=per.get)
    row = {
        "workload": tag, "frames": nframes, "samples_per_step": samples,
        "ms_per_step": round(ms, 4), "Msamples_per_s": round(samples / ms / 1e3, 1),
        "kernel_ms": {k: round(v, 4) for k, v in per.items()},
        "dominant_kernel": dom,
        "algorithmic_bytes": alg,
        "hbm_frac_dominant": round(alg / (per[dom] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        "hbm_frac_step": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        "per_kernel": per_kernel_bounds(p, n, nframes, per, rice_bytes, None, with_residual, case, cases),
        "bits_per_sample_out": round(float(info_np["rice_nbits"].clip(min=0).sum()) / samples, 3),
    }
    row["dominant_bound"] = dominant_bound(row["per_kernel"], dom)
    enc.close()
    del pcm, info, bits, resid
    torch.cuda.empty_cache()
    if cpu:
        try:
            row["cpu_baseline"] = small_cpu_baseline(p, n, cpu_frames)
        except Exception as e:
            row["cpu_baseline"] = {"error": repr(e)}
    return row


def vbs_case(dev_index, level, nblocks, steps, cpu=True, case=None, cases=None):
    """BASELINE configs[4] shape: variable block size (vbs.c) + exhaustive order / partition
    search, blocks and stream device-resident through fhip_encode_blocks_vbs_dev (no host
    synchronisation inside): ms per batch at settled clocks and the serial per-kernel sums."""
    import torch
    import flake_amd
    dev = torch.device("cuda", dev_index)
    p = flake_amd.level_params(level)
    n = p.block_size
    pcm_h = flake_amd.synth_pcm(nblocks, n, 2, 16)
    pcm_h[::3, n // 2:, :] //= 16            # a transient in every third block: something to split
    pcm = torch.from_numpy(pcm_h).to(dev)
    cap = pcm_h.size * 5
    packed = torch.zeros(cap, dtype=torch.uint8, device=dev)
    totals = torch.zeros(4, dtype=torch.int64, device=dev)
    enc = flake_amd.Encoder(p, max_frames=8 * nblocks, device=dev_index)
    enc.set_stream(torch.cuda.current_stream(dev).cuda_stream)

    def step():
        enc.encode_blocks_vbs_dev(pcm, nblocks, n, packed, cap, totals)
    ms = settled_ms(step, lambda: torch.cuda.synchronize(dev), steps)
    per = kernel_profile(enc, step, 3, per_step=True)      # (profiling serialises the bins: sums, not the step)
    t = totals.cpu().numpy()
    samples = nblocks * n * 2
    alg = samples * 4 + int(t[1])
    dom = max(per, key=per.get)
    tag = ("-l 12 -m 5 -r 8 -v 1, n 4096" if level == 10 else "-l 32 -m 5 -r 8 -v 1, n 8192")
    row = {
        "workload": f"configs[4]: level {level} (variable block size, SEARCH order method, partition orders "
                    f"0-8: {tag}), stereo 16-bit, {nblocks} blocks device-resident -> packed stream in HBM",
        "blocks": nblocks, "frames_out": int(t[0]), "bytes_out": int(t[1]), "samples_per_step": samples,
        "ms_per_step": round(ms, 4), "Msamples_per_s": round(samples / ms / 1e3, 1),
        "kernel_ms_serial_sums": {k: round(v, 4) for k, v in per.items()},
        "dominant_kernel": dom,
        "algorithmic_bytes": alg,
        "hbm_frac_step": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        "note": "eight bins of equal piece length; K0 / K1 / K2 one launch over all bins; order search / K3 per bin (the "
                "thinly filled 256-thread bins of a batch of <= 2048 blocks in one launch each) and K4 per lane on three "
                "lanes: the handle's stream and two internal ones; no host synchronisation inside",
    }
    if cases and case and isinstance(cases.get(case), dict):
        # the order searches and K3 of all bins against their instruction issue (sums over the bins' instances)
        pk = {}
        for k in ("k_order_search", "k_encode", "k_assemble"):
            if k in per:
                b = issue_bound(k, cases[case], per[k])
                if b:
                    b["source"] = f"profiles/pmc_cases.json [{case}], tag {cases.get('_tag')}"
                    b["current"] = cases.get("_current")
                    pk[k] = b
        row["per_kernel"] = pk
        row["dominant_bound"] = dominant_bound(pk, dom)
    enc.close()
    del pcm, packed
    torch.cuda.empty_cache()
    if cpu:
        try:
            row["cpu_baseline"] = small_cpu_baseline(p, n, 24, pcm=pcm_h[:24], transient=True)
        except Exception as e:
            row["cpu_baseline"] = {"error": repr(e)}
    return row


def other_configs(dev_index, steps, cpu=True):
    """The BASELINE configs the headline is NOT quoted on, at their full sizes, two more presets,
    and stage A of the headline (int32 residual out): reported next to the headline, never as
    `value`; every row carries CPU figures over >= 1 s each (port, one core; the reference's own loop,
    timing only) and prices its search / encode kernels by instruction issue from profiles/pmc_cases.json."""
    import flake_amd
    P = flake_amd.level_params
    rows = []
    pc = load_pmc_cases()
    cases = [
        ("configs[2]: stereo 24-bit 96 kHz, n 4096, LPC order SEARCH 1-32, partition orders 0-8",
         P(5, bits_per_sample=24, sample_rate=96000, order_method=flake_amd.OM_SEARCH,
           max_prediction_order=32, max_partition_order=8), 4096, False, 6, "c2"),
        ("configs[3]: 8-channel 24-bit 192 kHz, n 4096, LPC-12 (MAX), 32768 subframes",
         P(5, channels=8, bits_per_sample=24, sample_rate=192000, order_method=flake_amd.OM_MAX,
           max_prediction_order=12), 4096, False, 16, "c3"),
        ("configs[0] (the reference's CPU case) on the GPU: mono 16-bit, n 4096, fixed orders 0-4, "
         "partition orders 0-3", P(2, channels=1, block_size=4096), 8192, False, 64, "c0"),
        ("level 8: stereo 16-bit, n 4096, LPC <= 12 LOG search, partition orders 0-6", P(8), 4096, False, 24, "l8"),
        ("level 2: stereo 16-bit, n 1152, fixed orders 0-4, partition orders 0-3", P(2),
         4096 * 4096 // 1152, False, 128, "l2"),
        ("configs[1], stage A (SURVEY 8d): the headline workload with the int32 residual written too",
         P(5, order_method=flake_amd.OM_MAX), 4096, True, 48, None),
    ]
    for tag, p, nframes, with_res, cpu_frames, case in cases:
        row = subframe_case(dev_index, tag, p, nframes, steps, with_res, cpu, cpu_frames, case, pc)
        if case == "c0" and cpu:
            # BASELINE configs[0] as it is defined: "CPU reference via flake CLI, no GPU"
            cli = reference_cli_leg()
            if cli is not None:
                row["cpu_reference_cli"] = cli
        rows.append(row)
    for level in (10, 12):
        rows.append(vbs_case(dev_index, level, 1024, max(5, steps // 2), cpu, f"l{level}", pc))
    # the size SURVEY 8d gives configs[4] per GPU (>= 65536 blocks over 8 GPUs): 1024 blocks are latency-bound
    for level in (10, 12):
        rows.append(vbs_case(dev_index, level, 8192, 5, False, f"l{level}x8", pc))
    return rows


def small_batch(dev_index, steps, full_ms):
    """What a shard of an 8-way (4-, 2-way) split of ONE batch costs: the step at 512 / 1024 / 2048
    frames of configs[1] and at 512 / 1024 frames of configs[3], per kernel, with the strong-scaling
    efficiency it implies (step(4096) / (k x step(4096 / k)); launch gaps and K1's chain walk do
    not shrink with the batch)."""
    rows = []
    for name, frames in (("configs[1]", 512), ("configs[1]", 1024), ("configs[1]", 2048),
                         ("configs[3]", 512), ("configs[3]", 1024)):
        p, _, _ = workload(name)
        r = subframe_case(dev_index, f"{name} at {frames} frames", p, frames, steps, cpu=False)
        ways = 4096 // frames
        base = full_ms.get(name)
        rows.append({"workload": name, "frames": frames, "ms_per_step": r["ms_per_step"],
                     "kernel_ms": r["kernel_ms"], "split_ways": ways,
                     "implied_strong_scaling_efficiency":
                         (round(base / (ways * r["ms_per_step"]), 3) if base else None)})
    return rows


def host_path(nframes=4096, n=4096, vbs=True):
    """The PCIe-inclusive rate through the host C layer (flake_amd_encode_frames: pageable host
    PCM -> complete FLAC frames in host memory), with and without the stream MD5.  Reported
    beside the headline, never as `value`."""
    import ctypes as C
    import flake_amd
    pcm = flake_amd.synth_pcm(nframes, n, 2, 16)
    flat = np.ascontiguousarray(pcm, dtype=np.int32).reshape(-1, 2)
    cap = 64 + pcm.size * 5 + 64 * (nframes + 1) * 8
    out = np.ones(cap, dtype=np.uint8)                   # touched: no page faults in the timed calls
    sizes = np.zeros(nframes, dtype=np.int32)
    res = {"frames": nframes, "samples": nframes * n * 2}
    saved = {k: os.environ.get(k) for k in ("FLAKE_AMD_MD5", "FLAKE_AMD_BATCH")}
    try:
        os.environ["FLAKE_AMD_BATCH"] = str(nframes)
        for key, md5, pin in (("ms_md5_off", "0", True), ("ms_md5_off_pageable", "0", False), ("ms_md5_on", "1", True)):
            if (md5 == "1" or not pin) and not vbs:
                continue                                 # (the per-rank leg: the stream's bytes only)
            os.environ["FLAKE_AMD_MD5"] = md5
            enc = flake_amd.HostEncoder(level=5, channels=2, bits_per_sample=16, sample_rate=44100,
                                        block_size=n, order_method=flake_amd.OM_MAX)
            if pin:      # the caller's two buffers page-locked in place (flake_amd_pin_buffers), once, before the loop
                t0 = time.perf_counter()
                prc = enc.lib.flake_amd_pin_buffers(C.byref(enc.ctx), flat.ctypes.data, flat.nbytes, out.ctypes.data,
                                                    min(cap, flat.nbytes // 2 + (1 << 20)))
                res["pin_buffers_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
                res["pin_buffers_rc"] = int(prc)
            best = None
            for call in range(4 if md5 == "0" else 2):   # the first call pays one-time allocations
                t0 = time.perf_counter()
                w = enc.lib.flake_amd_encode_frames(C.byref(enc.ctx), flat.ctypes.data, nframes, n, 0,
                                                    out.ctypes.data, cap, sizes.ctypes.data)
                dt = time.perf_counter() - t0
                if w <= 0:
                    raise RuntimeError("flake_amd_encode_frames failed")
                if call:
                    best = dt if best is None else min(best, dt)
            enc.close()
            res[key] = round(best * 1e3, 3)
        res["Msamples_per_s_md5_off"] = round(res["samples"] / res["ms_md5_off"] / 1e3, 1)
        if not vbs:
            return res
        # BASELINE configs[4] territory through the same host entry: upload, the device-resident
        # batch (fhip_encode_blocks_vbs_dev), download of the stream's bytes
        os.environ["FLAKE_AMD_MD5"] = "0"
        vb = {}
        for level in (10, 12):
            nblk = 1024
            os.environ["FLAKE_AMD_BATCH"] = str(nblk)
            enc = flake_amd.HostEncoder(level=level, channels=2, bits_per_sample=16, sample_rate=44100)
            bs = enc.params().block_size
            vp = flake_amd.synth_pcm(nblk, bs, 2, 16)
            vp[::3, bs // 2:, :] //= 16                  # a transient in every third block: something to split
            vflat = np.ascontiguousarray(vp, dtype=np.int32).reshape(-1, 2)
            vcap = 64 + vp.size * 5 + 64 * (nblk + 1) * 8
            vout = np.ones(vcap, dtype=np.uint8)
            vsizes = np.zeros(nblk, dtype=np.int32)
            best = None
            for call in range(3):
                t0 = time.perf_counter()
                w = enc.lib.flake_amd_encode_frames(C.byref(enc.ctx), vflat.ctypes.data, nblk, bs, 0,
                                                    vout.ctypes.data, vcap, vsizes.ctypes.data)
                dt = time.perf_counter() - t0
                if w <= 0:
                    raise RuntimeError("flake_amd_encode_frames (vbs) failed")
                if call:
                    best = dt if best is None else min(best, dt)
            enc.close()
            vb[f"level{level}"] = {"blocks": nblk, "block_size": bs, "ms": round(best * 1e3, 3),
                                   "Msamples_per_s": round(nblk * bs * 2 / best / 1e6, 1)}
        res["vbs_presets_md5_off"] = vb
        res["note"] = ("ms_md5_off / ms_md5_on: the caller's PCM and output buffers page-locked in place once "
                       "(flake_amd_pin_buffers, pin_buffers_ms); ms_md5_off_pageable: pageable memory in and out as in "
                       "rounds 1-3; copies over PCIe, kernels and frame packing overlap across two handles; MD5 "
                       "(sequential over the stream) on a helper thread.  An unmodified single-stream caller of "
                       "flake_encode_frame() always has the MD5 on (encode.c:1006): it is MD5-bound at ms_md5_on")
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return res


def launch_ranks(ngpus):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD
    processes (torch.distributed.run, one per GPU) and pass their output through;
    rank 0 prints the JSON line.  This parent has only parsed arguments -- it has
    not initialised the GPU and it never exec()s."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={ngpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def host_path_child():
    """host_path() in a process of its own: inside the long bench process (torch's streams and allocator, the CPU
    legs' worker pools before it) the same calls measured 3.5 ms where a fresh process measures 2.86 -- the caller
    this row stands for is a plain C program that does nothing else."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--host-path-only"], capture_output=True, text=True,
                       timeout=600)
    for line in reversed(r.stdout.strip().splitlines()):
        if line.startswith("{"):
            out = json.loads(line)
            out["process"] = "child process of bench.py (python bench.py --host-path-only)"
            return out
    raise RuntimeError("host path child failed: " + r.stderr[-300:])


def main():
    args = parse_args()
    if args.host_path_only:
        print(json.dumps(host_path()), flush=True)
        return
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    import torch
    import flake_amd
    from flake_amd.shard import shard_range

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)          # several ranks may share a GPU in rehearsals
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # BENCH_FORCE_DIST=1: go through the collective path (RCCL init, all-reduce, barrier,
    # all-gather on this rank's stream) even with one rank -- what a one-GPU box can rehearse
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST", "") == "1"
    backend = None
    if use_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        if "MASTER_ADDR" not in os.environ:            # forced single-rank rehearsal without a launcher
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]),
                              RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
            sk.close()
        # "nccl" is RCCL on ROCm.  With fewer GPUs than ranks (a rehearsal of the N > 1
        # path on a one-GPU box) ranks share a card, which RCCL refuses: gloo then.
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl" if ndev >= world else "gloo")
        # RCCL and gloo both print a banner on stdout from their C++ side when the first
        # communicator comes up; stdout carries the one JSON line, so it points at stderr meanwhile
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=dev)
                warm = torch.zeros(1, dtype=torch.int64, device=dev)
            else:
                dist.init_process_group(backend=backend)
                warm = torch.zeros(1, dtype=torch.int64)
            dist.all_reduce(warm)                      # brings the communicator up
            dist.barrier()
            torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    # ---- workload: BASELINE.json configs[1] (or configs[3]) ------------------
    p, metric, wl_text = workload(args.workload)
    n = p.block_size
    if args.scaling == "strong":
        first, last = shard_range(args.frames, rank, world)     # one job, contiguous shards
        nframes = last - first
        if nframes < 1:
            raise SystemExit("--scaling strong needs at least one frame per rank")
        first_frame = first
    else:
        nframes = args.frames                                     # the same batch size on every rank
        first_frame = rank * args.frames
    nsub = nframes * p.channels
    slot = flake_amd.rice_slot_bytes(p, n)

    # default: the one resident batch.  With --ahead two batches of this rank's shard alternate
    # (different frames of the same signal model): while batch i is in flight the handle is told
    # batch i+1 is ready (fhip_prepare_ahead)
    nbatches = 2 if args.ahead else 1
    pcms = [torch.from_numpy(flake_amd.synth_pcm(nframes, n, p.channels, p.bits_per_sample,
                                                 first_frame=nbatches * first_frame + k * nframes)).to(dev)
            for k in range(nbatches)]
    if nbatches == 1:
        pcms.append(pcms[0])
    info_bytes = flake_amd.INFO_DTYPE.itemsize
    info = torch.zeros(nsub * info_bytes, dtype=torch.uint8, device=dev)
    bits = torch.zeros(nsub * slot, dtype=torch.uint8, device=dev)
    resid = torch.zeros((nframes, p.channels, n), dtype=torch.int32, device=dev) \
        if args.with_residual else None

    enc = flake_amd.Encoder(p, max_frames=nframes, device=dev_index)
    # one stream of our own for the encoder and for torch's reads of its outputs (torch's default
    # stream has handle 0, which the C ABI takes as "the handle's own stream": not ordered with it)
    torch.cuda.synchronize(dev)          # uploads and fills above ran on the default stream
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    enc.set_stream(stream.cuda_stream)
    # gloo reduces host tensors; RCCL device tensors
    cdev = dev if (not use_dist or backend == "nccl") else torch.device("cpu")

    ahead = args.ahead
    count = [0]

    def step():
        i = count[0]
        count[0] += 1
        enc.encode_subframes_dev(pcms[i % 2], nframes, n, info, residual=resid, rice_bits=bits,
                                 slot_bytes=slot)
        if ahead:                 # the next batch's feeder stage, beside the kernels just queued
            enc.prepare_ahead(pcms[(i + 1) % 2], nframes, n)

    batch_bits = [0, 0]           # residual bits of the two batches (filled before the timed region)
    batch_rice_bytes = [0, 0]
    nb_view = info.view(torch.int32).view(nsub, info_bytes // 4)[:, 10]

    def info_bits():
        nb = nb_view.clamp(min=0)
        return int(nb.sum().item()), int(((nb + 31) // 32 * 4).sum().item())

    # (the constant words of the exchange live on the device before the timed region starts: building a device
    # tensor from a Python number is a pageable upload with a synchronisation of its own -- two of them were ~0.1 ms
    # of a 20-step run's 3 ms)
    stats_words = torch.zeros(3, dtype=torch.int64, device=dev)
    stats_words[2] = 1

    def job_stats(steps):
        """The job's only exchange: {frames, residual bits} summed over ranks.  The
        residual bits are reduced ON THE DEVICE from the records the last step wrote (one torch
        reduction, inside the timed region) and count `steps` times: every step encodes the same
        resident batch (two alternating ones with --ahead, whose totals were read beforehand)."""
        last_bits = nb_view.clamp(min=0).sum()                        # device scalar, no host sync
        if ahead:
            other = batch_bits[count[0] % 2]
            job_bits = last_bits * ((steps + 1) // 2) + other * (steps // 2)
        else:
            job_bits = last_bits * steps
        stats_words[0].fill_(nframes * steps)
        stats_words[1].copy_(job_bits)
        st = stats_words.to(cdev)
        if use_dist:
            dist.all_reduce(st)
        return st

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(2):           # also the first use of these torch kernels (loads their code
        step()                   # objects: seconds of host time, before the clocks are ramped)
        torch.cuda.synchronize(dev)
        batch_bits[k], batch_rice_bytes[k] = info_bits()
    job_stats(0)
    fence()
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        for _ in range(10):
            step()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    stats = job_stats(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if args.steps:
        # behind the closing fence: what the last timed step wrote is the batch's total
        last_bits = info_bits()[0]
        if last_bits != batch_bits[(count[0] - 1) % 2]:
            raise SystemExit(f"bench: the last timed step wrote {last_bits} residual bits, "
                             f"the batch's total is {batch_bits[(count[0] - 1) % 2]}")
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        per_rank = [torch.zeros(2, dtype=torch.int64, device=cdev) for _ in range(world)]
        dist.all_gather(per_rank, torch.tensor([rank, nframes], dtype=torch.int64, device=cdev))
        rank_batch = {int(r[0]): int(r[1]) for r in per_rank}
    else:
        rank_batch = {0: nframes}
    frames_per_step = sum(rank_batch.values())
    samples_per_step = frames_per_step * n * p.channels
    value = samples_per_step * args.steps / dt / 1e6

    # ---- N > 1: the PCIe-inclusive host path of every rank's shard, side by side ----
    host_ranks = None
    if use_dist and world > 1 and not args.no_other_configs and args.workload == "configs[1]":
        try:
            dist.barrier()
            hp = host_path(nframes=max(64, nframes), n=n, vbs=False)
            mine = torch.tensor([hp["ms_md5_off"], hp["samples"]], dtype=torch.float64, device=cdev)
        except Exception:
            mine = torch.tensor([0.0, 0.0], dtype=torch.float64, device=cdev)
        allhp = [torch.zeros(2, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(allhp, mine)
        if rank == 0:
            ms = [float(x[0]) for x in allhp]
            host_ranks = {"per_rank_ms_md5_off": [round(m, 3) for m in ms],
                          "sum_Msamples_per_s": round(sum(float(x[1]) / float(x[0]) / 1e3 for x in allhp if float(x[0]) > 0), 1),
                          "note": "each rank: pageable host PCM -> FLAC frames in host memory for its shard "
                                  "(flake_amd_encode_frames), all ranks at once; never `value`"}

    # ---- roofline of the dominant kernel (rank 0, hipEvents on the launch stream)
    roofline = None
    cpu = None
    others = None
    host = None
    small = None
    if rank == 0:
        rice_bytes = sum(batch_rice_bytes) // 2            # the two alternating batches, averaged
        alg_bytes = (nframes * n * p.channels * 4          # int32 PCM in
                     + rice_bytes                          # packed residual sections out
                     + nsub * info_bytes                   # side info out
                     + (nframes * n * p.channels * 4 if args.with_residual else 0))
        per = kernel_profile(enc, step, args.profile_steps) if args.profile_steps else {}
        torch.cuda.synchronize(dev)
        enc.encode_subframes_dev(pcms[0], nframes, n, info, residual=resid, rice_bits=bits,
                                 slot_bytes=slot)           # batch 0's records for the CPU cross-check
        enc.sync()
        info_np = np.frombuffer(info.cpu().numpy().tobytes(), dtype=flake_amd.INFO_DTYPE)
        pmc = load_pmc() if args.workload == "configs[1]" else None
        if per:                          # empty with --profile-steps 0 (the PMC passes)
            dom = max(per, key=per.get)
            achieved = alg_bytes / (per[dom] * 1e-3) / 1e9
            traffic = (pmc or {}).get(dom, {}).get("hbm_bytes_per_launch") if nframes == 4096 else None
            roofline = {
                "bound": "hbm", "kernel": dom,
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "traffic_source": f"profiles/pmc_traffic.json, tag {pmc.get('_tag')}" if pmc else None,
                "traffic_current": pmc.get("_current") if pmc else None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": {k: round(v, 4) for k, v in per.items()},
                # what binds each kernel and how close it runs to THAT bound: the step is three
                # launches of which only the first is an HBM problem
                "per_kernel": per_kernel_bounds(p, n, nframes, per, rice_bytes, pmc, args.with_residual),
                "prepare_ahead": ahead,
                "step_ms": round(dt / args.steps * 1e3, 4),
                "pipeline_achieved": round(alg_bytes / (dt / args.steps) / 1e9, 1),
                "pipeline_frac": round(alg_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBPS, 4),
            }
            try:                             # the fraction of a *measured* copy too
                roofline.update(copy_bandwidth(dev, achieved))
            except Exception as e:
                roofline["copy_GBps_measured"] = None
                roofline["copy_error"] = repr(e)
            try:                             # all HBM bytes the step's kernels moved (PMC) / step
                if pmc is None or nframes != 4096:
                    raise KeyError(args.workload)
                moved = sum(v["hbm_bytes_per_launch"] for k, v in pmc.items()
                            if not k.startswith("_") and isinstance(v, dict) and "hbm_bytes_per_launch" in v)
                roofline["pipeline_traffic"] = moved
                roofline["pipeline_traffic_GBps"] = round(moved / (dt / args.steps) / 1e9, 1)
            except Exception:
                roofline["pipeline_traffic"] = None
        if world == 1 and not args.no_cpu_baseline:
            per_frame_bits = info_np["rice_nbits"].clip(min=0).astype(np.int64) \
                .reshape(nframes, p.channels).sum(axis=1)
            try:
                cpu = cpu_baseline(p, n, args.cpu_seconds, per_frame_bits)
            except Exception as e:
                cpu = {"error": repr(e)}
        if world == 1 and not args.no_other_configs and args.workload == "configs[1]":
            try:
                others = other_configs(dev_index, args.other_steps, cpu=not args.no_cpu_baseline)
            except Exception as e:                      # reported extras must not cost the headline line
                others = [{"error": repr(e)}]
            try:
                full = {"configs[1]": dt / args.steps * 1e3 if nframes == 4096 else None}
                for r in others or []:
                    if isinstance(r, dict) and str(r.get("workload", "")).startswith("configs[3]"):
                        full["configs[3]"] = r["ms_per_step"]
                small = small_batch(dev_index, args.other_steps, full)
            except Exception as e:
                small = [{"error": repr(e)}]
            try:
                host = host_path_child()
            except Exception as e:                      # a reported extra, never fatal
                host = {"error": repr(e)}

    if use_dist:
        torch.cuda.synchronize(dev)
        dist.barrier()

    if rank == 0:
        out = {
            "metric": metric,
            "value": round(value, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "int64/f64",
            "data": "synthetic",
            "config": {
                "workload": wl_text,
                "frames_per_gpu": nframes if args.scaling == "weak" else [rank_batch.get(r, 0) for r in range(world)],
                "job_frames_per_step": frames_per_step,
                "samples_per_step": samples_per_step,
                "batches": ("two batches of this shard alternate; the next one's feeder stage is "
                            "hinted ahead (fhip_prepare_ahead)") if ahead else "one resident batch",
                "outputs": "subframe info + packed Rice residual sections"
                           + (" + int32 residual" if args.with_residual else ""),
                "parallelism": f"frame-sharded x{world}"
                               + (" (one job cut into contiguous shards)" if args.scaling == "strong" else ""),
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
            "other_configs": others,
            "small_batch_ms": small,
            "host_path": host,
            "host_path_ranks": host_ranks,
        }
        out["job_frames"] = int(stats[0].item())
        out["job_residual_bits"] = int(stats[1].item())
        out["job_residual_bits_note"] = ("the last timed step's records reduced on the device inside the timed "
                                         "region, times the steps (every step encodes the same resident batch)")
        out["ranks_seen"] = int(stats[2].item())
        out["rank_frames"] = [rank_batch.get(r, 0) * args.steps for r in range(world)]
        out["dist_backend"] = backend if use_dist else None
        print(json.dumps(out), flush=True)

    enc.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
