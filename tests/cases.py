"""Deterministic edge-case inputs shared by the CPU and GPU tests."""
from __future__ import annotations

import numpy as np

import flake_amd


def _rng(seed):
    return np.random.RandomState(seed)     # legacy generator: stable across numpy versions


def full_scale(bps):
    return (1 << (bps - 1)) - 1


def edge_blocks(n: int, bps: int = 16, seed: int = 7) -> dict:
    """Single-channel blocks of length n exercising the branches of encode_residual()."""
    r = _rng(seed)
    fs = full_scale(bps)
    t = np.arange(n)
    out = {
        "zeros": np.zeros(n, np.int64),
        "dc": np.full(n, 1234 % (fs + 1), np.int64),
        "dc_neg_fs": np.full(n, -fs - 1, np.int64),
        "impulse": np.where(t == n // 3, fs, 0),
        "step": np.where(t >= n // 2, fs // 2, -(fs // 2)),
        "ramp": (t * 3 - n) % (fs + 1),
        "alt_full_scale": np.where(t & 1, fs, -fs - 1),
        "white": r.randint(-fs - 1, fs + 1, n),
        "small_noise": r.randint(-3, 4, n),
        "sine": np.round(0.7 * fs * np.sin(2 * np.pi * t * 0.0123)).astype(np.int64),
        "sine_plus_noise": np.round(0.5 * fs * np.sin(2 * np.pi * t * 0.031)).astype(np.int64)
                           + r.randint(-40, 41, n),
        "wasted_3": r.randint(-(fs >> 3) - 1, (fs >> 3) + 1, n) << 3,
        "wasted_max": np.where(t % 7 == 0, -fs - 1, 0),          # only -2^(bps-1): tz = bps-1
        "one_nonzero": np.where(t == n - 1, 1, 0),
        "decay": np.round(fs * 0.9 * np.exp(-t / max(n / 6, 1)) * np.cos(t * 0.4)).astype(np.int64),
    }
    return {k: np.clip(v, -fs - 1, fs).astype(np.int32) for k, v in out.items()}


def stereo_frames(n: int, bps: int = 16, seed: int = 11) -> dict:
    """Interleaved stereo frames [n][2] covering every channel mode."""
    r = _rng(seed)
    fs = full_scale(bps)
    b = edge_blocks(n, bps, seed)
    base = b["sine_plus_noise"].astype(np.int64)
    noise = r.randint(-20, 21, n)
    out = {
        "identical": (base, base),                              # side channel CONSTANT 0
        "left_only": (base, np.zeros(n, np.int64)),
        "right_only": (np.zeros(n, np.int64), base),
        "anti": (base, -base),                                  # mid ~ 0, side = 2x: obits+1 matters
        "near": (base, base + noise),
        "near_r": (base + 3 * noise, base),
        "independent": (b["white"].astype(np.int64), r.randint(-fs - 1, fs + 1, n)),
        "both_zero": (np.zeros(n, np.int64), np.zeros(n, np.int64)),
        "full_scale_anti": (np.where(np.arange(n) & 1, fs, -fs - 1), np.where(np.arange(n) & 1, -fs - 1, fs)),
        "wasted_pair": ((base >> 4) << 4, (base >> 4 << 4) + (noise >> 2 << 2)),
        "dc_pair": (np.full(n, 100, np.int64), np.full(n, -100, np.int64)),
    }
    res = {}
    for k, (l, rr) in out.items():
        fr = np.stack([np.clip(l, -fs - 1, fs), np.clip(rr, -fs - 1, fs)], axis=1)
        res[k] = np.ascontiguousarray(fr.astype(np.int32))
    return res


# (name, Params kwargs, block sizes to run) -- the BASELINE.json configs plus
# the parameter corners of flake_validate_params()
def param_sets() -> list:
    P = flake_amd.level_params
    return [
        ("c1_mono16_fixed", P(2, channels=1, block_size=4096), 4096),
        ("c2_stereo16_lpc8_max", P(5, order_method=flake_amd.OM_MAX), 4096),
        ("c2_stereo16_lpc8_est", P(5), 4096),
        ("c3_stereo24_search32", P(5, bits_per_sample=24, sample_rate=96000,
                                   order_method=flake_amd.OM_SEARCH, min_prediction_order=1,
                                   max_prediction_order=32, min_partition_order=0,
                                   max_partition_order=8), 4096),
        ("c4_8ch24_lpc12", P(5, channels=8, bits_per_sample=24, sample_rate=192000,
                             order_method=flake_amd.OM_MAX, max_prediction_order=12), 4096),
        ("c5_level10", P(10, variable_block_size=0), 4096),
        ("level8_log", P(8), 4096),
        ("level7_4level", P(7), 4096),
        ("level12_bs8192", P(12, variable_block_size=0), 8192),
        ("two_level", P(5, order_method=flake_amd.OM_2LEVEL, max_prediction_order=16), 4096),
        ("eight_level", P(5, order_method=flake_amd.OM_8LEVEL, max_prediction_order=32,
                          min_prediction_order=4), 4096),
        ("pred_none", P(5, prediction_type=flake_amd.PRED_NONE), 1024),
        ("fixed_only_o3", P(2, min_prediction_order=3, max_prediction_order=3), 1152),
        ("porder_pinned", P(5, min_partition_order=4, max_partition_order=4,
                            order_method=flake_amd.OM_MAX), 4096),
        ("bps8", P(5, bits_per_sample=8, order_method=flake_amd.OM_MAX), 4096),
        ("bps32", P(5, bits_per_sample=32, order_method=flake_amd.OM_MAX), 2048),
        ("bps20_mono", P(6, channels=1, bits_per_sample=20, block_size=4608), 4608),
    ]


ODD_BLOCK_SIZES = (16, 17, 33, 100, 192, 255, 576, 1152, 1536, 2304, 4608, 5000)
TINY_BLOCK_SIZES = (1, 2, 3, 4, 5, 8, 9, 12)


# ---- the seeded random sweep of tests/test_gpu_fuzz.py and tests/test_ref_replay.py ----
BLOCKS = (16, 31, 64, 100, 192, 256, 384, 500, 576, 1024, 1152, 1536, 2048, 2304, 4096, 4608, 5000,
          8192, 16384)


def fuzz_signal(r, kind, nfr, n, ch, bps):
    full = 1 << (bps - 1)
    if kind == 0:                                    # the bench's resonator
        return flake_amd.synth_pcm(nfr, n, ch, bps, first_frame=int(r.randint(0, 1000)))
    if kind == 1:                                    # white noise, a random number of low bits zero
        sh = int(r.randint(0, min(bps - 1, 6)))
        return ((r.randint(-full, full, (nfr, n, ch)).astype(np.int64) >> sh) << sh).astype(np.int32)
    if kind == 2:                                    # quiet: a few LSBs around a DC offset
        dc = int(r.randint(-full // 2, full // 2))
        return (dc + r.randint(-3, 4, (nfr, n, ch))).astype(np.int32)
    if kind == 3:                                    # sine + correlated second channel
        t = np.arange(n)[None, :, None]
        a = (0.7 * full * np.sin(t * r.uniform(0.001, 0.3))).astype(np.int64)
        x = a + r.randint(-8, 9, (nfr, n, ch))
        return np.clip(x, -full, full - 1).astype(np.int32)
    if kind == 5:                                    # +-full scale alternating, phase flips
        sgn = np.where((np.arange(n) + (np.arange(n) // max(int(r.randint(3, 50)), 1))) % 2 == 0, 1, -1)
        x = (sgn[None, :, None] * (full - 1) - (sgn[None, :, None] < 0)).astype(np.int64)
        return np.broadcast_to(x, (nfr, n, ch)).astype(np.int32).copy()
    if kind == 6:                                    # exact polynomial: a fixed predictor leaves zeros
        t = np.arange(n, dtype=np.int64)[None, :, None]
        a, b, c = int(r.randint(-3, 4)), int(r.randint(-200, 200)), int(r.randint(-full // 4, full // 4))
        x = np.clip(a * t * t // 64 + b * t // 8 + c, -full, full - 1)
        return np.broadcast_to(x, (nfr, n, ch)).astype(np.int32).copy()
    if kind == 7:                                    # silence, then a burst of noise, then quiet
        x = np.zeros((nfr, n, ch), dtype=np.int64)
        lo, hi = sorted(int(v) for v in r.randint(0, n + 1, 2))
        x[:, lo:hi, :] = r.randint(-full, full, (nfr, hi - lo, ch))
        x[:, hi:, :] = r.randint(-2, 3, (nfr, n - hi, ch))
        return x.astype(np.int32)
    x = np.zeros((nfr, n, ch), dtype=np.int32)       # constant blocks, one full-scale click
    x[:, :, :] = int(r.randint(-full, full))
    if n > 8:
        x[0, n // 2, 0] = full - 1
    return x


def fuzz_params(r, n, ch, bps):
    ptype = int(r.choice([flake_amd.PRED_LEVINSON] * 6 + [flake_amd.PRED_FIXED] * 2 + [flake_amd.PRED_NONE]))
    if ptype == flake_amd.PRED_FIXED:
        lo = int(r.randint(0, 5)); hi = int(r.randint(lo, 5))
        om = flake_amd.OM_EST
    else:
        lo = int(r.randint(1, 33)); hi = int(r.randint(lo, 33))
        if r.rand() < 0.6:
            hi = min(hi, 12)
            lo = min(lo, hi)
        om = int(r.randint(0, 7))
        if om in (flake_amd.OM_SEARCH,) and hi > 16 and n >= 4096:
            hi = 16; lo = min(lo, hi)                # keeps the oracle's share of the run short
    if ptype != flake_amd.PRED_FIXED and n <= hi:
        # the reference then takes its FIXED branch with this min order (optimize.c:168-173);
        # above 4 that is an out-of-bounds `bits[]` write and an unwritten residual -- undefined
        lo = min(lo, 4)
    plo = int(r.randint(0, 9)); phi = int(r.randint(plo, 9))
    return flake_amd.level_params(5, channels=ch, bits_per_sample=bps, block_size=max(n, 16),
                                  prediction_type=ptype, order_method=om,
                                  min_prediction_order=lo, max_prediction_order=hi,
                                  min_partition_order=plo, max_partition_order=phi,
                                  stereo_method=int(r.randint(0, 2)))


def fuzz_case(seed: int):
    """(params, pcm[nfr][n][ch], n, description) of subframe-level fuzz seed `seed`."""
    r = np.random.RandomState(1000 + seed)
    n = int(BLOCKS[r.randint(0, len(BLOCKS))])
    ch = int(r.choice([1, 2, 2, 2, 3, 6, 8]))
    bps = int(r.choice([8, 12, 16, 16, 16, 20, 24, 24, 32]))
    p = fuzz_params(r, n, ch, bps)
    nfr = 2 if n * ch > 20000 else int(r.randint(2, 6))
    kind = int(r.randint(0, 5))
    pcm = fuzz_signal(r, kind, nfr, n, ch, bps)
    what = f"seed {seed}: n={n} ch={ch} bps={bps} pred={p.prediction_type} om={p.order_method} " \
           f"order {p.min_prediction_order}..{p.max_prediction_order} " \
           f"porder {p.min_partition_order}..{p.max_partition_order}"
    return p, pcm, n, what
