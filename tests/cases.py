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
