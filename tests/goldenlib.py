"""Loaders for the committed golden vectors (tests/golden/*.npz)."""
import os

import numpy as np

import flake_amd

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def params_from_array(a) -> flake_amd.Params:
    p = flake_amd.Params()
    for (k, _), v in zip(p._fields_, a):
        setattr(p, k, int(v))
    return p


def rice_records(z):
    for i in range(int(z["count"])):
        yield {k: z[f"{k}_{i}"] for k in ("res", "lpc", "order", "pmin", "pmax", "bits", "method",
                                          "porder", "params", "emit_nbits", "emit", "emit_rc")}


def split_bits(info, flat):
    """Undo the concatenation of per-subframe residual sections."""
    out, pos = [], 0
    for s in range(info.size):
        nb = max(int(info["rice_nbits"][s]), 0)
        ln = (nb + 7) // 8
        out.append(flat[pos:pos + ln])
        pos += ln
    return out
