"""Comparison helpers shared by the GPU parity tests."""
from __future__ import annotations

import numpy as np

SCALARS = ("type", "type_code", "order", "shift", "obits", "wasted", "rice_method", "porder",
           "est_bits", "ch_mode", "rice_nbits")


def assert_info_equal(got: np.ndarray, exp: np.ndarray, what: str = "") -> None:
    """fhip_subframe_info records vs the oracle's, field by field (bit-exact)."""
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    for k in SCALARS:
        bad = np.nonzero(got[k] != exp[k])[0]
        assert bad.size == 0, (
            f"{what}: field {k!r} differs in {bad.size}/{got.size} subframes; first at "
            f"{bad[0]}: got {got[k][bad[0]]} expected {exp[k][bad[0]]}")
    for s in range(got.size):
        if exp["type"][s] == 32:
            o = exp["order"][s]
            assert (got["coefs"][s][:o] == exp["coefs"][s][:o]).all(), (
                what, "coefs", s, got["coefs"][s][:o], exp["coefs"][s][:o])
        nw = 1 if exp["type"][s] == 0 else exp["order"][s]
        assert (got["warmup"][s][:nw] == exp["warmup"][s][:nw]).all(), (what, "warmup", s)
        if exp["type"][s] in (8, 32):
            npart = 1 << exp["porder"][s]
            assert (got["rparams"][s][:npart] == exp["rparams"][s][:npart]).all(), (
                what, "rparams", s, got["rparams"][s][:npart], exp["rparams"][s][:npart])


def assert_residual_equal(got: np.ndarray, exp: np.ndarray, info: np.ndarray, what: str = "") -> None:
    n = got.shape[-1]
    g = got.reshape(-1, n)
    e = exp.reshape(-1, n)
    for s in range(info.size):
        if info["type"][s] == 0:        # CONSTANT: only residual[0] is defined (optimize.c:147)
            assert g[s, 0] == e[s, 0], (what, "constant residual", s)
        else:
            bad = np.nonzero(g[s] != e[s])[0]
            assert bad.size == 0, (
                f"{what}: residual of subframe {s} differs at {bad.size} samples, first i={bad[0]}: "
                f"got {g[s, bad[0]]} expected {e[s, bad[0]]}")


def assert_bits_equal(got: np.ndarray, exp: np.ndarray, info: np.ndarray, what: str = "") -> None:
    """Residual sections byte for byte up to their bit length (pad bits zero)."""
    for s in range(info.size):
        nb = int(info["rice_nbits"][s])
        if nb <= 0:
            continue
        nbytes = (nb + 7) // 8
        bad = np.nonzero(got[s, :nbytes] != exp[s, :nbytes])[0]
        assert bad.size == 0, (
            f"{what}: rice bits of subframe {s} differ in {bad.size}/{nbytes} bytes, first at "
            f"byte {bad[0]}: got {got[s, bad[0]]:#x} expected {exp[s, bad[0]]:#x}")
