"""Pins the CPU restatement (oracle/flake_oracle.c) to the REAL reference code.

oracle/_ref/libflake_ref.so is the reference's own lpc.c, rice.c, crc.c and
bitio.h compiled where they lie (oracle/Makefile).  Everything here must agree
bit for bit -- fp64 outputs are compared through their uint64 images.
Skipped where _ref is absent (no /root/reference and no prebuilt file).
"""
import numpy as np
import pytest

import flake_amd
from cases import edge_blocks, _rng


def bits64(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def synth_blocks(nblocks, n, bps):
    pcm = flake_amd.synth_pcm(nblocks, n, 2, bps)
    return np.ascontiguousarray(pcm.transpose(0, 2, 1)).reshape(-1, n)


@pytest.mark.parametrize("n,bps", [(4096, 16), (4096, 24), (1152, 16), (256, 8), (8192, 24), (34, 16)])
def test_autocorr(oracle, ref, n, bps):
    """lpc.c:28-71: window + autocorrelation, all lags, every bit."""
    blocks = list(synth_blocks(3, n, bps)) + list(edge_blocks(n, bps).values())
    for lag in (1, 2, 8, 12, 32):
        if n <= lag:
            continue
        for b in blocks:
            got = oracle.window_autocorr(b, lag)[:lag + 1]
            exp = ref.compute_autocorr(b, lag)[:lag + 1]
            assert (bits64(got) == bits64(exp)).all(), (n, bps, lag)


@pytest.mark.parametrize("max_order", [1, 2, 5, 8, 12, 31, 32])
def test_levinson_schur_quantise(oracle, ref, max_order):
    """lpc.c:77-219 on real autocorrelations and on adversarial ones."""
    r = _rng(3)
    acs = [ref.compute_autocorr(b, max_order) for b in synth_blocks(6, 4096, 16)]
    acs += [ref.compute_autocorr(b, max_order) for b in edge_blocks(2048, 24).values()]
    # hand-made sequences: near-singular, tiny, huge, negative lags
    acs.append(np.concatenate([[1.0], 0.999 ** np.arange(1, 33)]))
    acs.append(np.concatenate([[1e30], 1e30 * 0.5 ** np.arange(1, 33)]))
    acs.append(np.concatenate([[3.0], r.uniform(-1, 1, 32)]))
    acs.append(np.concatenate([[1e-20], np.zeros(32)]))
    for ac in acs:
        ac = np.ascontiguousarray(ac[:33], np.float64)
        with np.errstate(all="ignore"):
            a = oracle.levinson(ac, max_order)
            b = ref.compute_lpc_coefs(ac, max_order)
        for i in range(max_order):
            assert (bits64(a[i, :i + 1]) == bits64(b[i, :i + 1])).all(), (max_order, i)
        ea, la = oracle.schur_order_est(ac, max_order)
        eb, lb = ref.compute_lpc_coefs_est(ac, max_order)
        assert ea == eb
        assert (bits64(la[ea - 1, :ea]) == bits64(lb[eb - 1, :eb])).all()
        for i in range(max_order):
            row = b[i, :i + 1].copy()
            if not np.isfinite(row).all():
                continue
            for prec in (15, 12, 5):
                qa = oracle.quantize_coefs(row, i + 1, prec)
                qb = ref.quantize_lpc_coefs(row, i + 1, prec)
                assert qa[1] == qb[1] and (qa[0][:i + 1] == qb[0][:i + 1]).all(), (max_order, i, prec)


def test_quantiser_corners(oracle, ref):
    """lpc.c:167-219: zero-out, shift clamp, in-place rescale, clamp of q."""
    rows = [
        [0.0], [1e-6, -1e-6], [3.05e-5, 3.04e-5], [0.49999, -0.5, 0.5],
        [1.0, -1.0, 0.99997], [1.99, -1.99, 0.3], [255.9, -17.3, 4.0], [16383.0, -16383.0],
        [16384.0, -3.0], [40000.0, -39999.5, 12.25, -0.125], [1e9, -1e9, 1.0],
        [0.333333, 0.666666, -0.999999, 0.5, -0.5, 0.25, -0.25, 0.125],
    ]
    for row in rows:
        for prec in (15, 14, 8, 2):
            qa = oracle.quantize_coefs(row, len(row), prec)
            qb = ref.quantize_lpc_coefs(row, len(row), prec)
            assert qa[1] == qb[1] and (qa[0] == qb[0]).all(), (row, prec)


@pytest.mark.parametrize("omethod", range(7))
def test_lpc_calc_coefs(oracle, ref, omethod):
    """lpc.c:224-257 end to end, all order methods."""
    blocks = list(synth_blocks(4, 4096, 16)) + list(synth_blocks(2, 1152, 24))
    blocks += [b for k, b in edge_blocks(1024, 16).items() if k not in ("zeros",)]
    for b in blocks:
        for max_order in (1, 8, 12, 32):
            with np.errstate(all="ignore"):
                ca, sa, oa = oracle.lpc_calc_coefs(b, max_order, 15, omethod)
                cb, sb, ob = ref.lpc_calc_coefs(b, max_order, 15, omethod)
            assert oa == ob
            assert (ca == cb).all() and (sa == sb).all(), (omethod, max_order)


def test_find_optimal_rice_param(oracle, ref):
    """rice.c:30-45 incl. the uint64 wrap (sum < n/2) and the 32-bit truncation."""
    r = _rng(5)
    sums = [0, 1, 2, 7, 8, 100, 2047, 2048, 2049, 65535, 1 << 20, (1 << 32) - 1, 1 << 32,
            (1 << 40) + 12345, (1 << 48) - 1, (1 << 63), (1 << 64) - 1]
    sums += [int(x) for x in r.randint(0, 1 << 31, 200)]
    sums += [int(x) << 20 for x in r.randint(0, 1 << 30, 100)]
    for n in (0, 1, 2, 16, 17, 128, 4095, 4096, 65535):
        for s in sums:
            assert oracle.rice_best_k(s, n) == ref.find_optimal_rice_param(s, n), (s, n)


@pytest.mark.parametrize("n", [4096, 1152, 576, 192, 4608, 16, 5000])
def test_calc_rice_params(oracle, ref, n):
    """rice.c:47-187: sums pyramid, per-partition k, partition order, bit estimate."""
    r = _rng(n)
    residuals = [
        r.randint(-50, 51, n), r.randint(-30000, 30001, n), np.zeros(n, np.int64),
        (r.standard_cauchy(n) * 20).clip(-2 ** 30, 2 ** 30), r.randint(-2 ** 31, 2 ** 31 - 1, n),
        np.where(np.arange(n) < n // 2, r.randint(-3, 4, n), r.randint(-20000, 20001, n)),
        np.full(n, -2 ** 31), np.full(n, 2 ** 31 - 1),
    ]
    for res in residuals:
        res = np.asarray(res).astype(np.int64).astype(np.int32)
        for lpc, order in ((1, 1), (1, 8), (1, 32), (0, 0), (0, 2), (0, 4)):
            if order >= n:
                continue
            for pmin, pmax in ((0, 0), (0, 3), (0, 5), (0, 8), (2, 6), (8, 8), (4, 4)):
                bits_a, sf = oracle.subframe_bits(res, pmin, pmax, order, 17, 15, lpc)
                bits_b, meth, por, par = ref.calc_rice_params(lpc, pmin, pmax, res, order, 17, 15)
                assert bits_a == bits_b, (n, lpc, order, pmin, pmax)
                assert sf["rice_method"] == meth and sf["porder"] == por
                assert (sf["rparams"][:1 << por] == par[:1 << por]).all()


@pytest.mark.parametrize("n", [4096, 1152, 64, 16])
def test_emit_residual(oracle, ref, n):
    """encode.c:766-798 loop through the reference BitWriter (bitio.h:83-141)."""
    r = _rng(n + 1)
    for amp, order, lpc in ((40, 8, 1), (3000, 2, 0), (2, 0, 0), (2 ** 20, 12, 1), (2 ** 31 - 1, 1, 1)):
        if order >= n:
            continue
        res = r.randint(-amp, amp + 1, n).astype(np.int64).astype(np.int32)
        if amp > 2 ** 16:
            res[::97] = 0
        for pmax in (0, 4, 8):
            _, sf = oracle.subframe_bits(res, 0, pmax, order, 16, 15, lpc)
            sf = sf.copy()
            sf["order"] = order
            nb_expect = oracle.residual_section_bits(sf, res)
            cap = int(nb_expect // 8 + 64)
            if cap > 1 << 26:
                continue
            nb, out = oracle.emit_residual(sf, res, cap)
            rc, nbr, outr = ref.emit_residual(int(sf["rice_method"]), int(sf["porder"]),
                                              sf["rparams"], order, res, cap)
            assert nb == nb_expect == nbr, (n, amp, pmax)
            nbytes = (nb + 7) // 8
            assert rc == nbytes
            assert (out[:nbytes] == outr[:nbytes]).all(), (n, amp, pmax)


def test_emit_overflow_matches_eof(oracle, ref):
    """A slot that is too small: the oracle reports -1 where the reference hits eof."""
    r = _rng(9)
    res = r.randint(-2000, 2001, 1024).astype(np.int32)
    _, sf = oracle.subframe_bits(res, 0, 4, 4, 16, 15, 1)
    sf = sf.copy()
    sf["order"] = 4
    need = (oracle.residual_section_bits(sf, res) + 7) // 8
    nb, _ = oracle.emit_residual(sf, res, int(need) - 8)
    assert nb == -1
    rc, _, _ = ref.emit_residual(int(sf["rice_method"]), int(sf["porder"]), sf["rparams"], 4,
                                 res, int(need) - 8)
    assert rc == -1


def test_crc(oracle, ref):
    r = _rng(2)
    for ln in (0, 1, 2, 15, 16, 255, 4096, 17000):
        d = r.randint(0, 256, ln).astype(np.uint8)
        assert oracle.crc8(d) == ref.crc8(d)
        assert oracle.crc16(d) == ref.crc16(d)


def test_porder_limit_and_log2(ref):
    """rice.c:148-155 / common.h:53-65 against their closed forms."""
    for v in (1, 2, 3, 4, 255, 256, 257, 65535, 65536, 2 ** 31, 2 ** 32 - 1):
        assert ref.L.ref_log2i(v) == v.bit_length() - 1
    for n in (16, 192, 576, 1152, 4096, 4608, 5000, 65535):
        for order in (0, 1, 4, 8, 32):
            if order >= n:
                continue
            for mp in range(9):
                tz = (n & -n).bit_length() - 1
                exp = min(mp, tz)
                if order:
                    exp = min(exp, (n // order).bit_length() - 1)
                assert ref.L.ref_limit_max_partition_order(mp, n, order) == exp
