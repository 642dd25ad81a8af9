"""GPU parity: the HIP path through the C ABI vs the CPU oracle, bit for bit."""
import numpy as np
import pytest

import flake_amd
from parity import assert_bits_equal, assert_info_equal, assert_residual_equal

pytestmark = pytest.mark.gpu


def run_both(oracle, p, pcm, n):
    with flake_amd.Encoder(p, max_frames=pcm.shape[0]) as enc:
        got = enc.encode_subframes(pcm, n, want_samples=True, want_autoc=True)
    exp = oracle.encode_subframes_batch(p, pcm, n, slot_bytes=got["slot_bytes"])
    return got, exp


def check(oracle, p, pcm, n, what):
    got, exp = run_both(oracle, p, pcm, n)
    assert_info_equal(got["info"], exp["info"], what)
    assert_residual_equal(got["residual"], exp["residual"], exp["info"], what)
    assert_bits_equal(got["rice_bits"], exp["rice_bits"], exp["info"], what)
    return got, exp


def test_autocorr_bit_exact(oracle):
    """compute_autocorr (lpc.c:46-71): every fp64 output bit-identical."""
    pcm = flake_amd.synth_pcm(24, 4096, 2, 16)
    smp = np.ascontiguousarray(pcm.transpose(0, 2, 1)).reshape(-1, 4096)
    p = flake_amd.level_params(5)
    with flake_amd.Encoder(p, max_frames=24) as enc:
        for max_order in (1, 8, 12, 32):
            _, _, _, autoc = enc.lpc_calc_coefs(smp, max_order, 15, flake_amd.OM_MAX)
            for s in range(smp.shape[0]):
                exp = oracle.window_autocorr(smp[s], max_order)
                assert (autoc[s, :max_order + 1].view(np.uint64) ==
                        exp[:max_order + 1].view(np.uint64)).all(), (max_order, s)


@pytest.mark.parametrize("omethod", range(7))
def test_lpc_calc_coefs(oracle, omethod):
    """lpc_calc_coefs (lpc.c:224-257): quantised rows, shifts, order estimate."""
    pcm = flake_amd.synth_pcm(16, 4096, 2, 24)
    smp = np.ascontiguousarray(pcm.transpose(0, 2, 1)).reshape(-1, 4096)
    p = flake_amd.level_params(5, bits_per_sample=24)
    with flake_amd.Encoder(p, max_frames=16) as enc:
        for max_order in (8, 12, 32):
            coefs, shift, opt, _ = enc.lpc_calc_coefs(smp, max_order, 15, omethod)
            for s in range(smp.shape[0]):
                ec, es, eo = oracle.lpc_calc_coefs(smp[s], max_order, 15, omethod)
                assert opt[s] == eo, (max_order, s)
                assert (coefs[s] == ec).all(), (max_order, s)
                assert (shift[s] == es).all(), (max_order, s)


def test_config2_stereo16_lpc8(oracle):
    """BASELINE config 2 shape: stereo 16-bit, n=4096, LPC-8 (order method MAX)."""
    pcm = flake_amd.synth_pcm(96, 4096, 2, 16)
    p = flake_amd.level_params(5, order_method=flake_amd.OM_MAX)
    got, exp = check(oracle, p, pcm, 4096, "config2")
    # feeder stage: samples after decorrelation + wasted bits
    for f in range(pcm.shape[0]):
        _, smp, _ = oracle.prepare_frame(p, pcm[f], 4096)
        assert (got["samples"][f] == smp).all(), f


@pytest.mark.parametrize("level", range(9))
def test_levels_0_to_8(oracle, level):
    p = flake_amd.level_params(level)
    n = p.block_size
    pcm = flake_amd.synth_pcm(12, n, 2, 16)
    check(oracle, p, pcm, n, f"level{level}")


@pytest.mark.parametrize("omethod", range(7))
def test_order_methods_24bit(oracle, omethod):
    p = flake_amd.level_params(5, bits_per_sample=24, sample_rate=96000, order_method=omethod,
                               max_prediction_order=12, max_partition_order=8)
    pcm = flake_amd.synth_pcm(8, 4096, 2, 24)
    check(oracle, p, pcm, 4096, f"omethod{omethod}")


# ---------------------------------------------------------------------------
# committed golden vectors (outputs of the real reference functions)
# ---------------------------------------------------------------------------
import goldenlib as G                                   # noqa: E402
from cases import (ODD_BLOCK_SIZES, TINY_BLOCK_SIZES, edge_blocks, param_sets,   # noqa: E402
                   stereo_frames)


def test_golden_ref_lpc():
    """HIP lpc_calc_coefs vs reference lpc.c outputs: autoc bit-for-bit, coefs, shifts."""
    z = G.load("ref_lpc.npz")
    blocks = z["blocks"]
    p = flake_amd.level_params(5, block_size=blocks.shape[1])
    with flake_amd.Encoder(p, max_frames=blocks.shape[0]) as enc:
        for oi, mo in enumerate(z["orders"]):
            mo = int(mo)
            for om in range(7):
                coefs, shift, opt, autoc = enc.lpc_calc_coefs(blocks, mo, 15, om)
                assert (autoc[:, :mo + 1].view(np.uint64) == z["autoc_bits"][:, oi, :mo + 1]).all()
                assert (opt == z["opt_order"][:, oi, om]).all(), (mo, om)
                assert (coefs == z["coefs"][:, oi, om]).all(), (mo, om)
                assert (shift == z["shift"][:, oi, om]).all(), (mo, om)


def test_golden_ref_rice_and_emit():
    """HIP Rice search + emit vs reference rice.c / bitio.h outputs."""
    z = G.load("ref_rice.npz")
    for rec in G.rice_records(z):
        res = rec["res"][None, :]
        n = res.shape[1]
        p = flake_amd.level_params(5, block_size=max(n, 16), bits_per_sample=17)
        slot = (len(rec["emit"]) + 64 + 3) & ~3
        with flake_amd.Encoder(p, max_frames=1) as enc:
            out = enc.calc_rice_params(res, int(rec["order"]), bool(rec["lpc"]), 17,
                                       int(rec["pmin"]), int(rec["pmax"]), slot_bytes=slot)
        info = out["info"][0]
        assert info["est_bits"] == int(rec["bits"])
        assert info["rice_method"] == rec["method"] and info["porder"] == rec["porder"]
        npart = 1 << int(rec["porder"])
        assert (info["rparams"][:npart] == rec["params"][:npart]).all()
        assert info["rice_nbits"] == int(rec["emit_nbits"])
        assert (out["rice_bits"][0, :len(rec["emit"])] == rec["emit"]).all()


def test_golden_path_configs():
    """Whole path vs the committed regression vectors for every BASELINE config shape."""
    z = G.load("path_configs.npz")
    for name in z["names"]:
        p = G.params_from_array(z[f"params_{name}"])
        n = int(z[f"n_{name}"])
        pcm = z[f"pcm_{name}"]
        with flake_amd.Encoder(p, max_frames=pcm.shape[0]) as enc:
            got = enc.encode_subframes(pcm, n)
        exp_info = z[f"info_{name}"]
        assert_info_equal(got["info"], exp_info, name)
        assert_residual_equal(got["residual"], z[f"residual_{name}"], exp_info, name)
        for s, sec in enumerate(G.split_bits(exp_info, z[f"bits_{name}"])):
            assert (got["rice_bits"][s, :len(sec)] == sec).all(), (name, s)


# ---------------------------------------------------------------------------
# edge cases, every parameter corner, ragged block sizes
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("name,p,n", param_sets(), ids=[c[0] for c in param_sets()])
def test_param_sets(oracle, name, p, n):
    nfr = 6 if p.channels <= 2 else 3
    pcm = flake_amd.synth_pcm(nfr, n, p.channels, p.bits_per_sample, first_frame=40)
    check(oracle, p, pcm, n, name)


@pytest.mark.parametrize("bps", [16, 24])
def test_stereo_edge_frames(oracle, bps):
    fr = stereo_frames(4096, bps)
    pcm = np.stack([fr[k] for k in sorted(fr)])
    for om in (flake_amd.OM_MAX, flake_amd.OM_EST, flake_amd.OM_LOG):
        p = flake_amd.level_params(5, bits_per_sample=bps, order_method=om)
        got, exp = check(oracle, p, pcm, 4096, f"stereo_edges_{bps}_{om}")
        if bps == 16:
            assert {1, 8, 9, 10} <= set(int(m) for m in exp["info"]["ch_mode"])


@pytest.mark.parametrize("bps", [8, 16, 24, 32])
def test_mono_edge_blocks(oracle, bps):
    blocks = edge_blocks(4096, bps)
    pcm = np.stack([blocks[k] for k in sorted(blocks)])[:, :, None]
    for kw in (dict(order_method=flake_amd.OM_MAX), dict(order_method=flake_amd.OM_SEARCH,
               max_prediction_order=12, max_partition_order=8), dict(prediction_type=flake_amd.PRED_FIXED,
               min_prediction_order=0, max_prediction_order=4)):
        p = flake_amd.level_params(5, channels=1, bits_per_sample=bps, **kw)
        got, exp = check(oracle, p, pcm, 4096, f"mono_edges_{bps}")
        types = set(int(t) for t in exp["info"]["type"])
        assert 0 in types                                   # CONSTANT occurs


@pytest.mark.parametrize("n", ODD_BLOCK_SIZES + TINY_BLOCK_SIZES)
def test_ragged_block_sizes(oracle, n):
    """Short final blocks, non-power-of-two sizes, n <= max_order (FIXED path),
    n < 5 (VERBATIM)."""
    for lvl, ch in ((5, 2), (8, 1), (2, 2)):
        p = flake_amd.level_params(lvl, channels=ch, block_size=max(n, 16))
        pcm = flake_amd.synth_pcm(5, n, ch, 16, first_frame=n)
        check(oracle, p, pcm, n, f"n{n}_l{lvl}")


def test_short_block_in_bigger_handle(oracle):
    """block_size < params.block_size (the last block of a stream, encode.c:987-990)."""
    p = flake_amd.level_params(5)
    pcm = flake_amd.synth_pcm(7, 1000, 2, 16)
    with flake_amd.Encoder(p, max_frames=16) as enc:
        got = enc.encode_subframes(pcm, 1000)
    exp = oracle.encode_subframes_batch(p, pcm, 1000, slot_bytes=got["slot_bytes"])
    assert_info_equal(got["info"], exp["info"], "short")
    assert_residual_equal(got["residual"], exp["residual"], exp["info"], "short")
    assert_bits_equal(got["rice_bits"], exp["rice_bits"], exp["info"], "short")


def test_slot_too_small_reports_minus_one(oracle):
    """A residual section that does not fit its slot: rice_nbits = -1, nothing written."""
    p = flake_amd.level_params(5, channels=1, order_method=flake_amd.OM_MAX)
    pcm = edge_blocks(4096, 16)["white"][None, :, None]
    with flake_amd.Encoder(p, max_frames=1) as enc:
        out = enc.calc_rice_params(pcm[:, :, 0], 0, False, 16, 0, 5, slot_bytes=1024)
    assert out["info"]["rice_nbits"][0] == -1
    assert not out["rice_bits"].any()


def test_full_size_batch_properties():
    """BASELINE configs[1] at full size (4096 frames): properties that need no oracle --
    the residual reproduces the samples through the FLAC decoder recurrence, and the
    residual section length equals the sum of its codeword lengths."""
    p = flake_amd.level_params(5, order_method=flake_amd.OM_MAX)
    n, nfr = 4096, 4096
    pcm = flake_amd.synth_pcm(nfr, n, 2, 16)
    with flake_amd.Encoder(p, max_frames=nfr) as enc:
        got = enc.encode_subframes(pcm, n, want_samples=True)
    info, res, smp = got["info"], got["residual"].reshape(-1, n), got["samples"].reshape(-1, n)
    assert (info["type"] == 32).all() and (info["order"] == 8).all()
    # decoder recurrence (vectorised over subframes): x[i] = r[i] + (sum c_j x[i-j] >> shift)
    rec = res.astype(np.int64).copy()
    coefs = info["coefs"][:, :8].astype(np.int64)
    shift = info["shift"].astype(np.int64)
    for i in range(8, n):
        pred = (coefs * rec[:, i - 8:i][:, ::-1]).sum(axis=1) >> shift
        rec[:, i] += pred
    assert (rec == smp).all()
    # sum of codeword lengths == rice_nbits
    u = (res.astype(np.int64) << 1) ^ (res.astype(np.int64) >> 63)
    for s in range(0, info.size, 257):
        po, k = int(info["porder"][s]), info["rparams"][s]
        psz = n >> po
        kk = np.repeat(k[:1 << po], psz)[8:]
        bits = 6 + (4 + int(info["rice_method"][s])) * (1 << po) + int(((u[s, 8:] >> kk) + 1 + kk).sum())
        assert bits == info["rice_nbits"][s], s
    # linearity of the bookkeeping: ch_mode consistent within a frame, obits rule
    assert (info["ch_mode"][0::2] == info["ch_mode"][1::2]).all()


def test_vbs_split_matches_reference_rule(oracle):
    """K-vbs vs split_frame_v1 (vbs.c:36-83) incl. the 32-bit abs/multiply quirk."""
    r = np.random.RandomState(5)
    for ch, bps, n in ((2, 16, 4096), (1, 24, 8192), (8, 24, 1024), (2, 32, 2048)):
        p = flake_amd.level_params(10, channels=ch, bits_per_sample=bps, block_size=n)
        base = flake_amd.synth_pcm(12, n, ch, bps)
        blocks = []
        for b in range(12):
            blk = base[b].copy()
            cut = (b % 8) * n // 8
            if b % 3 == 0:
                blk[:cut] //= 128
            elif b % 3 == 1:
                blk[cut:] = r.randint(-5, 6, blk[cut:].shape)
            blocks.append(blk)
        if bps == 32:      # large scores: the int abs()/imul wrap of vbs.c:69 comes into play
            blocks[0][: n // 2] = r.randint(-2 ** 31, 2 ** 31 - 1, (n // 2, ch))
        pcm = np.stack(blocks).astype(np.int32)
        with flake_amd.Encoder(p, max_frames=12) as enc:
            nf, sizes = enc.vbs_split(pcm, n)
        for b in range(12):
            enf, esz = oracle.vbs_split(pcm[b], ch, n)
            assert nf[b] == enf, (ch, bps, b)
            assert (sizes[b, :enf] == esz).all() and (sizes[b, enf:] == 0).all()


# ---------------------------------------------------------------------------
# BASELINE configs[2] and [3] at full size: size-independent properties
# ---------------------------------------------------------------------------

def _check_properties(p, pcm, n, got, what):
    """(1) the FLAC decoder recurrence inverts the residual to FlacSubframe.samples,
    (2) rice_nbits equals the sum of the codeword lengths it stands for,
    (3) the device-assembled frames decode (CRC-8/16) -- checked by the caller."""
    info = got["info"]
    res = got["residual"].reshape(-1, n).astype(np.int64)
    smp = got["samples"].reshape(-1, n).astype(np.int64)
    assert (info["type"] == 32).all(), what
    order = info["order"].astype(np.int64)
    shift = info["shift"].astype(np.int64)
    coefs = info["coefs"].astype(np.int64)
    rec = res.copy()
    maxo = int(order.max())
    for i in range(1, n):
        lo = max(0, i - maxo)
        # prediction with per-subframe order: taps beyond `order` have zero coefficients
        hist = rec[:, lo:i][:, ::-1]                       # x[i-1], x[i-2], ...
        pred = (coefs[:, :hist.shape[1]] * hist).sum(axis=1) >> shift
        active = i >= order
        rec[:, i] = np.where(active, rec[:, i] + pred, rec[:, i])
    assert (rec == smp).all(), what
    u = (res << 1) ^ (res >> 63)
    for s in range(0, info.size, max(1, info.size // 64)):
        po, o = int(info["porder"][s]), int(order[s])
        psz = n >> po
        kk = np.repeat(info["rparams"][s][:1 << po].astype(np.int64), psz)[o:]
        bits = 6 + (4 + int(info["rice_method"][s])) * (1 << po) + int(((u[s, o:] >> kk) + 1 + kk).sum())
        assert bits == info["rice_nbits"][s], (what, s)


def test_config3_full_size_properties(oracle, decoder):
    """configs[2]: stereo 24-bit 96 kHz, n=4096, order search 1-32 + partition search 0-8,
    4096 frames on the GPU; the oracle checks a 48-frame slice bit for bit."""
    p = flake_amd.level_params(5, bits_per_sample=24, sample_rate=96000,
                               order_method=flake_amd.OM_SEARCH, min_prediction_order=1,
                               max_prediction_order=32, min_partition_order=0, max_partition_order=8)
    n, nfr = 4096, 4096
    pcm = flake_amd.synth_pcm(nfr, n, 2, 24)
    with flake_amd.Encoder(p, max_frames=nfr) as enc:
        got = enc.encode_subframes(pcm, n, want_samples=True, want_frames=True)
    sl = slice(1000, 1048)
    exp = oracle.encode_subframes_batch(p, pcm[sl], n, slot_bytes=got["slot_bytes"])
    assert_info_equal(got["info"][2 * sl.start:2 * sl.stop], exp["info"], "config3 slice")
    assert_residual_equal(got["residual"][sl], exp["residual"], exp["info"], "config3 slice")
    sub = {k: (v[::16] if k in ("residual", "samples") else v) for k, v in got.items()}
    sub["info"] = got["info"].reshape(-1, 2)[::16].reshape(-1)
    _check_properties(p, pcm[::16], n, sub, "config3")
    stream = np.concatenate([got["frames"][f, :got["frame_bytes"][f]] for f in range(0, nfr, 8)])
    # frames carry their own numbers, so any subset decodes frame by frame
    out, _ = decoder.decode(stream, 2, 24, (nfr // 8) * n)
    assert (out.reshape(-1, n, 2) == pcm[::8]).all()


def test_config4_full_size_properties(oracle, decoder):
    """configs[3]: 8 channels, 24-bit, 192 kHz, n=4096, LPC-12, 1024 frames (8192 subframes)."""
    p = flake_amd.level_params(5, channels=8, bits_per_sample=24, sample_rate=192000,
                               order_method=flake_amd.OM_MAX, max_prediction_order=12)
    n, nfr = 4096, 1024
    pcm = flake_amd.synth_pcm(nfr, n, 8, 24)
    with flake_amd.Encoder(p, max_frames=nfr) as enc:
        got = enc.encode_subframes(pcm, n, want_samples=True, want_frames=True)
    sl = slice(500, 516)
    exp = oracle.encode_subframes_batch(p, pcm[sl], n, slot_bytes=got["slot_bytes"])
    assert_info_equal(got["info"][8 * sl.start:8 * sl.stop], exp["info"], "config4 slice")
    assert_residual_equal(got["residual"][sl], exp["residual"], exp["info"], "config4 slice")
    sub = {k: (v[::8] if k in ("residual", "samples") else v) for k, v in got.items()}
    sub["info"] = got["info"].reshape(-1, 8)[::8].reshape(-1)
    _check_properties(p, pcm[::8], n, sub, "config4")
    stream = np.concatenate([got["frames"][f, :got["frame_bytes"][f]] for f in range(0, nfr, 16)])
    out, _ = decoder.decode(stream, 8, 24, (nfr // 16) * n)
    assert (out.reshape(-1, n, 8) == pcm[::16]).all()
    # sharding property: two half batches give the same records as the whole
    with flake_amd.Encoder(p, max_frames=nfr // 2) as enc:
        a = enc.encode_subframes(pcm[:nfr // 2], n, want_residual=False, want_bits=False)
        b = enc.encode_subframes(pcm[nfr // 2:], n, want_residual=False, want_bits=False)
    assert np.concatenate([a["info"], b["info"]]).tobytes() == got["info"].tobytes()


@pytest.mark.parametrize("n", [192, 256, 384, 512, 576, 768, 1024, 1152, 1536, 2048, 2304, 2560, 3072,
                               3584, 4608, 5120, 6144, 7168, 8192, 9216, 12288, 16384])
@pytest.mark.parametrize("kw", [
    dict(order_method=flake_amd.OM_MAX),
    dict(order_method=flake_amd.OM_LOG, max_prediction_order=12, max_partition_order=8),
    dict(order_method=flake_amd.OM_SEARCH, max_prediction_order=32, max_partition_order=8),
    dict(prediction_type=flake_amd.PRED_FIXED, min_prediction_order=0, max_prediction_order=4),
], ids=["max8", "log12", "search32", "fixed"])
def test_every_fast_path_geometry(oracle, n, kw):
    """k_encode_pow2<C,T>: every (C,T) the launcher can pick -- C = 4, 8, 16 for
    powers of two, C = 9 and 3 for FLAC's 576/1152/2304/4608 and 192/384/... -- for
    stereo 16-bit, mono 24-bit and (wide residuals, long codewords) mono 32-bit."""
    for ch, bps in ((2, 16), (1, 24), (1, 32)):
        p = flake_amd.level_params(5, channels=ch, bits_per_sample=bps, block_size=n, **kw)
        pcm = flake_amd.synth_pcm(3, n, ch, bps, first_frame=n // 256)
        check(oracle, p, pcm, n, f"n{n}")


def test_fused_prepare_path_subprocess():
    """FHIP_FUSE=1: K0 only decides, the K1 producers apply channel mode / wasted
    bits and write the samples.  The switch is read once per process, so the
    comparison with the oracle runs in a child."""
    import subprocess, sys, os, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import flake_amd
        from oraclelib import Oracle
        from cases import stereo_frames
        from parity import assert_info_equal, assert_residual_equal, assert_bits_equal
        o = Oracle()
        fr = stereo_frames(4096, 16)
        pcm = np.concatenate([np.stack([fr[k] for k in sorted(fr)]), flake_amd.synth_pcm(40, 4096, 2, 16)])
        for kw in (dict(order_method=flake_amd.OM_MAX), dict()):
            p = flake_amd.level_params(5, **kw)
            with flake_amd.Encoder(p, max_frames=pcm.shape[0]) as enc:
                got = enc.encode_subframes(pcm, 4096)
            exp = o.encode_subframes_batch(p, pcm, 4096, slot_bytes=got["slot_bytes"])
            assert_info_equal(got["info"], exp["info"], "fused")
            assert_residual_equal(got["residual"], exp["residual"], exp["info"], "fused")
            assert_bits_equal(got["rice_bits"], exp["rice_bits"], exp["info"], "fused")
        print("fused ok")
    """ % (root, os.path.join(root, "tests")))
    env = dict(os.environ, FHIP_FUSE="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fused ok" in r.stdout, r.stdout + r.stderr


def test_matrix_search_fallback_subprocess():
    """mm_search32 gives a subframe to the general way when its folded residuals or leaf sums leave the 32-bit fast
    forms (values many times full scale) -- AFTER a finished matrix pass has used the workgroup's LDS.  Real signals
    never get there; FHIP_MM32_FORCE_FALLBACK=1 sends every subframe that way: the whole bits[] table must still be
    the oracle's, at every instance the matrix search serves (runs of 8 ... 28 and 16)."""
    import subprocess, sys, os, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import flake_amd
        from oraclelib import Oracle
        from parity import assert_info_equal, assert_bits_equal
        o = Oracle()
        for n in (2048, 3072, 4096, 5120, 6144, 7168):
            for bps in (16, 24):
                p = flake_amd.level_params(5, bits_per_sample=bps, block_size=n, order_method=flake_amd.OM_SEARCH,
                                           max_prediction_order=32, max_partition_order=8)
                pcm = flake_amd.synth_pcm(9, n, 2, bps, first_frame=n)
                with flake_amd.Encoder(p, max_frames=9) as enc:
                    got = enc.encode_subframes(pcm, n)
                exp = o.encode_subframes_batch(p, pcm, n, slot_bytes=got["slot_bytes"])
                assert_info_equal(got["info"], exp["info"], "mm fallback")
                assert_bits_equal(got["rice_bits"], exp["rice_bits"], exp["info"], "mm fallback")
        print("mm fallback ok")
    """ % (root, os.path.join(root, "tests")))
    env = dict(os.environ, FHIP_MM32_FORCE_FALLBACK="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "mm fallback ok" in r.stdout, r.stdout + r.stderr


def test_lag_split_tail_subprocess():
    """FHIP_SPLIT_TAIL=1: in a lag-split K1 launch (small batches, two workgroups per tile of 32 subframes) the
    second workgroup of a tile to arrive runs K2 as its tail (per-tile arrival counters, agent-scope release /
    acquire).  Off by default (measured no faster); kept correct: several batches through one handle (the counters
    run on from launch to launch), orders 8 and 12, a ragged last tile.  The switch is read once per process."""
    import subprocess, sys, os, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import flake_amd
        from oraclelib import Oracle
        from parity import assert_info_equal, assert_bits_equal
        o = Oracle()
        for kw, nfr in ((dict(order_method=flake_amd.OM_MAX), 200), (dict(order_method=flake_amd.OM_MAX, max_prediction_order=12), 77),
                        (dict(), 130)):
            p = flake_amd.level_params(5, **kw)
            with flake_amd.Encoder(p, max_frames=nfr) as enc:
                for rep in range(3):
                    pcm = flake_amd.synth_pcm(nfr - rep, 4096, 2, 16, first_frame=100 * rep)
                    got = enc.encode_subframes(pcm, 4096)
                    exp = o.encode_subframes_batch(p, pcm, 4096, slot_bytes=got["slot_bytes"])
                    assert_info_equal(got["info"], exp["info"], "split tail")
                    assert_bits_equal(got["rice_bits"], exp["rice_bits"], exp["info"], "split tail")
        print("split tail ok")
    """ % (root, os.path.join(root, "tests")))
    env = dict(os.environ, FHIP_SPLIT_TAIL="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "split tail ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("n", [512, 2048, 4096, 8192])
def test_narrow_sample_rows(oracle, n):
    """K0 stores a channel as int16 when all its (shifted) samples fit: both channels
    narrow, only one (a 17-bit side / a loud right channel), none (24-bit), and narrow
    only thanks to wasted bits.  Outputs must not depend on the storage width, and
    the int32 rows are still what a caller of `samples` gets."""
    r = np.random.RandomState(n)
    t = np.arange(n)
    quiet = (3000 * np.sin(t * 0.01)).astype(np.int64)
    frames16 = [
        np.stack([quiet + r.randint(-50, 50, n), quiet + r.randint(-50, 50, n)], 1),          # narrow / narrow
        np.stack([30000 * np.sign(np.sin(t * 0.3)), -30000 * np.sign(np.sin(t * 0.3))], 1),   # side needs 17 bits
        np.stack([r.randint(-32768, 32768, n), r.randint(-100, 100, n)], 1),                  # noise left, quiet right
        np.stack([np.full(n, -32768), np.full(n, 32767)], 1),                                 # constants at the rails
    ]
    pcm16 = np.stack(frames16).astype(np.int32)
    for om, kw in ((flake_amd.OM_MAX, {}), (flake_amd.OM_EST, {}), (flake_amd.OM_4LEVEL, {}),
                   (flake_amd.OM_EST, dict(prediction_type=flake_amd.PRED_FIXED, min_prediction_order=0,
                                           max_prediction_order=4))):
        p = flake_amd.level_params(5, block_size=n, order_method=om, **kw)
        with flake_amd.Encoder(p, max_frames=pcm16.shape[0]) as enc:
            got = enc.encode_subframes(pcm16, n)                       # narrow rows allowed
            rows = enc.encode_subframes(pcm16, n, want_samples=True)   # int32 rows requested
        exp = oracle.encode_subframes_batch(p, pcm16, n, slot_bytes=got["slot_bytes"])
        for g in (got, rows):
            assert_info_equal(g["info"], exp["info"], f"narrow16 n{n} om{om}")
            assert_residual_equal(g["residual"], exp["residual"], exp["info"], f"narrow16 n{n}")
            assert_bits_equal(g["rice_bits"], exp["rice_bits"], exp["info"], f"narrow16 n{n}")
        assert (got["info"]["reserved"] == 0).all()
    # 24-bit: wide as such, narrow when the low byte is all zeros (wasted bits)
    loud = (r.randint(-2 ** 22, 2 ** 22, (2, n, 2))).astype(np.int32)
    shifted = (r.randint(-20000, 20000, (2, n, 2)) << 8).astype(np.int32)
    pcm24 = np.concatenate([loud, shifted])
    p = flake_amd.level_params(5, bits_per_sample=24, block_size=n)
    with flake_amd.Encoder(p, max_frames=4) as enc:
        got = enc.encode_subframes(pcm24, n)
    exp = oracle.encode_subframes_batch(p, pcm24, n, slot_bytes=got["slot_bytes"])
    assert_info_equal(got["info"], exp["info"], f"narrow24 n{n}")
    assert_residual_equal(got["residual"], exp["residual"], exp["info"], f"narrow24 n{n}")
    assert_bits_equal(got["rice_bits"], exp["rice_bits"], exp["info"], f"narrow24 n{n}")
    assert (exp["info"]["wasted"][4:] >= 8).all()


@pytest.mark.parametrize("n", [192, 512, 1152, 2048, 4096, 4608])
@pytest.mark.parametrize("orders", [(0, 4), (1, 3), (2, 4), (0, 1), (3, 4), (2, 2)])
def test_fixed_order_and_partition_ranges(oracle, n, orders):
    """optimize.c:170-188 on the one-pass search over all fixed orders (and on the
    candidate loop it replaces when the partition orders go past 5): every order
    window, partition windows that clamp differently per order, edge signals whose
    warm-up samples dominate the first partition."""
    edges = edge_blocks(n, 16)
    mono = np.stack([edges[k] for k in sorted(edges)])[:, :, None]
    for (pmin, pmax) in ((0, 0), (0, 3), (2, 5), (5, 5), (0, 8)):
        kw = dict(prediction_type=flake_amd.PRED_FIXED, min_prediction_order=orders[0],
                  max_prediction_order=orders[1], min_partition_order=pmin, max_partition_order=pmax)
        p = flake_amd.level_params(5, channels=1, bits_per_sample=16, block_size=n, **kw)
        check(oracle, p, mono, n, f"fixed_mono_n{n}_{orders}_{pmin}-{pmax}")
        p = flake_amd.level_params(5, channels=2, bits_per_sample=24, block_size=n, **kw)
        pcm = flake_amd.synth_pcm(3, n, 2, 24, first_frame=7)
        check(oracle, p, pcm, n, f"fixed_stereo24_n{n}_{orders}_{pmin}-{pmax}")


@pytest.mark.parametrize("n", [16385, 20000, 24576, 32768, 40000, 65535])
def test_long_blocks(oracle, n):
    """Blocks above 16384 (outside FLAC's subset, inside libflake's 65535 limit,
    encode.h:35) take the streaming K0 and K3: stereo with every decorrelation mode in
    reach, mono 24-bit through an order search, a fixed-order range, wasted bits, and a
    constant subframe."""
    r = np.random.RandomState(n)
    t = np.arange(n)
    base = (9000 * np.sin(t * 0.013) + 2000 * np.sin(t * 0.31)).astype(np.int64)
    stereo = np.stack([
        np.stack([base + r.randint(-40, 40, n), base + r.randint(-40, 40, n)], 1),           # mid/side wins
        np.stack([base, r.randint(-3000, 3000, n)], 1),                                       # independent
        np.stack([(base >> 3) << 3, np.full(n, 1234)], 1),                                    # wasted bits, constant
    ]).astype(np.int32)
    for kw in (dict(order_method=flake_amd.OM_MAX),
               dict(order_method=flake_amd.OM_LOG, max_prediction_order=12, max_partition_order=8),
               dict(prediction_type=flake_amd.PRED_FIXED, min_prediction_order=0, max_prediction_order=4)):
        p = flake_amd.level_params(5, block_size=n, **kw)
        check(oracle, p, stereo, n, f"long stereo n{n}")
    mono = flake_amd.synth_pcm(2, n, 1, 24, first_frame=3)
    p = flake_amd.level_params(5, channels=1, bits_per_sample=24, block_size=n,
                               order_method=flake_amd.OM_4LEVEL, max_partition_order=8)
    check(oracle, p, mono, n, f"long mono24 n{n}")


@pytest.mark.parametrize("n", [512, 2048, 4096, 8192])
@pytest.mark.parametrize("order", [9, 12, 16])
def test_orders_9_to_16_on_16bit_rows(oracle, n, order):
    """MAX / EST with a maximum order of 9..16 on 16-bit stereo: K3's instance with the
    eight-pair packed FIR (and its fall-back to the fp64 FIR when sum|coef| * 2^magbits
    could leave int32: the full-scale square wave below)."""
    r = np.random.RandomState(n + order)
    t = np.arange(n)
    loud = np.stack([30000 * np.sign(np.sin(t * 0.3)), 28000 * np.sign(np.sin(t * 0.31))], 1)
    pcm = np.concatenate([flake_amd.synth_pcm(5, n, 2, 16, first_frame=order),
                          loud[None].astype(np.int32),
                          r.randint(-2000, 2000, (1, n, 2)).astype(np.int32)])
    for om in (flake_amd.OM_MAX, flake_amd.OM_EST):
        p = flake_amd.level_params(5, block_size=n, max_prediction_order=order, order_method=om)
        check(oracle, p, pcm, n, f"order{order} n{n} om{om}")


def test_log_walk_every_order_range(oracle):
    """LOG (optimize.c:239-261) for every (min, max) prediction-order pair: the search kernel
    merges consecutive steps of the walk into one round where their candidates do not depend on
    the winner; the order chosen (and everything after it) must be the reference's."""
    r = np.random.RandomState(77)
    for lo in range(1, 33):
        for hi in range(lo, 33):
            n = 4096 if (lo + hi) % 5 == 0 else 2048
            bps = 24 if (lo * 7 + hi) % 3 == 0 else 16
            p = flake_amd.level_params(5, bits_per_sample=bps, block_size=n, order_method=flake_amd.OM_LOG,
                                       min_prediction_order=lo, max_prediction_order=hi, max_partition_order=6)
            pcm = flake_amd.synth_pcm(2, n, 2, bps, first_frame=int(r.randint(0, 1000)))
            check(oracle, p, pcm, n, f"LOG {lo}..{hi} n={n} bps={bps}")


@pytest.mark.parametrize("n", [4096, 8192, 16384, 2048, 512, 1024, 1536, 2560, 3072, 3584, 5120, 6144, 7168])
@pytest.mark.parametrize("om,mo", [(flake_amd.OM_SEARCH, 32), (flake_amd.OM_SEARCH, 9), (flake_amd.OM_8LEVEL, 32),
                                   (flake_amd.OM_4LEVEL, 12), (flake_amd.OM_2LEVEL, 8), (flake_amd.OM_LOG, 32)])
def test_order_search_kernel(oracle, n, om, mo):
    """k_order_search (the LEVEL / SEARCH / LOG walks over a table of candidates): rounds of up
    to four candidates, a wave per candidate from the thread sums (leaf mode), incl. the
    fall-back to the 64-bit pyramid and node pass of every wave -- 32-bit noise leaves thread
    sums above 32 bits -- and LOG's merged steps.  Mixed batch: resonator frames, full-scale
    32-bit noise, a constant frame, a frame whose first tile is an impulse (warm-up masking)."""
    for bps in (24, 32, 16):
        p = flake_amd.level_params(5, bits_per_sample=bps, block_size=n, order_method=om,
                                   min_prediction_order=1 if om != flake_amd.OM_8LEVEL else 3,
                                   max_prediction_order=mo, max_partition_order=8)
        r = np.random.RandomState(n + mo + bps)
        full = 1 << (bps - 1)
        pcm = flake_amd.synth_pcm(5, n, 2, bps, first_frame=7).astype(np.int64)
        pcm[1] = r.randint(-full, full, (n, 2))                       # noise: the fall-back for 32 bits
        pcm[2] = 1234
        pcm[3, :16] = 0
        pcm[3, 3] = full - 1
        pcm[4, :, 1] = pcm[4, :, 0] >> 1
        check(oracle, p, pcm.astype(np.int32), n, f"order search n={n} om={om} mo={mo} bps={bps}")


def _many_tones(nframes, n, ch, bps, seed):
    """Fifteen sinusoids and a little noise: prediction keeps gaining up to order ~30, so the
    order searches end on the highest orders."""
    r = np.random.RandomState(seed)
    t = np.arange(nframes * n, dtype=np.float64)
    x = np.zeros((nframes * n, ch))
    for c in range(ch):
        for _ in range(15):
            x[:, c] += r.uniform(0.3, 1.0) * np.sin(t * r.uniform(0.02, 3.0) + r.uniform(0, 6.28))
    x *= (1 << (bps - 1)) / 16.0
    x += r.uniform(-2, 2, x.shape)
    return np.round(x).astype(np.int32).reshape(nframes, n, ch)


@pytest.mark.parametrize("n", [4096, 8192, 2560, 3584, 5120, 6144, 7168, 1024])
@pytest.mark.parametrize("om,lo,hi,pmin", [(flake_amd.OM_SEARCH, 1, 32, 0), (flake_amd.OM_8LEVEL, 10, 29, 8),
                                            (flake_amd.OM_4LEVEL, 10, 29, 8), (flake_amd.OM_8LEVEL, 3, 32, 0),
                                            (flake_amd.OM_LOG, 1, 32, 0), (flake_amd.OM_SEARCH, 20, 31, 7)])
def test_order_search_high_orders(oracle, n, om, lo, hi, pmin):
    """The order searches where the HIGH orders win (a signal of many tones): every run length's FIR at
    taps 29 .. 32, partition-order windows clamped below the leaves' level (n / order < 2^8), ragged
    order ranges.  (Round 3 fuzz, seed 3000359: runs of 20 / 28 read candidate rows past tap 32.)"""
    top = 0
    for bps in (24, 16):
        p = flake_amd.level_params(5, bits_per_sample=bps, block_size=n, order_method=om,
                                   min_prediction_order=lo, max_prediction_order=hi,
                                   min_partition_order=pmin, max_partition_order=8)
        pcm = _many_tones(3, n, 2, bps, n + hi + bps)
        _, exp = check(oracle, p, pcm, n, f"high orders n={n} om={om} {lo}..{hi} pmin={pmin} bps={bps}")
        top = max(top, int(exp["info"]["order"].max()))
    assert top >= min(hi, 27), top              # the case did reach up there


@pytest.mark.parametrize("n", [512, 1024, 1536, 2048, 2560, 3072, 3584, 4096, 5120, 6144, 7168, 8192, 16384])
@pytest.mark.parametrize("om,lo,hi", [(flake_amd.OM_SEARCH, 1, 32), (flake_amd.OM_SEARCH, 1, 12),
                                       (flake_amd.OM_8LEVEL, 3, 32), (flake_amd.OM_LOG, 1, 32)])
def test_order_search_table(oracle, n, om, lo, hi):
    """fhip_order_search_bits: EVERY entry of the bits[] table behind the order searches (optimize.c:201-261),
    not only the winner -- quantised rows (lpc.c:224-257), FIR (optimize.c:70-122) and
    calc_rice_params_lpc (rice.c:180-187) per visited order from the oracle.  Widths: 16 bits with the
    feeder's magnitude record (packed FIRs; SEARCH above order 16: the matrix pipe takes the rest),
    16 bits without it, 24 bits; resonator frames, many tones (high orders win), noise."""
    for bps, with_mag in ((16, True), (16, False), (24, False)):
        p = flake_amd.level_params(5, bits_per_sample=bps, block_size=n, order_method=om,
                                   min_prediction_order=lo, max_prediction_order=hi, max_partition_order=8)
        r = np.random.RandomState(n * 3 + hi + bps)
        full = 1 << (bps - 1)
        blocks = np.concatenate([flake_amd.synth_pcm(2, n, 1, bps, first_frame=11).reshape(2, n),
                                 _many_tones(1, n, 1, bps, n + bps).reshape(1, n),
                                 r.randint(-full, full, (1, n)).astype(np.int32)])
        blocks[1, : n // 3] >>= 5
        obits = np.full(blocks.shape[0], bps, dtype=np.int32)
        mag = None
        if with_mag:
            mag = [max(1, int(np.abs(b.astype(np.int64)).max()).bit_length()) for b in blocks]
        with flake_amd.Encoder(p, max_frames=blocks.shape[0]) as enc:
            table = enc.order_search_bits(blocks, obits, magbits=mag)
        visited = 0
        for s in range(blocks.shape[0]):
            coefs, shift, _ = oracle.lpc_calc_coefs(blocks[s], hi, p.lpc_precision, om)
            for o in range(1, hi + 1):
                if table[s, o - 1] == 0xFFFFFFFF:
                    continue
                res = oracle.residual_lpc(blocks[s], o, coefs[o - 1], int(shift[o - 1]))
                bits, _ = oracle.subframe_bits(res, p.min_partition_order, p.max_partition_order, o, bps,
                                               p.lpc_precision, True)
                assert int(table[s, o - 1]) == int(bits) & 0xFFFFFFFF, (n, om, bps, with_mag, s, o, table[s, o - 1], bits)
                visited += 1
            if om == flake_amd.OM_SEARCH:
                assert (table[s, :hi] != 0xFFFFFFFF).all(), (n, bps, s)       # every order 1 .. max (optimize.c:226)
        assert visited >= blocks.shape[0] * 3
