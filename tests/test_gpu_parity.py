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
