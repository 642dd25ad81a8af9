"""Whole frames from the oracle decode back to the input through an independent
FLAC decoder (oracle/flac_decode.c): header, CRC-8, subframes, Rice sections,
stereo modes, wasted bits, verbatim fallback, CRC-16, VBS.  This is the check of
the oracle parts that cannot be pinned against a reference binary."""
import numpy as np
import pytest

import flake_amd
from cases import edge_blocks, param_sets, stereo_frames, _rng


@pytest.mark.parametrize("name,p,n", param_sets(), ids=[c[0] for c in param_sets()])
def test_param_sets_roundtrip(oracle, decoder, name, p, n):
    n = min(n, 2048)
    q = p.copy()
    q.block_size = n
    pcm = flake_amd.synth_pcm(3, n, p.channels, p.bits_per_sample, first_frame=9)
    stream = []
    for f in range(3):
        rc, fb, sf, _, _ = oracle.encode_frame(q, f, pcm[f], n)
        assert rc > 0
        stream.append(fb)
    out, sizes = decoder.decode(np.concatenate(stream), p.channels, p.bits_per_sample, 3 * n)
    assert (sizes == n).all()
    assert (out.reshape(3, n, p.channels) == pcm).all(), name


@pytest.mark.parametrize("bps", [16, 24])
def test_stereo_edges_roundtrip(oracle, decoder, bps):
    fr = stereo_frames(1024, bps)
    p = flake_amd.level_params(5, bits_per_sample=bps, block_size=1024)
    for i, (k, pcm) in enumerate(sorted(fr.items())):
        rc, fb, sf, _, verb = oracle.encode_frame(p, i, pcm, 1024)
        out, _ = decoder.decode(fb, 2, bps, 1024)
        assert (out == pcm).all(), k


def test_white_noise_takes_the_verbatim_fallback(oracle, decoder):
    """encode.c:949-964: a frame larger than its verbatim size is re-emitted verbatim.
    Full-scale mono white noise costs >= 16.5 bits/sample in Rice codes vs 16 raw."""
    r = _rng(4)
    pcm = r.randint(-32768, 32768, (4096, 1)).astype(np.int32)
    p = flake_amd.level_params(5, channels=1)
    rc, fb, sf, _, verb = oracle.encode_frame(p, 0, pcm, 4096)
    assert verb == 1 and (sf["type"] == 1).all()
    assert rc <= 16 + ((4096 * 16 + 7) >> 3)
    out, _ = decoder.decode(fb, 1, 16, 4096)
    assert (out == pcm).all()


@pytest.mark.parametrize("bps", [8, 16, 24, 32])
def test_mono_edges_roundtrip(oracle, decoder, bps):
    """At 32 bits the reference's int32 residual (optimize.c:120) cannot hold a
    full-scale prediction error, so libflake itself is lossy there; the 32-bit
    round trip is therefore checked on 28-bit material only."""
    p = flake_amd.level_params(8, channels=1, bits_per_sample=bps, block_size=1024)
    for k, b in edge_blocks(1024, min(bps, 28)).items():
        rc, fb, sf, _, _ = oracle.encode_frame(p, 3, b[:, None], 1024)
        out, _ = decoder.decode(fb, 1, bps, 1024)
        assert (out[:, 0] == b).all(), (bps, k)


@pytest.mark.parametrize("n", [1, 2, 4, 5, 16, 17, 100, 192, 1152, 4608])
def test_ragged_sizes_roundtrip(oracle, decoder, n):
    p = flake_amd.level_params(5, block_size=max(n, 16))
    pcm = flake_amd.synth_pcm(1, n, 2, 16, first_frame=n)[0]
    rc, fb, _, _, _ = oracle.encode_frame(p, 70000, pcm, n)       # 3-byte UTF-8 frame number
    out, sizes = decoder.decode(fb, 2, 16, n)
    assert sizes[0] == n and (out == pcm).all()


def test_vbs_blocks_roundtrip(oracle, decoder):
    """vbs.c: a block with a loud second half splits; the pieces decode to the block."""
    p = flake_amd.level_params(10)
    n = p.block_size
    base = flake_amd.synth_pcm(2, n, 2, 16)
    quiet = base[0] // 64
    loud = base[1]
    blk = np.concatenate([quiet[: n // 2], loud[n // 2:]]).astype(np.int32)
    nf, sizes = oracle.vbs_split(blk, 2, n)
    assert nf > 1 and sizes.sum() == n and (sizes % (n // 8) == 0).all()
    rc, data, fc = oracle.encode_block(p, 0, blk, n, 4 * n * 4)
    assert rc > 0 and fc == n                      # allow_vbs: the counter advances in samples
    out, got_sizes = decoder.decode(data, 2, 16, n)
    assert (got_sizes == sizes).all() and (out == blk).all()
    # a stationary block does not split and falls back to one frame
    t = np.arange(n)
    tone = np.round(8000 * np.sin(t * 0.05)).astype(np.int32)
    still = np.stack([tone, tone // 2], axis=1) + _rng(1).randint(-50, 51, (n, 2)).astype(np.int32)
    nf1, sizes1 = oracle.vbs_split(still, 2, n)
    assert nf1 == 1 and sizes1[0] == n
    rc1, data1, _ = oracle.encode_block(p, 0, still, n, 4 * n * 4)
    out1, s1 = decoder.decode(data1, 2, 16, n)
    assert len(s1) == 1 and (out1 == still).all()
