"""K4: whole frames assembled on the device (header + CRC-8, subframes, residual
sections, CRC-16, verbatim fallback) vs the oracle's encode_frame(), byte for byte."""
import numpy as np
import pytest

import flake_amd
from cases import param_sets, stereo_frames, edge_blocks, _rng

pytestmark = pytest.mark.gpu


def check_frames(oracle, decoder, p, pcm, n, first=0, what=""):
    nfr = pcm.shape[0]
    with flake_amd.Encoder(p, max_frames=nfr) as enc:
        got = enc.encode_subframes(pcm, n, want_residual=False, want_frames=True,
                                   first_frame_number=first)
    step = n if p.allow_vbs else 1
    stream = []
    for f in range(nfr):
        rc, exp, _, _, _ = oracle.encode_frame(p, first + f * step, pcm[f], n)
        nb = int(got["frame_bytes"][f])
        assert nb == rc, (what, f, nb, rc)
        frame = got["frames"][f, :nb]
        bad = np.nonzero(frame != exp)[0]
        assert bad.size == 0, (what, f, "first differing byte", int(bad[0]) if bad.size else -1)
        stream.append(frame)
    if p.bits_per_sample <= 24:
        out, sizes = decoder.decode(np.concatenate(stream), p.channels, p.bits_per_sample, nfr * n)
        assert (out.reshape(nfr, n, p.channels) == pcm.reshape(nfr, n, p.channels)).all(), what


@pytest.mark.parametrize("name,p,n", param_sets(), ids=[c[0] for c in param_sets()])
def test_param_sets(oracle, decoder, name, p, n):
    nfr = 5 if p.channels <= 2 else 2
    pcm = flake_amd.synth_pcm(nfr, n, p.channels, p.bits_per_sample, first_frame=17)
    check_frames(oracle, decoder, p, pcm, n, first=126, what=name)      # crosses the 1-/2-byte UTF-8 edge


def test_stereo_edges_and_verbatim_fallback(oracle, decoder):
    fr = stereo_frames(4096, 16)
    r = _rng(12)
    frames = [fr[k] for k in sorted(fr)]
    frames.append(r.randint(-32768, 32768, (4096, 2)).astype(np.int32))   # white noise
    pcm = np.stack(frames)
    p = flake_amd.level_params(5)
    check_frames(oracle, decoder, p, pcm, 4096, first=70000, what="stereo edges")   # 3-byte numbers


def test_mono_verbatim_and_constant(oracle, decoder):
    b = edge_blocks(4096, 16)
    pcm = np.stack([b["white"], b["zeros"], b["dc"], b["wasted_3"], b["alt_full_scale"], b["sine"]])[:, :, None]
    p = flake_amd.level_params(5, channels=1)
    check_frames(oracle, decoder, p, pcm, 4096, first=0, what="mono edges")


@pytest.mark.parametrize("n", [16, 100, 192, 255, 1152, 4608, 5000])
def test_ragged_sizes(oracle, decoder, n):
    p = flake_amd.level_params(5, block_size=max(n, 16))
    pcm = flake_amd.synth_pcm(4, n, 2, 16, first_frame=n)
    check_frames(oracle, decoder, p, pcm, n, first=2 ** 21 - 2, what=f"n{n}")       # 4-byte numbers


def test_allow_vbs_numbers_count_samples(oracle, decoder):
    p = flake_amd.level_params(9, variable_block_size=0)
    n = p.block_size
    pcm = flake_amd.synth_pcm(4, n, 2, 16)
    check_frames(oracle, decoder, p, pcm, n, first=3 * n, what="allow_vbs")


def test_full_batch_crc_and_decode(decoder):
    """configs[1] at 1024 frames: every frame decodes (CRC-8/CRC-16 verified by the decoder)."""
    p = flake_amd.level_params(5, order_method=flake_amd.OM_MAX)
    pcm = flake_amd.synth_pcm(1024, 4096, 2, 16)
    with flake_amd.Encoder(p, max_frames=1024) as enc:
        got = enc.encode_subframes(pcm, 4096, want_residual=False, want_frames=True)
    stream = np.concatenate([got["frames"][f, :got["frame_bytes"][f]] for f in range(1024)])
    out, sizes = decoder.decode(stream, 2, 16, 1024 * 4096)
    assert len(sizes) == 1024 and (out.reshape(pcm.shape) == pcm).all()


@pytest.mark.parametrize("n", [20000, 32768, 65535])
def test_long_block_frames(oracle, decoder, n):
    """Frames of blocks above 16384 (libflake allows 65535): 16-bit block-size field in the
    header, long residual sections, CRC over up to 260 KB."""
    p = flake_amd.level_params(5, block_size=n)
    pcm = flake_amd.synth_pcm(3, n, 2, 16, first_frame=5)
    check_frames(oracle, decoder, p, pcm, n, first=7, what=f"long n{n}")
    p = flake_amd.level_params(2, channels=1, bits_per_sample=24, block_size=n)
    pcm = flake_amd.synth_pcm(2, n, 1, 24, first_frame=9)
    check_frames(oracle, decoder, p, pcm, n, first=0, what=f"long mono24 n{n}")
