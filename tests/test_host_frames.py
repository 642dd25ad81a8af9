"""The host C layer on top of the HIP ABI: complete FLAC frames.

libflake's call sequence (flake_set_defaults -> validate -> encode_init ->
encode_frame(s) -> close) through include/flake_amd.h.  The frames must equal
the oracle's encode_frame()/encode_block() byte for byte, decode back to the
input through the independent decoder, and the STREAMINFO MD5 must be the MD5 of
the raw little-endian PCM (md5.c:281-320)."""
import hashlib

import numpy as np
import pytest

import flake_amd
from cases import stereo_frames, _rng

pytestmark = pytest.mark.gpu


def oracle_stream(oracle, p, pcm, block_size, tail=0):
    """What flake_encode_frame() would write for consecutive blocks (+ short tail)."""
    ch = p.channels
    nblocks = (pcm.shape[0] - tail) // block_size
    out, sizes, fc = [], [], 0
    for b in range(nblocks + (1 if tail else 0)):
        n = block_size if b < nblocks else tail
        blk = pcm[b * block_size: b * block_size + n]
        rc, data, fc = oracle.encode_block(p, fc, blk, n, 8 * n * ch * 4 + 4096)
        assert rc > 0
        out.append(data)
        sizes.append(rc)
    return np.concatenate(out), np.array(sizes)


def pcm_md5(pcm, bps):
    nbytes = (bps + 7) // 8
    raw = pcm.astype("<i4").reshape(-1).view(np.uint8).reshape(-1, 4)[:, :nbytes]
    return hashlib.md5(raw.tobytes()).digest()


@pytest.mark.parametrize("level", [0, 2, 5, 8])
def test_levels_stream_equals_oracle(oracle, decoder, level):
    with flake_amd.HostEncoder(level) as enc:
        p = enc.params()
        n = p.block_size
        pcm = flake_amd.synth_pcm(9, n, 2, 16).reshape(-1, 2)
        tail = n // 3
        pcm = np.concatenate([pcm, flake_amd.synth_pcm(1, tail, 2, 16, first_frame=99)[0]])
        data, sizes = enc.encode_frames(pcm, n, tail)
        exp, esizes = oracle_stream(oracle, p, pcm, n, tail)
        assert (sizes == esizes).all()
        assert data.tobytes() == exp.tobytes()
        out, bs = decoder.decode(data, 2, 16, pcm.shape[0])
        assert (out == pcm).all() and bs[-1] == tail
        si = enc.streaminfo()
        assert bytes(si.md5sum) == pcm_md5(pcm, 16)
        assert si.max_frame_size >= sizes.max()
        assert enc.header[:4] == b"fLaC"


@pytest.mark.parametrize("level", [9, 10, 12])
def test_vbs_levels(oracle, decoder, level):
    """vbs.c: blocks split into ragged frames; frame numbers count samples."""
    with flake_amd.HostEncoder(level) as enc:
        p = enc.params()
        n = p.block_size
        base = flake_amd.synth_pcm(6, n, 2, 16)
        blocks = []
        for b in range(6):
            blk = base[b].copy()
            if b % 2 == 0:                      # quiet first part, loud rest: forces a split
                cut = (1 + b) * n // 8
                blk[:cut] //= 64
            blocks.append(blk)
        pcm = np.concatenate(blocks)
        data, sizes = enc.encode_frames(pcm, n)
        exp, esizes = oracle_stream(oracle, p, pcm, n)
        assert (sizes == esizes).all()
        assert data.tobytes() == exp.tobytes()
        out, bs = decoder.decode(data, 2, 16, pcm.shape[0])
        assert (out == pcm).all()
        assert len(bs) > 6 and (bs % (n // 8) == 0).all()


@pytest.mark.parametrize("level,nblocks", [(10, 3072), (12, 2048)])
def test_vbs_large_corpus_properties(oracle, decoder, level, nblocks):
    """BASELINE configs[4] at a corpus size the oracle cannot follow in test time (33.5 M samples
    at level 12): size-independent properties of the whole stream -- it decodes to the input
    (independent decoder), STREAMINFO carries the input's MD5 (the sample count is the caller's to
    fill in, as in libflake), every frame is a
    multiple of n/8 samples, split blocks exist -- plus byte equality with the oracle's stream on
    the corpus' first blocks encoded as a stream of their own (VBS frame numbers count samples
    from the stream's start, so a prefix of the corpus is a stream the oracle can check)."""
    with flake_amd.HostEncoder(level) as enc:
        p = enc.params()
        n = p.block_size
        pcm = flake_amd.synth_pcm(nblocks, n, 2, 16)
        for b in range(0, nblocks, 3):          # every third block: a quiet first part, loud rest
            cut = (1 + b % 7) * n // 8
            pcm[b, :cut] //= 64
        pcm = pcm.reshape(-1, 2)
        data, sizes = enc.encode_frames(pcm, n)
        si = enc.streaminfo()
        assert bytes(si.md5sum) == pcm_md5(pcm, 16)
        assert sizes.sum() == data.size and si.max_frame_size >= sizes.max()
        out, bs = decoder.decode(data, 2, 16, pcm.shape[0])
        assert (out == pcm).all()
        assert bs.sum() == pcm.shape[0] and (bs % (n // 8) == 0).all()
        assert len(bs) > nblocks + nblocks // 6                 # the forced splits happened
    npre = 6
    with flake_amd.HostEncoder(level) as enc:
        pre, psizes = enc.encode_frames(pcm[:npre * n], n)
        exp, esizes = oracle_stream(oracle, enc.params(), pcm[:npre * n], n)
        assert (psizes == esizes).all() and pre.tobytes() == exp.tobytes()
    assert data[:pre.size].tobytes() == pre.tobytes()            # the corpus starts with that stream


def test_verbatim_fallback_and_constant(oracle, decoder):
    r = _rng(4)
    n = 4096
    noise = r.randint(-32768, 32768, (n, 1)).astype(np.int32)
    silence = np.zeros((n, 1), np.int32)
    tone = flake_amd.synth_pcm(1, n, 1, 16)[0]
    pcm = np.concatenate([noise, silence, tone])
    with flake_amd.HostEncoder(5, channels=1) as enc:
        data, sizes = enc.encode_frames(pcm, n)
        exp, esizes = oracle_stream(oracle, enc.params(), pcm, n)
        assert data.tobytes() == exp.tobytes()
        assert sizes[0] == 16 + n * 2 - 5 or sizes[0] <= 16 + n * 2     # verbatim-sized
        assert sizes[1] < 16                                            # CONSTANT subframe
        out, _ = decoder.decode(data, 1, 16, 3 * n)
        assert (out == pcm).all()


def test_stereo_edge_frames_24bit(oracle, decoder):
    fr = stereo_frames(4096, 24)
    pcm = np.concatenate([fr[k] for k in sorted(fr)])
    with flake_amd.HostEncoder(5, bits_per_sample=24, sample_rate=96000) as enc:
        data, sizes = enc.encode_frames(pcm, 4096)
        exp, _ = oracle_stream(oracle, enc.params(), pcm, 4096)
        assert data.tobytes() == exp.tobytes()
        out, _ = decoder.decode(data, 2, 24, pcm.shape[0])
        assert (out == pcm).all()
        assert bytes(enc.streaminfo().md5sum) == pcm_md5(pcm, 24)


def test_eight_channels_192k(oracle, decoder):
    """BASELINE configs[3]: 8 channels, 24 bit, 192 kHz (sample-rate code 12 + 8-bit kHz)."""
    pcm = flake_amd.synth_pcm(3, 4096, 8, 24).reshape(-1, 8)
    with flake_amd.HostEncoder(5, channels=8, bits_per_sample=24, sample_rate=192000,
                               order_method=flake_amd.OM_MAX, max_prediction_order=12) as enc:
        data, sizes = enc.encode_frames(pcm, 4096)
        exp, _ = oracle_stream(oracle, enc.params(), pcm, 4096)
        assert data.tobytes() == exp.tobytes()
        out, _ = decoder.decode(data, 8, 24, pcm.shape[0])
        assert (out == pcm).all()


def test_single_frame_entry_and_last_block_latch(oracle):
    """flake_encode_frame(): one block per call; a short block ends the stream
    (encode.c:989-994)."""
    with flake_amd.HostEncoder(5) as enc:
        p = enc.params()
        pcm = flake_amd.synth_pcm(3, 4096, 2, 16)
        fc = 0
        for b in range(2):
            got = enc.encode_frame(pcm[b])
            rc, exp, fc = oracle.encode_block(p, fc, pcm[b], 4096, 1 << 16)
            assert got == exp.tobytes()
        short = enc.encode_frame(pcm[2][:1000])
        rc, exp, fc = oracle.encode_block(p, fc, pcm[2][:1000], 1000, 1 << 16)
        assert short == exp.tobytes()
        with pytest.raises(flake_amd.FlakeHipError):
            enc.encode_frame(pcm[0])


def test_invalid_parameters_are_rejected_on_the_host():
    with pytest.raises(ValueError):
        flake_amd.HostEncoder(5, channels=9)
    with pytest.raises(ValueError):
        flake_amd.HostEncoder(5, variable_block_size=1)          # needs allow_vbs
    with pytest.raises(ValueError):
        flake_amd.HostEncoder(13)


def test_cli_wav_to_flac(tmp_path, decoder):
    """The C command-line harness (flake/flake.c's loop on the host API): WAV in,
    a complete .flac out -- stream marker, STREAMINFO with the final MD5, frames
    that decode to the input."""
    import os
    import struct
    import subprocess
    cli = os.path.join(flake_amd.LIB_DIR, "flake_amd_cli")
    if not os.path.exists(cli):
        pytest.skip("flake_amd_cli not built")
    n_total = 4096 * 5 + 1234
    pcm = flake_amd.synth_pcm(6, 4096, 2, 16).reshape(-1, 2)[:n_total]
    raw = pcm.astype("<i2").tobytes()
    wav = tmp_path / "in.wav"
    with open(wav, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE" + b"fmt " +
                struct.pack("<IHHIIHH", 16, 1, 2, 44100, 44100 * 4, 4, 16) + b"data" +
                struct.pack("<I", len(raw)) + raw)
    out = tmp_path / "out.flac"
    r = subprocess.run([cli, "-5", str(wav), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    data = np.fromfile(out, dtype=np.uint8)
    assert data[:4].tobytes() == b"fLaC"
    # walk the metadata blocks to the first frame
    pos, last = 4, 0
    md5 = None
    while not last:
        hdr = data[pos:pos + 4]
        last, typ = hdr[0] >> 7, hdr[0] & 0x7F
        ln = (int(hdr[1]) << 16) | (int(hdr[2]) << 8) | int(hdr[3])
        if typ == 0:
            md5 = data[pos + 4 + 18: pos + 4 + 34].tobytes()
        pos += 4 + ln
    dec, sizes = decoder.decode(data[pos:], 2, 16, n_total)
    assert (dec == pcm).all() and sizes[-1] == 1234
    assert md5 == hashlib.md5(raw).digest()


def test_link_level_dropin_client(tmp_path, decoder, oracle):
    """oracle/_ref/dropin_client is compiled against the REFERENCE's flake.h and linked
    to our libflake.so (libflake's own symbol names): same call sequence as the
    reference CLI, one flake_encode_frame() per block."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref",
                       "dropin_client")
    if not os.path.exists(exe):
        pytest.skip("dropin_client not built (needs /root/reference at build time)")
    out = tmp_path / "d.flac"
    r = subprocess.run([exe, "5", "7", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    data = np.fromfile(out, dtype=np.uint8)
    pos, last = 4, 0
    while not last:
        last = data[pos] >> 7
        pos += 4 + ((int(data[pos + 1]) << 16) | (int(data[pos + 2]) << 8) | int(data[pos + 3]))
    pcm = flake_amd.synth_pcm(7, 4096, 2, 16).reshape(-1, 2)
    dec, sizes = decoder.decode(data[pos:], 2, 16, 7 * 4096)
    assert (dec == pcm).all() and len(sizes) == 7
    p = flake_amd.level_params(5)
    exp, _ = oracle_stream(oracle, p, pcm, 4096)
    assert data[pos:].tobytes() == exp.tobytes()


@pytest.mark.parametrize("level,nblocks,look", [(5, 50, 16), (8, 33, 64), (10, 21, 8)])
def test_lookahead_queue_is_transparent(tmp_path, level, nblocks, look):
    """FLAKE_AMD_LOOKAHEAD=N: the unmodified one-block-per-call client gets GPU batches;
    the stream it writes is byte-identical to the unqueued one (frame numbers, MD5,
    STREAMINFO frame sizes included).  nblocks is not a multiple of N, so the last
    flush comes from the announced stream length; level 10 is variable block size."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref",
                       "dropin_client")
    if not os.path.exists(exe):
        pytest.skip("dropin_client not built (needs /root/reference at build time)")
    outs = []
    for tag, env in (("plain", {}), ("queued", {"FLAKE_AMD_LOOKAHEAD": str(look)})):
        out = tmp_path / f"{tag}.flac"
        r = subprocess.run([exe, str(level), str(nblocks), str(out)], capture_output=True, text=True,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr + r.stdout
        outs.append(np.fromfile(out, dtype=np.uint8))
    assert outs[0].size > 1000 and outs[0].tobytes() == outs[1].tobytes()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("FLAKE_FUZZ_HOST_SEEDS", "48"))))
def test_random_streams(oracle, decoder, seed):
    """Seeded sweep through the host layer: level, channel count, bit depth, block
    size, variable block size on or off, a random number of blocks and a random short
    tail -- the byte stream against the oracle's block loop, the decoder and the MD5."""
    r = np.random.RandomState(9000 + seed)
    level = int(r.choice([0, 1, 2, 3, 5, 5, 6, 7, 8, 9, 10, 12]))
    ch = int(r.choice([1, 2, 2, 2, 3, 6]))
    bps = int(r.choice([8, 16, 16, 20, 24]))
    over = {}
    if r.rand() < 0.5:
        over["block_size"] = int(r.choice([256, 512, 1024, 1152, 2048, 4096, 4608]))
    if r.rand() < 0.3:
        over["variable_block_size"] = int(r.randint(0, 2))
        if over["variable_block_size"]:
            over["allow_vbs"] = 1                        # encode.c:363: the splitter needs it
    with flake_amd.HostEncoder(level, channels=ch, bits_per_sample=bps, **over) as enc:
        p = enc.params()
        n = p.block_size
        nblocks = int(r.randint(1, 7)) if n * ch <= 16384 else int(r.randint(1, 4))
        tail = int(r.choice([0, 0, 1, 7, n // 3, n - 1]))
        kind = int(r.randint(0, 3))
        total = nblocks * n + tail
        if kind == 0:
            pcm = flake_amd.synth_pcm(1, total, ch, bps, first_frame=seed)[0]
        elif kind == 1:
            full = 1 << (bps - 1)
            pcm = r.randint(-full, full, (total, ch)).astype(np.int32)
        else:
            t = np.arange(total)[:, None]
            full = 1 << (bps - 1)
            pcm = (0.6 * full * np.sin(t * r.uniform(0.002, 0.2)) + r.randint(-4, 5, (total, ch))).astype(np.int32)
            pcm[total // 2:] //= 64                      # a level change for the VBS splitter
        what = f"stream seed {seed}: level {level} ch {ch} bps {bps} n {n} blocks {nblocks} tail {tail} vbs {p.variable_block_size}"
        data, sizes = enc.encode_frames(pcm, n, tail)
        exp, esizes = oracle_stream(oracle, p, pcm, n, tail)
        assert (sizes == esizes).all(), what
        assert data.tobytes() == exp.tobytes(), what
        out, _ = decoder.decode(data, ch, bps, total)
        assert (out == pcm).all(), what
        assert bytes(enc.streaminfo().md5sum) == pcm_md5(pcm, bps), what


@pytest.mark.parametrize("n", [24000, 65535])
def test_long_blocks_through_the_host_layer(oracle, decoder, n):
    with flake_amd.HostEncoder(5, block_size=n) as enc:
        p = enc.params()
        pcm = flake_amd.synth_pcm(1, 3 * n + 1000, 2, 16, first_frame=2)[0]
        data, sizes = enc.encode_frames(pcm, n, 1000)
        exp, esizes = oracle_stream(oracle, p, pcm, n, 1000)
        assert (sizes == esizes).all()
        assert data.tobytes() == exp.tobytes()
        out, _ = decoder.decode(data, 2, 16, pcm.shape[0])
        assert (out == pcm).all()
        assert bytes(enc.streaminfo().md5sum) == pcm_md5(pcm, 16)


def test_pinned_staging_is_transparent(monkeypatch):
    """flake_amd_pin_buffers(): the caller's PCM and output ranges page-locked in place.  The stream is the
    pageable one byte for byte, batch after batch through the same buffers, after a release, and with the
    look-ahead queue's own page-locked buffers (FLAKE_AMD_LOOKAHEAD)."""
    import ctypes as C
    n, nblocks = 1152, 260
    pcm = np.ascontiguousarray(flake_amd.synth_pcm(nblocks, n, 2, 16, first_frame=5).reshape(-1, 2), dtype=np.int32)
    monkeypatch.setenv("FLAKE_AMD_BATCH", "4096")
    monkeypatch.setenv("FLAKE_AMD_CHUNK", "64")
    monkeypatch.setenv("FLAKE_AMD_PIN", "0")
    with flake_amd.HostEncoder(5, block_size=n) as enc:
        ref, ref_sizes = enc.encode_frames(pcm, n, 0)
        assert enc.lib.flake_amd_pin_buffers(C.byref(enc.ctx), pcm.ctypes.data, pcm.nbytes, None, 0) == 0   # off: a no-op
    monkeypatch.delenv("FLAKE_AMD_PIN")
    cap = 64 + pcm.size * 5
    out = np.zeros(cap, dtype=np.uint8)
    sizes = np.zeros(nblocks, dtype=np.int32)
    for rep in range(3):                              # the same buffers batch after batch, stream after stream
        with flake_amd.HostEncoder(5, block_size=n) as enc:
            assert enc.lib.flake_amd_pin_buffers(C.byref(enc.ctx), pcm.ctypes.data, pcm.nbytes, out.ctypes.data, cap) == 0
            out[:] = 0
            w = enc.lib.flake_amd_encode_frames(C.byref(enc.ctx), pcm.ctypes.data, nblocks, n, 0, out.ctypes.data, cap,
                                                sizes.ctypes.data)
            assert w == ref.size and (sizes == ref_sizes).all()
            assert out[:w].tobytes() == ref.tobytes()
            if rep == 1:                               # released: the next batch runs from pageable memory again
                assert enc.lib.flake_amd_pin_buffers(C.byref(enc.ctx), pcm.ctypes.data, 0, out.ctypes.data, 0) == 0
            w2 = enc.lib.flake_amd_encode_frames(C.byref(enc.ctx), pcm.ctypes.data, nblocks, n, 0, out.ctypes.data, cap,
                                                 sizes.ctypes.data)
            assert w2 >= w                             # (the frame numbers went on counting: longer headers)
    # the look-ahead queue: the library's own page-locked buffers
    monkeypatch.setenv("FLAKE_AMD_LOOKAHEAD", "64")
    got = bytearray()
    with flake_amd.HostEncoder(5, block_size=n, samples=nblocks * n) as enc:
        for b in range(nblocks):
            got += enc.encode_frame(pcm[b * n:(b + 1) * n])
    assert bytes(got) == ref.tobytes()


def test_chunked_two_handle_batches_equal_the_single_pass(monkeypatch, decoder):
    """Large uniform batches run in chunks through two handles on two host threads
    (run_chunked): the stream must be the single-pass stream byte for byte, sizes included,
    whatever the chunk size (ragged last chunk, odd and even chunk counts)."""
    n, nblocks = 576, 301
    pcm = flake_amd.synth_pcm(nblocks, n, 2, 16, first_frame=11).reshape(-1, 2)
    monkeypatch.setenv("FLAKE_AMD_BATCH", "4096")
    monkeypatch.setenv("FLAKE_AMD_CHUNK", "0")
    with flake_amd.HostEncoder(5, block_size=n) as enc:
        ref, ref_sizes = enc.encode_frames(pcm, n, 0)
        ref_md5 = bytes(enc.streaminfo().md5sum)
    for chunk in (32, 50, 100, 150):
        monkeypatch.setenv("FLAKE_AMD_CHUNK", str(chunk))
        with flake_amd.HostEncoder(5, block_size=n) as enc:
            data, sizes = enc.encode_frames(pcm, n, 0)
            assert (sizes == ref_sizes).all(), chunk
            assert data.tobytes() == ref.tobytes(), chunk
            assert bytes(enc.streaminfo().md5sum) == ref_md5
    out, bs = decoder.decode(ref, 2, 16, pcm.shape[0])
    assert (out == pcm).all() and len(bs) == nblocks
