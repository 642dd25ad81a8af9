"""fhip_encode_blocks_vbs_dev: a variable-block-size batch (encode_frame_vbs, vbs.c:85-119, around
encode_frame; BASELINE configs[4]) with blocks and stream device-resident and no host
synchronisation inside -- split_frame_v1, the piece tables (eight bins of equal piece length,
counted and scanned on the device), the path per bin, the frames packed in stream order.  The
packed stream must equal the oracle's flake_encode_frame() loop byte for byte and decode back to
the input; the per-frame / per-block tables must agree with it."""
import numpy as np
import pytest
import torch

import flake_amd
from test_host_frames import oracle_stream

pytestmark = pytest.mark.gpu


def split_blocks(nblocks, n, ch, bps, every=2):
    """Blocks with a quiet first part of 1 .. 7 eighths (every `every`-th block): the splitter cuts
    there, so every piece length k * n / 8 occurs."""
    pcm = flake_amd.synth_pcm(nblocks, n, ch, bps)
    for b in range(0, nblocks, every):
        cut = (1 + (b // every) % 7) * n // 8
        pcm[b, :cut] //= 64
    return pcm


def run_dev(p, pcm, n, cap=None, first=0):
    nblocks = pcm.shape[0]
    dev = torch.device("cuda", 0)
    d_pcm = torch.from_numpy(np.ascontiguousarray(pcm, dtype=np.int32)).to(dev)
    cap = pcm.size * 5 + 4096 if cap is None else cap
    guard = 64
    packed = torch.full((cap + guard,), 0xA5, dtype=torch.uint8, device=dev)
    totals = torch.zeros(4, dtype=torch.int64, device=dev)
    fbytes = torch.zeros(8 * nblocks, dtype=torch.int32, device=dev)
    bbytes = torch.zeros(nblocks, dtype=torch.int32, device=dev)
    bframes = torch.zeros(nblocks, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    with flake_amd.Encoder(p, max_frames=8 * nblocks) as enc:
        enc.encode_blocks_vbs_dev(d_pcm, nblocks, n, packed, cap, totals, frame_bytes=fbytes,
                                  block_bytes=bbytes, block_frames=bframes, first_frame_number=first)
        enc.sync()
    t = totals.cpu().numpy()
    return dict(totals=t, packed=packed.cpu().numpy(), frame_bytes=fbytes.cpu().numpy(),
                block_bytes=bbytes.cpu().numpy(), block_frames=bframes.cpu().numpy(), cap=cap, guard=guard)


def check_against_oracle(oracle, decoder, p, pcm, n, what):
    ch, bps = p.channels, p.bits_per_sample
    got = run_dev(p, pcm, n)
    flat = pcm.reshape(-1, ch)
    exp, esizes = oracle_stream(oracle, p, flat, n)
    nfr, nbytes, mx, cut = (int(v) for v in got["totals"])
    assert cut == 0, what
    assert nbytes == exp.size, (what, nbytes, exp.size)
    assert got["packed"][:nbytes].tobytes() == exp.tobytes(), what
    assert (got["packed"][got["cap"]:] == 0xA5).all(), what                # nothing past the buffer
    assert (got["block_bytes"] == esizes).all(), what
    out, bs = decoder.decode(got["packed"][:nbytes], ch, bps, flat.shape[0])
    assert (out == flat).all(), what
    assert len(bs) == nfr and got["block_frames"].sum() == nfr, what
    fb = got["frame_bytes"][:nfr]
    assert fb.sum() == nbytes and fb.max() == mx and (got["frame_bytes"][nfr:] == 0).all(), what
    # frames per block against the oracle's splitter
    for b in range(pcm.shape[0]):
        enf, _ = oracle.vbs_split(pcm[b], ch, n)
        assert got["block_frames"][b] == max(enf, 1), (what, b)
    return bs


@pytest.mark.parametrize("level", [9, 10, 11, 12])
def test_vbs_dev_levels(oracle, decoder, level):
    p = flake_amd.level_params(level)
    n = p.block_size
    pcm = split_blocks(8, n, 2, 16, every=1)
    pcm[7] = flake_amd.synth_pcm(1, n, 2, 16, first_frame=77)[0]          # one block left whole
    bs = check_against_oracle(oracle, decoder, p, pcm, n, f"level {level}")
    assert len(set(int(v) * 8 // n for v in bs)) >= 5                       # most bins were used


def test_vbs_dev_levels_a_launch_per_bin_subprocess():
    """Batches of up to 2048 blocks send their thinly filled bins (four eighths and longer) through ONE order-search
    launch and ONE K3 launch (k_order_search_bins / k_encode_bins); larger batches keep a launch per bin.
    FHIP_VBS_MERGE_MAX=0 takes the tests' small batches the second way: the stream must be the oracle's either way
    (test_vbs_dev_levels above runs the first)."""
    import subprocess, sys, os, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import flake_amd, test_gpu_vbs_dev as t
        from oraclelib import Oracle, Decoder
        o, d = Oracle(), Decoder()
        for level in (9, 10, 11, 12):
            p = flake_amd.level_params(level)
            n = p.block_size
            pcm = t.split_blocks(8, n, 2, 16, every=1)
            pcm[7] = flake_amd.synth_pcm(1, n, 2, 16, first_frame=77)[0]
            t.check_against_oracle(o, d, p, pcm, n, "level %%d, a launch per bin" %% level)
        print("per bin ok")
    """ % (root, os.path.join(root, "tests")))
    env = dict(os.environ, FHIP_VBS_MERGE_MAX="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "per bin ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("ch,bps,n", [(1, 16, 4096), (3, 24, 2048), (2, 24, 4096), (8, 16, 1024),
                                      (2, 16, 1000), (2, 8, 128)])
def test_vbs_dev_shapes(oracle, decoder, ch, bps, n):
    """Other channel counts, sample widths and block sizes (an eighth that is no multiple of four:
    the generic K0 / K3; the smallest block the splitter accepts, vbs.c:93)."""
    p = flake_amd.level_params(9, channels=ch, bits_per_sample=bps, block_size=n)
    pcm = split_blocks(9, n, ch, bps, every=1)
    check_against_oracle(oracle, decoder, p, pcm, n, f"ch {ch} bps {bps} n {n}")


def test_vbs_dev_frame_numbers_and_reuse(oracle, decoder):
    """Frame numbers count samples from first_frame_number (encode.c:969-975); a handle serves
    batch after batch (the bins' stale slots from the batch before are never read)."""
    p = flake_amd.level_params(10)
    n = p.block_size
    a = split_blocks(5, n, 2, 16, every=1)
    b = split_blocks(3, n, 2, 16, every=2)
    dev = torch.device("cuda", 0)
    with flake_amd.Encoder(p, max_frames=8 * 5) as enc:
        outs = []
        first = 0
        for pcm in (a, b, a):
            nb = pcm.shape[0]
            d_pcm = torch.from_numpy(pcm).to(dev)
            cap = pcm.size * 5
            packed = torch.zeros(cap, dtype=torch.uint8, device=dev)
            totals = torch.zeros(4, dtype=torch.int64, device=dev)
            torch.cuda.synchronize(dev)
            enc.encode_blocks_vbs_dev(d_pcm, nb, n, packed, cap, totals, first_frame_number=first)
            enc.sync()
            outs.append(packed.cpu().numpy()[:int(totals[1].item())])
            first += nb * n
    stream = np.concatenate(outs)
    allpcm = np.concatenate([a, b, a]).reshape(-1, 2)
    exp, _ = oracle_stream(oracle, p, allpcm, n)
    assert stream.tobytes() == exp.tobytes()
    out, _ = decoder.decode(stream, 2, 16, allpcm.shape[0])
    assert (out == allpcm).all()


def test_vbs_dev_refuses_misaligned_pcm():
    """The device entry reads the blocks with 16-byte loads: a view that starts on an odd stereo
    sample-frame (8-byte aligned only) is refused, not read (include/flakehip.h)."""
    p = flake_amd.level_params(9)
    n = p.block_size
    dev = torch.device("cuda", 0)
    buf = torch.zeros(2 * n * 2 + 2, dtype=torch.int32, device=dev)
    view = buf[2:]                                   # 8 bytes past a 16-byte boundary
    assert view.data_ptr() % 16 == 8
    packed = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
    totals = torch.zeros(4, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    with flake_amd.Encoder(p, max_frames=16) as enc:
        with pytest.raises(flake_amd.FlakeHipError) as ei:
            enc.encode_blocks_vbs_dev(view, 2, n, packed, packed.numel(), totals)
        assert ei.value.code == flake_amd.E_INVALID
        enc.encode_blocks_vbs_dev(buf, 2, n, packed, packed.numel(), totals)      # the aligned buffer is fine
        enc.sync()
    assert int(totals.cpu()[3]) == 0


def test_vbs_dev_buffer_too_small():
    """A stream that does not fit packed_cap: flagged in totals[3], frames past the end are not
    written, nothing is written behind the buffer."""
    p = flake_amd.level_params(9)
    n = p.block_size
    pcm = split_blocks(6, n, 2, 16, every=1)
    full = run_dev(p, pcm, n)
    nbytes = int(full["totals"][1])
    cut = run_dev(p, pcm, n, cap=nbytes // 2)
    assert int(cut["totals"][3]) == 1 and int(cut["totals"][1]) == nbytes
    assert (cut["packed"][cut["cap"]:] == 0xA5).all()
    # whole frames that end inside the buffer are there
    ends = np.cumsum(full["frame_bytes"][:int(full["totals"][0])])
    last = ends[ends <= nbytes // 2].max()
    assert cut["packed"][:last].tobytes() == full["packed"][:last].tobytes()


@pytest.mark.parametrize("level,nblocks", [(10, 1024), (12, 1024)])
def test_vbs_dev_corpus_properties(oracle, decoder, level, nblocks):
    """BASELINE configs[4] at bench size through the device entry: the stream decodes to the input,
    frames are multiples of n/8, the tables are consistent, and the stream's first blocks equal the
    oracle's stream of those blocks."""
    p = flake_amd.level_params(level)
    n = p.block_size
    pcm = split_blocks(nblocks, n, 2, 16, every=3)
    got = run_dev(p, pcm, n)
    nfr, nbytes, mx, cut = (int(v) for v in got["totals"])
    assert cut == 0 and nfr > nblocks + nblocks // 6
    flat = pcm.reshape(-1, 2)
    out, bs = decoder.decode(got["packed"][:nbytes], 2, 16, flat.shape[0])
    assert (out == flat).all() and len(bs) == nfr and (bs % (n // 8) == 0).all()
    assert got["block_bytes"].sum() == nbytes and got["block_frames"].sum() == nfr
    npre = 6
    exp, esizes = oracle_stream(oracle, p, flat[:npre * n], n)
    assert (got["block_bytes"][:npre] == esizes).all()
    assert got["packed"][:exp.size].tobytes() == exp.tobytes()


_VBS_SEEDS = int(__import__("os").environ.get("FLAKE_FUZZ_VBS_SEEDS", "24"))
_VBS_FIRST = int(__import__("os").environ.get("FLAKE_FUZZ_FIRST", "0"))


@pytest.mark.parametrize("seed", range(_VBS_FIRST, _VBS_FIRST + _VBS_SEEDS))
def test_vbs_dev_random_streams(oracle, decoder, seed):
    """Seeded sweep through the device-resident VBS entry: block sizes (multiples of 8 from 128),
    channel counts, sample widths, prediction types, every order method, ragged order / partition
    ranges, signal kinds with quiet stretches -- the packed stream against the oracle's
    flake_encode_frame() loop, byte for byte, and decoded back."""
    from cases import fuzz_params, fuzz_signal
    r = np.random.RandomState(9000 + seed)
    n = int(r.choice([128, 256, 512, 1000, 1024, 1152, 2048, 2304, 4096, 4096, 4608, 8192]))
    ch = int(r.choice([1, 2, 2, 2, 3, 8]))
    bps = int(r.choice([8, 16, 16, 16, 20, 24, 24]))
    p = fuzz_params(r, n, ch, bps)
    p.variable_block_size = 1
    p.allow_vbs = 1
    if p.prediction_type != flake_amd.PRED_FIXED and n // 8 <= p.max_prediction_order:
        # a piece no longer than the maximum order takes the reference's FIXED branch with this minimum
        # order; above 4 that is undefined behaviour there (optimize.c:168-173)
        p.min_prediction_order = min(p.min_prediction_order, 4)
    nblocks = 3 if n * ch > 16000 else int(r.randint(3, 9))
    pcm = fuzz_signal(r, int(r.randint(0, 5)), nblocks, n, ch, bps).reshape(nblocks, n, ch).copy()
    for b in range(nblocks):                          # quiet stretches of whole eighths: the splitter cuts
        if r.rand() < 0.7:
            a, c = sorted(int(v) for v in r.randint(0, 9, 2))
            pcm[b, a * n // 8: c * n // 8] //= int(r.choice([8, 64, 1024]))
    first = int(r.choice([0, 5 * n, 2 ** 21 - 8 * n]))
    what = f"vbs seed {seed}: n={n} ch={ch} bps={bps} pred={p.prediction_type} om={p.order_method} " \
           f"order {p.min_prediction_order}..{p.max_prediction_order} porder {p.min_partition_order}..{p.max_partition_order}"
    got = run_dev(p, pcm, n, first=first)
    flat = pcm.reshape(-1, ch)
    out, esizes, fc = [], [], first
    for b in range(nblocks):
        rc, data, fc = oracle.encode_block(p, fc, pcm[b], n, 8 * n * ch * 4 + 4096)
        assert rc > 0, what
        out.append(data)
        esizes.append(rc)
    exp = np.concatenate(out)
    nfr, nbytes, mx, cut = (int(v) for v in got["totals"])
    assert cut == 0 and nbytes == exp.size, (what, nbytes, exp.size)
    bad = np.nonzero(got["packed"][:nbytes] != exp)[0]
    assert bad.size == 0, (what, "first differing byte", int(bad[0]) if bad.size else -1)
    assert (got["block_bytes"] == np.array(esizes)).all(), what
    if first == 0:
        dec, _ = decoder.decode(got["packed"][:nbytes], ch, bps, flat.shape[0])
        assert (dec == flat).all(), what
