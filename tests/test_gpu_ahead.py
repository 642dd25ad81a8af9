"""fhip_prepare_ahead: the feeder stage of the next batch started beside the batch in
flight.  Results must be identical with and without the hint, for hints that match,
hints that do not, batches that alternate between buffers, and a caller-owned info
buffer shared by consecutive batches (the race K0's own records exist to prevent)."""
import numpy as np
import pytest
import torch

import flake_amd
from parity import assert_bits_equal

pytestmark = pytest.mark.gpu


def _same_info(got, exp, what):
    for k in flake_amd.INFO_DTYPE.names:
        bad = np.nonzero((got[k] != exp[k]).reshape(got.size, -1).any(axis=1))[0]
        assert bad.size == 0, (what, k, int(bad[0]), got[k][bad[0]], exp[k][bad[0]])


def _run(enc, pcm_t, nframes, n, p, hint=None):
    dev = pcm_t.device
    nsub = nframes * p.channels
    slot = flake_amd.rice_slot_bytes(p, n)
    info = torch.zeros(nsub * flake_amd.INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    bits = torch.zeros(nsub * slot, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()      # the fills ran on torch's stream, the encoder has its own
    if hint is not None:
        enc.prepare_ahead(hint, nframes, n)
    enc.encode_subframes_dev(pcm_t, nframes, n, info, rice_bits=bits, slot_bytes=slot)
    enc.sync()
    return (np.frombuffer(info.cpu().numpy().tobytes(), flake_amd.INFO_DTYPE).copy(),
            bits.cpu().numpy().reshape(nsub, slot).copy())


@pytest.mark.parametrize("level,bps,ch", [(5, 16, 2), (5, 24, 2), (8, 16, 2), (2, 16, 2), (5, 16, 6)])
def test_hint_changes_nothing(level, bps, ch):
    p = flake_amd.level_params(level, channels=ch, bits_per_sample=bps)
    n, nfr = p.block_size, 70
    dev = torch.device("cuda", 0)
    a = torch.from_numpy(flake_amd.synth_pcm(nfr, n, ch, bps, first_frame=0)).to(dev)
    b = torch.from_numpy(flake_amd.synth_pcm(nfr, n, ch, bps, first_frame=1000)).to(dev)
    torch.cuda.synchronize()
    with flake_amd.Encoder(p, max_frames=nfr) as enc:
        ref_a = _run(enc, a, nfr, n, p)
        ref_b = _run(enc, b, nfr, n, p)
        assert not (ref_a[1] == ref_b[1]).all()
        for name, pcm_t, hint, ref in (("match a", a, a, ref_a), ("match b", b, b, ref_b),
                                       ("mismatch", a, b, ref_a), ("match a again", a, a, ref_a)):
            got = _run(enc, pcm_t, nfr, n, p, hint=hint)
            _same_info(got[0], ref[0], name)
            assert (got[1] == ref[1]).all(), name


def test_pipelined_stream_of_batches_with_one_shared_info_buffer():
    """The bench's loop: hint batch i+1, encode batch i, same info / bits buffers throughout."""
    p = flake_amd.level_params(5, order_method=flake_amd.OM_MAX)
    n, nfr, ch = p.block_size, 256, 2
    dev = torch.device("cuda", 0)
    pcms = [torch.from_numpy(flake_amd.synth_pcm(nfr, n, ch, 16, first_frame=f)).to(dev)
            for f in (0, 5000)]
    # frames of the second batch differ in stereo mode / row width from the first
    x = pcms[1].clone()
    x[::2, :, 1] = x[::2, :, 0]
    x[1::4] *= 2
    pcms[1] = x.contiguous()
    torch.cuda.synchronize()
    nsub = nfr * ch
    slot = flake_amd.rice_slot_bytes(p, n)
    with flake_amd.Encoder(p, max_frames=nfr) as enc:
        refs = [_run(enc, t, nfr, n, p) for t in pcms]
        st = torch.cuda.Stream(dev)                # one stream for the encoder and the snapshots
        enc.set_stream(st.cuda_stream)
        outs = []
        with torch.cuda.stream(st):
            info = torch.zeros(nsub * flake_amd.INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            bits = torch.zeros(nsub * slot, dtype=torch.uint8, device=dev)
            enc.prepare_ahead(pcms[0], nfr, n)
            for i in range(6):
                enc.encode_subframes_dev(pcms[i % 2], nfr, n, info, rice_bits=bits, slot_bytes=slot)
                enc.prepare_ahead(pcms[(i + 1) % 2], nfr, n)   # runs beside the kernels just queued
                outs.append((info.clone(), bits.clone()))     # stream-ordered snapshots
        enc.sync()
        torch.cuda.synchronize()
        for i, (inf, bt) in enumerate(outs):
            got = np.frombuffer(inf.cpu().numpy().tobytes(), flake_amd.INFO_DTYPE)
            _same_info(got, refs[i % 2][0], f"step {i}")
            # the shared buffer keeps the previous batch's bytes behind each section's end
            assert_bits_equal(bt.cpu().numpy().reshape(nsub, slot), refs[i % 2][1], got, f"step {i}")
