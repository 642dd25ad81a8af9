"""The HIP path against tests/golden/ref_path.npz: whole-path vectors whose every
arithmetic step is the COMPILED reference (lpc.c, rice.c, bitio.h, crc.c) and whose
control flow is the replay of optimize.c:124-276 / encode.c:541-977 in
tests/refreplay.py (made here by tests/golden/make_golden.py; /root/reference is not
needed to run this).  Subframe decisions, coefficients, Rice parameters, residuals
(by SHA-1) and the frames K4 assembles, byte for byte."""
import numpy as np
import pytest

import flake_amd
import goldenlib as G

pytestmark = pytest.mark.gpu


def test_hip_path_matches_reference_replay():
    count = 0
    for name, p, n, pcm, first, z in G.ref_path_loaded():
        nfr = pcm.shape[0]
        with flake_amd.Encoder(p, max_frames=nfr) as enc:
            got = enc.encode_subframes(pcm, n, want_residual=True, want_frames=True,
                                       first_frame_number=first)
        info, sha = z[f"info_{name}"], z[f"ressha_{name}"]
        res = got["residual"].reshape(-1, n)
        for s in range(nfr * p.channels):
            G.assert_ref_path_info(got["info"][s], info[s], f"{name} sub{s}",
                                   slot_bytes=got["slot_bytes"])
            d = G.residual_digest({"type": info[s]["type"], "residual": res[s]}, n)
            assert (d == sha[s]).all(), (name, s, "residual")
        frames = [got["frames"][f, :int(got["frame_bytes"][f])] for f in range(nfr)]
        G.assert_ref_path_frames(name, z, frames)
        count += 1
    assert count >= 130
