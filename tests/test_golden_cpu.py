"""The oracle against the committed golden vectors (no GPU, no /root/reference).

ref_*.npz hold outputs of the real reference functions; ref_path.npz the whole-path
outputs of the replay built on them (tests/refreplay.py); path_*.npz hold
whole-path regression vectors made by the oracle itself (see make_golden.py).
"""
import numpy as np

import goldenlib as G


def test_ref_lpc(oracle):
    z = G.load("ref_lpc.npz")
    blocks, orders = z["blocks"], z["orders"]
    for b in range(blocks.shape[0]):
        for oi, mo in enumerate(orders):
            mo = int(mo)
            got = oracle.window_autocorr(blocks[b], mo)[:mo + 1]
            assert (got.view(np.uint64) == z["autoc_bits"][b, oi, :mo + 1]).all(), (b, mo)
            for om in range(7):
                with np.errstate(all="ignore"):
                    c, s, o = oracle.lpc_calc_coefs(blocks[b], mo, 15, om)
                assert o == z["opt_order"][b, oi, om]
                assert (c == z["coefs"][b, oi, om]).all() and (s == z["shift"][b, oi, om]).all()


def test_ref_rice_and_emit(oracle):
    z = G.load("ref_rice.npz")
    for rec in G.rice_records(z):
        res, order = rec["res"], int(rec["order"])
        bits, sf = oracle.subframe_bits(res, int(rec["pmin"]), int(rec["pmax"]), order, 17, 15,
                                        int(rec["lpc"]))
        assert bits == int(rec["bits"])
        assert sf["rice_method"] == rec["method"] and sf["porder"] == rec["porder"]
        np_ = 1 << int(rec["porder"])
        assert (sf["rparams"][:np_] == rec["params"][:np_]).all()
        sf = sf.copy()
        sf["order"] = order
        nb, out = oracle.emit_residual(sf, res, 1 << 22)
        assert nb == int(rec["emit_nbits"])
        assert (out[:(nb + 7) // 8] == rec["emit"]).all()


def test_ref_rice_k(oracle):
    z = G.load("ref_rice_k.npz")
    for i, n in enumerate(z["ns"]):
        for j, s in enumerate(z["sums"]):
            assert oracle.rice_best_k(int(s), int(n)) == z["k"][i, j]


def test_ref_crc(oracle):
    z = G.load("ref_crc.npz")
    for i, ln in enumerate(z["lens"]):
        assert oracle.crc8(z["data"][:ln]) == z["crc8"][i]
        assert oracle.crc16(z["data"][:ln]) == z["crc16"][i]


def test_path_configs(oracle):
    z = G.load("path_configs.npz")
    for name in z["names"]:
        p = G.params_from_array(z[f"params_{name}"])
        n = int(z[f"n_{name}"])
        pcm = z[f"pcm_{name}"]
        import flake_amd
        slot = flake_amd.rice_slot_bytes(p, n)
        out = oracle.encode_subframes_batch(p, pcm, n, slot_bytes=slot)
        assert out["info"].tobytes() == z[f"info_{name}"].tobytes(), name
        assert (out["residual"] == z[f"residual_{name}"]).all(), name
        pos = 0
        for f in range(pcm.shape[0]):
            rc, fb, _, _, _ = oracle.encode_frame(p, f, pcm[f], n)
            ln = int(z[f"framelens_{name}"][f])
            assert rc == ln and (fb == z[f"frames_{name}"][pos:pos + ln]).all(), (name, f)
            pos += ln


def test_path_stereo_edges(oracle):
    z = G.load("path_stereo_edges.npz")
    p = G.params_from_array(z["params"])
    out = oracle.encode_subframes_batch(p, z["pcm"], 512, slot_bytes=0)
    assert out["info"].tobytes() == z["info"].tobytes()
    assert (out["residual"] == z["residual"]).all()
    # every channel mode and the constant side channel occur in these frames
    modes = set(int(m) for m in z["info"]["ch_mode"])
    assert {1, 8, 9, 10} <= modes, modes
    assert (z["info"]["type"] == 0).any()


def test_ref_path(oracle):
    """The oracle's prepare -> encode_residual -> frame bytes against the replay's vectors
    (reference-compiled arithmetic + re-stated control flow), 141 cases."""
    count = 0
    for name, p, n, pcm, first, z in G.ref_path_loaded():
        info, sha = z[f"info_{name}"], z[f"ressha_{name}"]
        step = n if p.allow_vbs else 1
        frames = []
        for f in range(pcm.shape[0]):
            rc, fb, sfs, res, verb = oracle.encode_frame(p, first + f * step, pcm[f], n)
            assert rc > 0, name
            frames.append(fb)
            assert bool(verb) == bool(z[f"fallback_{name}"][f]), (name, f, "verbatim fallback")
            if verb:
                continue                         # the oracle reports the re-encoded subframes
            for c in range(p.channels):
                s = f * p.channels + c
                nb = oracle.residual_section_bits(sfs[c], res[c]) if sfs[c]["type"] in (8, 32) else 0
                G.assert_ref_path_info(sfs[c], info[s], f"{name} f{f} ch{c}", nbits=nb)
                got = G.residual_digest({"type": sfs[c]["type"], "residual": res[c]}, n)
                assert (got == sha[s]).all(), (name, f, c, "residual")
        G.assert_ref_path_frames(name, z, frames)
        count += 1
    assert count >= 130
