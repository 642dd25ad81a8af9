"""Loaders for the committed golden vectors (tests/golden/*.npz)."""
import hashlib
import os

import numpy as np

import flake_amd

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def params_from_array(a) -> flake_amd.Params:
    p = flake_amd.Params()
    for (k, _), v in zip(p._fields_, a):
        setattr(p, k, int(v))
    return p


def rice_records(z):
    for i in range(int(z["count"])):
        yield {k: z[f"{k}_{i}"] for k in ("res", "lpc", "order", "pmin", "pmax", "bits", "method",
                                          "porder", "params", "emit_nbits", "emit", "emit_rc")}


def split_bits(info, flat):
    """Undo the concatenation of per-subframe residual sections."""
    out, pos = [], 0
    for s in range(info.size):
        nb = max(int(info["rice_nbits"][s]), 0)
        ln = (nb + 7) // 8
        out.append(flat[pos:pos + ln])
        pos += ln
    return out


# ---- ref_path.npz: the replay's vectors (tests/refreplay.py) --------------------------------
def digest(a) -> np.ndarray:
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(a, np.int32).tobytes()).digest(), np.uint8)


def digest_bytes(a) -> np.ndarray:
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(a, np.uint8).tobytes()).digest(), np.uint8)


def residual_digest(sub, n) -> np.ndarray:
    """SHA-1 of the residual a subframe carries: residual[0] for CONSTANT (optimize.c:147), all n else."""
    res = np.ascontiguousarray(sub["residual"], np.int32)
    return digest(res[:1] if int(sub["type"]) == 0 else res[:n])


def ref_path_cases():
    """(name, params, n, pcm[nfr][n][ch], first_frame_number): the inputs of ref_path.npz,
    regenerated from seeds (their SHA-1 is stored beside the expected outputs)."""
    from cases import fuzz_case, param_sets, stereo_frames
    out = []
    for name, p, n in param_sets():
        nfr = 1 if (p.order_method == flake_amd.OM_SEARCH or p.channels > 2) else 2
        pcm = flake_amd.synth_pcm(nfr, n, p.channels, p.bits_per_sample, first_frame=3)
        out.append((name, p, n, pcm, 126))
    for bps in (16, 24):
        fr = stereo_frames(4096, bps)
        keys = sorted(fr)
        out.append((f"stereo_edges_{bps}", flake_amd.level_params(5, bits_per_sample=bps), 4096,
                    np.stack([fr[k] for k in keys]), 2 ** 21 - 3))
    r = np.random.RandomState(12)
    noise = r.randint(-32768, 32768, (2, 4096, 2)).astype(np.int32)
    out.append(("white_noise_verbatim_fallback", flake_amd.level_params(5), 4096, noise, 65535))
    # the order-search kernel's instances beyond n = 4096 (512 and 1024 tiles) and its VALU one
    for name, kw, n in (("search16_n16384", dict(order_method=flake_amd.OM_SEARCH, max_prediction_order=16,
                                                   max_partition_order=8), 16384),
                        ("level8_log32_n8192", dict(order_method=flake_amd.OM_LOG, max_prediction_order=32,
                                                    max_partition_order=8), 8192),
                        ("four_level_n2048_24bit", dict(order_method=flake_amd.OM_4LEVEL, max_prediction_order=12,
                                                        bits_per_sample=24), 2048)):
        p = flake_amd.level_params(5, block_size=n, **kw)
        out.append((name, p, n, flake_amd.synth_pcm(1, n, 2, p.bits_per_sample, first_frame=21), 5))
    for seed in range(128):
        p, pcm, n, what = fuzz_case(seed)
        out.append((f"fuzz{seed}", p, n, pcm[:2], [0, 120, 127, 2047, 65530][seed % 5]))
    return out


def fill_info(rec, sub, prep, c, n, ref):
    """One fhip_subframe_info record from the replay's subframe dict."""
    t = int(sub["type"])
    rec["type"], rec["type_code"] = t, int(sub["type_code"])
    rec["obits"], rec["wasted"], rec["ch_mode"] = prep["obits"][c], prep["wasted"][c], prep["ch_mode"]
    rec["est_bits"] = int(sub["est_bits"]) & 0xFFFFFFFF
    if t == 0:
        rec["warmup"][0] = sub["residual"][0]
    if t in (8, 32):
        o = int(sub["order"])
        rec["order"] = o
        rec["warmup"][:o] = sub["residual"][:o]
        rec["rice_method"], rec["porder"] = sub["method"], sub["porder"]
        rec["rparams"][:1 << sub["porder"]] = sub["rparams"][:1 << sub["porder"]]
        rc, nbits, _ = ref.emit_residual(sub["method"], sub["porder"], sub["rparams"], o,
                                         sub["residual"], 1 << 22)
        rec["rice_nbits"] = nbits
    if t == 32:
        rec["shift"] = sub["shift"]
        rec["coefs"][:sub["order"]] = sub["coefs"][:sub["order"]]


REF_PATH_SCALARS = ("type", "type_code", "obits", "wasted", "ch_mode", "est_bits")


def assert_ref_path_info(got, exp, what, nbits=None, slot_bytes=None):
    """fhip_subframe_info from the oracle or the HIP path vs the replay's record: the fields
    encode_residual() and the feeders define for the subframe's type."""
    for k in REF_PATH_SCALARS:
        assert int(got[k]) == int(exp[k]), (what, k, int(got[k]), int(exp[k]))
    t = int(exp["type"])
    if t == 0:
        assert int(got["warmup"][0]) == int(exp["warmup"][0]), (what, "constant value")
    if t in (8, 32):
        o = int(exp["order"])
        for k in ("order", "rice_method", "porder"):
            assert int(got[k]) == int(exp[k]), (what, k, int(got[k]), int(exp[k]))
        nb = int(got["rice_nbits"]) if nbits is None else int(nbits)
        want = int(exp["rice_nbits"])
        if slot_bytes is not None and want > 8 * slot_bytes:
            want = -1           # include/flakehip.h: a section that does not fit its slot (the frame
            #                     then falls back to verbatim, encode.c:949) is reported as -1
        assert nb == want, (what, "rice_nbits", nb, want)
        assert (got["warmup"][:o] == exp["warmup"][:o]).all(), (what, "warmup")
        npart = 1 << int(exp["porder"])
        assert (got["rparams"][:npart] == exp["rparams"][:npart]).all(), (what, "rparams")
    if t == 32:
        assert int(got["shift"]) == int(exp["shift"]), (what, "shift")
        assert (got["coefs"][:o] == exp["coefs"][:o]).all(), (what, "coefs")


def ref_path_loaded():
    """Yields (name, params, n, pcm, first, z) for every case stored in ref_path.npz, with the
    regenerated input verified against its stored SHA-1."""
    z = load("ref_path.npz")
    stored = set(str(s) for s in z["names"])
    for name, p, n, pcm, first in ref_path_cases():
        if name not in stored:
            continue
        assert (digest(pcm) == z[f"pcmsha_{name}"]).all(), f"{name}: regenerated input differs from the one the vectors were made from"
        assert (np.array([getattr(p, k) for k, _ in p._fields_], np.int32) == z[f"params_{name}"]).all(), name
        assert n == int(z[f"n_{name}"]) and first == int(z[f"first_{name}"]), name
        yield name, p, n, pcm, first, z


def assert_ref_path_frames(name, z, frames):
    """frames: list of uint8 arrays, one per frame, vs the stored bytes / SHA-1."""
    lens = z[f"framelens_{name}"]
    assert [len(f) for f in frames] == [int(x) for x in lens], (name, "frame sizes")
    if f"frames_{name}" in z.files:
        exp = z[f"frames_{name}"]
        got = np.concatenate(frames)
        bad = np.nonzero(got != exp)[0]
        assert bad.size == 0, (name, "first differing byte", int(bad[0]) if bad.size else -1)
    for f, fr in enumerate(frames):
        assert (digest_bytes(fr) == z[f"framesha_{name}"][f]).all(), (name, "frame digest", f)
