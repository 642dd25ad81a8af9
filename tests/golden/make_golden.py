#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/.

Run in the build container (needs /root/reference for oracle/_ref):

    python tests/golden/make_golden.py

Two families, kept in separate files and labelled in their `source` field:

* ref_*.npz     outputs of the REAL reference functions (lpc.c, rice.c,
                bitio.h, crc.c compiled where they lie into oracle/_ref) on
                committed inputs.  These pin both the oracle and the HIP path.
* ref_path.npz  whole-path outputs (prepare -> encode_residual -> frame bytes) of the
                REPLAY in tests/refreplay.py: every arithmetic step is the compiled
                reference (lpc.c, rice.c, bitio.h, crc.c) or a numpy integer
                one-liner; only ~40 lines of control flow of optimize.c / encode.c
                are re-stated.  Inputs are regenerated from committed seeds and
                checked against a stored SHA-1.  These pin the oracle (CPU suite) and
                the HIP path (-m gpu) to reference-compiled arithmetic end to end.
* path_*.npz    whole-path outputs (prepare -> encode_residual -> emit -> frame)
                produced by the oracle restatement, for the parts of libflake
                that cannot be built here (optimize.c / encode.c / vbs.c need
                the CMake-generated config.h).  Regression vectors, not
                reference outputs.

Only data is stored: inputs and expected outputs, never reference source.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import flake_amd                      # noqa: E402
import oraclelib                      # noqa: E402
from cases import edge_blocks, stereo_frames, param_sets, _rng   # noqa: E402
import goldenlib                      # noqa: E402
import refreplay                      # noqa: E402


def blocks_for_lpc():
    out = []
    pcm = flake_amd.synth_pcm(3, 1024, 2, 16)
    out += list(np.ascontiguousarray(pcm.transpose(0, 2, 1)).reshape(-1, 1024))
    pcm = flake_amd.synth_pcm(2, 1024, 2, 24, first_frame=5)
    out += list(np.ascontiguousarray(pcm.transpose(0, 2, 1)).reshape(-1, 1024))
    e = edge_blocks(1024, 16)
    out += [e[k] for k in ("impulse", "step", "alt_full_scale", "white", "sine", "decay", "wasted_3")]
    return np.stack(out).astype(np.int32)


def make_ref_lpc(ref):
    blocks = blocks_for_lpc()
    nb = blocks.shape[0]
    orders = np.array([1, 8, 12, 32], np.int32)
    autoc = np.zeros((nb, len(orders), 33), np.float64)
    coefs = np.zeros((nb, len(orders), 7, 32, 32), np.int32)
    shift = np.zeros((nb, len(orders), 7, 32), np.int32)
    opt = np.zeros((nb, len(orders), 7), np.int32)
    for b in range(nb):
        for oi, mo in enumerate(orders):
            autoc[b, oi] = ref.compute_autocorr(blocks[b], int(mo))
            for om in range(7):
                c, s, o = ref.lpc_calc_coefs(blocks[b], int(mo), 15, om)
                coefs[b, oi, om], shift[b, oi, om], opt[b, oi, om] = c, s, o
    np.savez_compressed(os.path.join(HERE, "ref_lpc.npz"), source="reference lpc.c via oracle/_ref",
                        blocks=blocks, orders=orders, autoc_bits=autoc.view(np.uint64),
                        coefs=coefs, shift=shift, opt_order=opt)


def make_ref_rice(ref):
    r = _rng(21)
    n_list = [1024, 576, 192]
    recs = []
    for n in n_list:
        residuals = [r.randint(-60, 61, n), r.randint(-20000, 20001, n),
                     (r.standard_cauchy(n) * 30).clip(-2 ** 29, 2 ** 29),
                     np.where(np.arange(n) < n // 2, r.randint(-3, 4, n), r.randint(-9000, 9001, n)),
                     r.randint(-2 ** 31, 2 ** 31 - 1, n)]
        for res in residuals:
            res = np.asarray(res).astype(np.int64).astype(np.int32)
            for lpc, order, pmin, pmax in ((1, 8, 0, 5), (1, 32, 0, 8), (0, 2, 0, 3), (1, 1, 2, 6), (0, 0, 0, 8)):
                bits, meth, por, par = ref.calc_rice_params(lpc, pmin, pmax, res, order, 17, 15)
                cap = 1 << 22
                rc, nbits, out = ref.emit_residual(meth, por, par, order, res, cap)
                recs.append(dict(res=res, lpc=lpc, order=order, pmin=pmin, pmax=pmax, bits=bits,
                                 method=meth, porder=por, params=par.copy(),
                                 emit_nbits=nbits, emit=out[:max(rc, 0)].copy(), emit_rc=rc))
    save = {"source": "reference rice.c + bitio.h via oracle/_ref", "count": len(recs)}
    for i, rec in enumerate(recs):
        for k, v in rec.items():
            save[f"{k}_{i}"] = v
    np.savez_compressed(os.path.join(HERE, "ref_rice.npz"), **save)

    sums = [0, 1, 2, 7, 8, 100, 2047, 2048, 2049, 65535, 1 << 20, (1 << 32) - 1, 1 << 32,
            (1 << 40) + 12345, (1 << 48) - 1, 1 << 63, (1 << 64) - 1]
    sums += [int(x) for x in r.randint(0, 1 << 31, 100)]
    ns = [0, 1, 2, 16, 17, 128, 4095, 4096, 65535]
    tab = np.zeros((len(ns), len(sums)), np.int32)
    for i, n in enumerate(ns):
        for j, s in enumerate(sums):
            tab[i, j] = ref.find_optimal_rice_param(s, n)
    np.savez_compressed(os.path.join(HERE, "ref_rice_k.npz"), source="reference rice.c:30-45",
                        sums=np.array(sums, np.uint64), ns=np.array(ns, np.int32), k=tab)

    data = r.randint(0, 256, 5000).astype(np.uint8)
    lens = np.array([0, 1, 2, 15, 16, 255, 4096, 5000], np.int32)
    np.savez_compressed(os.path.join(HERE, "ref_crc.npz"), source="reference crc.c", data=data, lens=lens,
                        crc8=np.array([ref.crc8(data[:l]) for l in lens], np.int32),
                        crc16=np.array([ref.crc16(data[:l]) for l in lens], np.int32))


def make_ref_path(ref):
    """ref_path.npz: the replay's outputs for every case goldenlib.ref_path_cases() lists."""
    save = {"source": "tests/refreplay.py: reference lpc.c/rice.c/bitio.h/crc.c via oracle/_ref + "
                      "re-stated control flow of optimize.c:124-276, encode.c:541-977"}
    kept = []
    for name, p, n, pcm, first in goldenlib.ref_path_cases():
        if (n & 1) and p.prediction_type == flake_amd.PRED_LEVINSON and n > p.max_prediction_order:
            continue            # lpc.c:35,53: uninitialised window centre for odd n
        nfr = pcm.shape[0]
        info = np.zeros(nfr * p.channels, flake_amd.INFO_DTYPE)
        sha = np.zeros((nfr * p.channels, 20), np.uint8)
        frames, ok = [], True
        fell = np.zeros(nfr, np.uint8)
        step = n if p.allow_vbs else 1
        for f in range(nfr):
            frame, subs, prep = refreplay.encode_frame(ref, p, first + f * step, pcm[f], n)
            if max(prep["obits"]) > 32:
                ok = False      # bitwriter_writebits(33, ..): undefined shift (bitio.h:103)
                break
            frames.append(frame)
            fell[f] = prep["fallback"]
            for c in range(p.channels):
                goldenlib.fill_info(info[f * p.channels + c], subs[c], prep, c, n, ref)
                sha[f * p.channels + c] = goldenlib.residual_digest(subs[c], n)
        if not ok:
            continue
        kept.append(name)
        save[f"params_{name}"] = np.array([getattr(p, k) for k, _ in p._fields_], np.int32)
        save[f"n_{name}"] = n
        save[f"first_{name}"] = first
        save[f"pcmsha_{name}"] = goldenlib.digest(pcm)
        save[f"info_{name}"] = info
        save[f"ressha_{name}"] = sha
        save[f"fallback_{name}"] = fell
        allf = np.concatenate(frames)
        if allf.size <= 24576:          # larger cases keep only the per-frame SHA-1
            save[f"frames_{name}"] = allf
        save[f"framesha_{name}"] = np.stack([goldenlib.digest_bytes(f) for f in frames])
        save[f"framelens_{name}"] = np.array([len(f) for f in frames], np.int32)
    save["names"] = np.array(kept)
    np.savez_compressed(os.path.join(HERE, "ref_path.npz"), **save)


def make_path(orc):
    """Whole-path regression vectors from the oracle (small batches)."""
    save = {"source": "oracle/flake_oracle.c (restatement; optimize.c/encode.c not buildable here)"}
    names = []
    for name, p, n in param_sets():
        nfr = 2 if p.channels <= 2 else 1
        if n * p.channels * nfr > 4000:
            n_use = min(n, 512) if name not in ("level12_bs8192",) else 1024
        else:
            n_use = n
        q = p.copy()
        q.block_size = max(n_use, 16)
        pcm = flake_amd.synth_pcm(nfr, n_use, p.channels, p.bits_per_sample, first_frame=3)
        slot = flake_amd.rice_slot_bytes(q, n_use)
        out = orc.encode_subframes_batch(q, pcm, n_use, slot_bytes=slot)
        frames = []
        for f in range(nfr):
            rc, fb, _, _, verb = orc.encode_frame(q, f, pcm[f], n_use)
            assert rc > 0
            frames.append(fb)
        names.append(name)
        save[f"params_{name}"] = np.array([getattr(q, k) for k, _ in q._fields_], np.int32)
        save[f"n_{name}"] = n_use
        save[f"pcm_{name}"] = pcm
        save[f"info_{name}"] = out["info"]
        save[f"residual_{name}"] = out["residual"]
        nb = out["info"]["rice_nbits"].clip(min=0)
        save[f"bits_{name}"] = np.concatenate(
            [out["rice_bits"][s, :(int(nb[s]) + 7) // 8] for s in range(len(nb))] or [np.zeros(0, np.uint8)])
        save[f"frames_{name}"] = np.concatenate(frames)
        save[f"framelens_{name}"] = np.array([len(f) for f in frames], np.int32)
    save["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "path_configs.npz"), **save)

    # stereo / edge frames at level-5 MAX
    p = flake_amd.level_params(5, order_method=flake_amd.OM_MAX, block_size=512)
    fr = stereo_frames(512, 16)
    keys = sorted(fr)
    pcm = np.stack([fr[k] for k in keys])
    slot = flake_amd.rice_slot_bytes(p, 512)
    out = orc.encode_subframes_batch(p, pcm, 512, slot_bytes=slot)
    np.savez_compressed(os.path.join(HERE, "path_stereo_edges.npz"), source=save["source"],
                        keys=np.array(keys), pcm=pcm, info=out["info"], residual=out["residual"],
                        params=np.array([getattr(p, k) for k, _ in p._fields_], np.int32))


if __name__ == "__main__":
    if not oraclelib.Ref.available():
        raise SystemExit("oracle/_ref is not built: /root/reference is needed to make golden vectors")
    make_ref_lpc(oraclelib.Ref())
    make_ref_rice(oraclelib.Ref())
    make_ref_path(oraclelib.Ref())
    make_path(oraclelib.Oracle())
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
