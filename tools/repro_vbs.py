"""Debug probe (tests' oracle as the checker): one seed of tests/test_gpu_vbs_dev.py::test_vbs_dev_random_streams,
frame by frame, then the differing piece through the subframe entry.  python tools/repro_vbs.py SEED"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, flake_amd, copy
import oraclelib
from cases import fuzz_params, fuzz_signal
from test_gpu_vbs_dev import run_dev
import parity

seed = int(sys.argv[1])
oracle = oraclelib.Oracle(); decoder = oraclelib.Decoder()
r = np.random.RandomState(9000 + seed)
n = int(r.choice([128, 256, 512, 1000, 1024, 1152, 2048, 2304, 4096, 4096, 4608, 8192]))
ch = int(r.choice([1, 2, 2, 2, 3, 8]))
bps = int(r.choice([8, 16, 16, 16, 20, 24, 24]))
p = fuzz_params(r, n, ch, bps)
p.variable_block_size = 1
p.allow_vbs = 1
if p.prediction_type != flake_amd.PRED_FIXED and n // 8 <= p.max_prediction_order:
    p.min_prediction_order = min(p.min_prediction_order, 4)
nblocks = 3 if n * ch > 16000 else int(r.randint(3, 9))
pcm = fuzz_signal(r, int(r.randint(0, 5)), nblocks, n, ch, bps).reshape(nblocks, n, ch).copy()
for b in range(nblocks):
    if r.rand() < 0.7:
        a, c = sorted(int(v) for v in r.randint(0, 9, 2))
        pcm[b, a * n // 8: c * n // 8] //= int(r.choice([8, 64, 1024]))
first = int(r.choice([0, 5 * n, 2 ** 21 - 8 * n]))
print(f"n={n} ch={ch} bps={bps} pred={p.prediction_type} om={p.order_method} order {p.min_prediction_order}..{p.max_prediction_order} porder {p.min_partition_order}..{p.max_partition_order} stereo {p.stereo_method}")
got = run_dev(p, pcm, n, first=first)
out, fc = [], first
for b in range(nblocks):
    rc, data, fc = oracle.encode_block(p, fc, pcm[b], n, 8 * n * ch * 4 + 4096)
    out.append(data)
exp = np.concatenate(out)
nfr, nbytes = int(got["totals"][0]), int(got["totals"][1])
print("bytes", nbytes, exp.size)
g = got["packed"][:nbytes]
_, bs_e = decoder.decode(exp, ch, bps, nblocks * n)
print("oracle pieces", [int(v) for v in bs_e])
fb = got["frame_bytes"][:nfr]
# frame sizes of the oracle stream: walk with the decoder's block sizes by re-encoding each piece
pos = 0; off_e = 0; off_g = 0
q = copy.copy(p); q.variable_block_size = 0; q.allow_vbs = 0
flat = pcm.reshape(-1, ch)
for i, m in enumerate(bs_e):
    m = int(m)
    piece = flat[pos:pos + m]
    q2 = copy.copy(q); q2.block_size = m
    same = None
    gsz = int(fb[i]) if i < nfr else -1
    print(f"piece {i}: n={m} dev bytes {gsz}")
    pos += m
    if i < nfr:
        ge = g[off_g:off_g + gsz]
        ee = exp[off_e:off_e + gsz]
        if ge.size != ee.size or (ge != ee).any():
            print("   first difference in this frame (or size differs)")
            # through the subframe entry
            oe = oracle.encode_subframes_batch(q2, piece.reshape(1, m, ch), m, want_residual=True, slot_bytes=flake_amd.rice_slot_bytes(q2, m))
            with flake_amd.Encoder(q2, max_frames=1) as enc:
                og = enc.encode_subframes(piece.reshape(1, m, ch), m)
            info_e, info_g = oe["info"], og["info"]
            for k in parity.SCALARS:
                print("   ", k, info_g[k], info_e[k])
            # every candidate order of the method on its own (MAX), both sides
            for o1 in sorted(set([11, 14, 16, 19, 21, 24, 26, 29, int(info_e["order"][0]), int(info_g["order"][0])])):
                q3 = copy.copy(q2); q3.order_method = flake_amd.OM_MAX; q3.max_prediction_order = o1; q3.min_prediction_order = 1
                oe3 = oracle.encode_subframes_batch(q3, piece.reshape(1, m, ch), m, want_residual=False)
                with flake_amd.Encoder(q3, max_frames=1) as enc:
                    og3 = enc.encode_subframes(piece.reshape(1, m, ch), m)
                print("    MAX order", o1, "est", og3["info"]["est_bits"], oe3["info"]["est_bits"], "porder", og3["info"]["porder"], oe3["info"]["porder"])
            for om in (2, 3, 4, 5, 6):
                q3 = copy.copy(q2); q3.order_method = om
                oe3 = oracle.encode_subframes_batch(q3, piece.reshape(1, m, ch), m, want_residual=False)
                with flake_amd.Encoder(q3, max_frames=1) as enc:
                    og3 = enc.encode_subframes(piece.reshape(1, m, ch), m)
                print("    method", om, "order", og3["info"]["order"], oe3["info"]["order"], "est", og3["info"]["est_bits"], oe3["info"]["est_bits"])
            for pm in (0, 6, 7):
                q3 = copy.copy(q2); q3.min_partition_order = pm
                oe3 = oracle.encode_subframes_batch(q3, piece.reshape(1, m, ch), m, want_residual=False)
                with flake_amd.Encoder(q3, max_frames=1) as enc:
                    og3 = enc.encode_subframes(piece.reshape(1, m, ch), m)
                print("    pmin", pm, "order", og3["info"]["order"], oe3["info"]["order"], "est", og3["info"]["est_bits"], oe3["info"]["est_bits"])
            break
        off_g += gsz; off_e += gsz
