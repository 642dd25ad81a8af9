"""ctypes views of the test oracles.  TEST INFRASTRUCTURE ONLY.

* ``oracle/libflake_oracle.so`` -- our CPU restatement (oracle/flake_oracle.c)
* ``oracle/_ref/libflake_ref.so`` -- the real reference lpc.c / rice.c / crc.c /
  bitio.h behind oracle/ref_harness.c; exists only where /root/reference was
  available at build time (it travels to the GPU box as a built file).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libflake_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libflake_ref.so")

MAX_ORDER, MAX_PARTS, MAX_LAGS = 32, 256, 33

INFO_DTYPE = np.dtype([
    ("type", "<i4"), ("type_code", "<i4"), ("order", "<i4"), ("shift", "<i4"),
    ("obits", "<i4"), ("wasted", "<i4"), ("rice_method", "<i4"), ("porder", "<i4"),
    ("est_bits", "<u4"), ("ch_mode", "<i4"), ("rice_nbits", "<i4"), ("reserved", "<i4"),
    ("coefs", "<i4", (MAX_ORDER,)), ("rparams", "<i4", (MAX_PARTS,)),
    ("warmup", "<i4", (MAX_ORDER,)),
])


class FoParams(C.Structure):
    _fields_ = [(k, C.c_int) for k in (
        "channels", "sample_rate", "bits_per_sample", "block_size", "order_method",
        "stereo_method", "prediction_type", "min_prediction_order", "max_prediction_order",
        "min_partition_order", "max_partition_order", "variable_block_size", "allow_vbs",
        "lpc_precision")]


def to_fo_params(p) -> FoParams:
    """Accepts flake_amd.Params (same field names) or a dict."""
    q = FoParams()
    for k, _ in FoParams._fields_:
        setattr(q, k, p[k] if isinstance(p, dict) else getattr(p, k))
    return q


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False) -> None:
    """(Re)build the oracle libraries with oracle/Makefile when sources are newer."""
    src = [os.path.join(ORACLE_DIR, f) for f in ("flake_oracle.c", "flake_oracle.h", "Makefile")]
    if force or _stale(ORACLE_SO, src):
        subprocess.run(["make", "-C", ORACLE_DIR, "libflake_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    have_ref_src = os.path.isdir("/root/reference/libflake")
    if have_ref_src and (force or _stale(REF_SO, [os.path.join(ORACLE_DIR, "ref_harness.c"),
                                                   os.path.join(ORACLE_DIR, "Makefile")])):
        subprocess.run(["make", "-C", ORACLE_DIR, "ref"], check=True, stdout=subprocess.DEVNULL)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self):
        build()
        L = self.L = C.CDLL(ORACLE_SO)
        vp, i, i64, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64
        L.fo_set_defaults.argtypes = [C.POINTER(FoParams), i]
        L.fo_window_autocorr.argtypes = [vp, i, i, vp]
        L.fo_levinson.argtypes = [vp, i, vp, vp]
        L.fo_schur_order_est.argtypes = [vp, i, vp]
        L.fo_schur_order_est.restype = i
        L.fo_quantize_coefs.argtypes = [vp, i, i, vp, vp]
        L.fo_lpc_calc_coefs.argtypes = [vp, i, i, i, i, vp, vp]
        L.fo_lpc_calc_coefs.restype = i
        L.fo_residual_fixed.argtypes = [vp, vp, i, i]
        L.fo_residual_lpc.argtypes = [vp, vp, i, i, vp, i]
        L.fo_encode_residual.argtypes = [C.POINTER(FoParams), vp, vp, vp, i]
        L.fo_encode_residual.restype = i
        L.fo_rice_best_k.argtypes = [u64, i]
        L.fo_rice_best_k.restype = i
        L.fo_subframe_bits.argtypes = [vp, i, i, vp, i, i, i, i, i]
        L.fo_subframe_bits.restype = C.c_uint32
        L.fo_stereo_mode.argtypes = [vp, vp, i]
        L.fo_stereo_mode.restype = i
        L.fo_prepare_frame.argtypes = [C.POINTER(FoParams), vp, i, vp, vp]
        L.fo_prepare_frame.restype = i
        L.fo_residual_section_bits.argtypes = [vp, vp, i]
        L.fo_residual_section_bits.restype = i64
        L.fo_emit_residual.argtypes = [vp, vp, i, vp, i64]
        L.fo_emit_residual.restype = i64
        L.fo_crc8.argtypes = [vp, C.c_uint32]
        L.fo_crc8.restype = C.c_uint8
        L.fo_crc16.argtypes = [vp, C.c_uint32]
        L.fo_crc16.restype = C.c_uint16
        L.fo_encode_frame.argtypes = [C.POINTER(FoParams), C.c_uint32, vp, i, vp, i, vp, vp, C.POINTER(i)]
        L.fo_encode_frame.restype = i
        L.fo_vbs_split.argtypes = [vp, i, i, C.POINTER(i), vp]
        L.fo_encode_block.argtypes = [C.POINTER(FoParams), C.POINTER(C.c_uint32), vp, i, vp, i]
        L.fo_encode_block.restype = i
        L.fo_encode_subframes_batch.argtypes = [C.POINTER(FoParams), vp, i, i, vp, vp, vp, i64]
        L.fo_encode_subframes_batch.restype = i

    # ---- lpc ----
    def window_autocorr(self, smp, lag):
        smp = np.ascontiguousarray(smp, np.int32)
        out = np.zeros(MAX_LAGS, np.float64)
        self.L.fo_window_autocorr(_p(smp), len(smp), lag, _p(out))
        return out

    def lpc_calc_coefs(self, smp, max_order, precision, omethod):
        smp = np.ascontiguousarray(smp, np.int32)
        coefs = np.zeros((MAX_ORDER, MAX_ORDER), np.int32)
        shift = np.zeros(MAX_ORDER, np.int32)
        opt = self.L.fo_lpc_calc_coefs(_p(smp), len(smp), max_order, precision, omethod,
                                       _p(coefs), _p(shift))
        return coefs, shift, opt

    def levinson(self, autoc, max_order, ref=None):
        autoc_a = None if autoc is None else np.ascontiguousarray(autoc, np.float64)
        ref_a = None if ref is None else np.ascontiguousarray(ref, np.float64)
        lpc = np.zeros((MAX_ORDER, MAX_ORDER), np.float64)
        self.L.fo_levinson(_p(autoc_a), max_order, _p(ref_a), _p(lpc))
        return lpc

    def schur_order_est(self, autoc, max_order):
        autoc = np.ascontiguousarray(autoc, np.float64)
        lpc = np.zeros((MAX_ORDER, MAX_ORDER), np.float64)
        est = self.L.fo_schur_order_est(_p(autoc), max_order, _p(lpc))
        return est, lpc

    def quantize_coefs(self, row, order, precision):
        row = np.array(row, np.float64)
        out = np.zeros(MAX_ORDER, np.int32)
        sh = C.c_int(0)
        self.L.fo_quantize_coefs(_p(row), order, precision, _p(out), C.byref(sh))
        return out, sh.value

    # ---- residual / rice ----
    def residual_lpc(self, smp, order, coefs, shift):
        smp = np.ascontiguousarray(smp, np.int32)
        coefs = np.ascontiguousarray(coefs, np.int32)
        res = np.zeros_like(smp)
        self.L.fo_residual_lpc(_p(res), _p(smp), len(smp), order, _p(coefs), shift)
        return res

    def residual_fixed(self, smp, order):
        smp = np.ascontiguousarray(smp, np.int32)
        res = np.zeros_like(smp)
        self.L.fo_residual_fixed(_p(res), _p(smp), len(smp), order)
        return res

    def rice_best_k(self, s, n):
        return self.L.fo_rice_best_k(int(s) & (2**64 - 1), n)

    def subframe_bits(self, res, pmin, pmax, pred_order, bps, precision, lpc):
        res = np.ascontiguousarray(res, np.int32)
        sf = np.zeros(1, INFO_DTYPE)
        bits = self.L.fo_subframe_bits(_p(sf), pmin, pmax, _p(res), len(res), pred_order, bps,
                                       precision, int(lpc))
        return bits, sf[0]

    def encode_residual(self, params, smp, obits):
        smp = np.ascontiguousarray(smp, np.int32)
        fp = to_fo_params(params)
        sf = np.zeros(1, INFO_DTYPE)
        sf["obits"] = obits
        res = np.zeros_like(smp)
        rc = self.L.fo_encode_residual(C.byref(fp), _p(sf), _p(smp), _p(res), len(smp))
        return rc, sf[0], res

    def residual_section_bits(self, sf, res):
        sfa = np.array([sf], INFO_DTYPE)
        res = np.ascontiguousarray(res, np.int32)
        return self.L.fo_residual_section_bits(_p(sfa), _p(res), len(res))

    def emit_residual(self, sf, res, cap):
        sfa = np.array([sf], INFO_DTYPE)
        res = np.ascontiguousarray(res, np.int32)
        out = np.zeros(cap, np.uint8)
        nb = self.L.fo_emit_residual(_p(sfa), _p(res), len(res), _p(out), cap)
        return nb, out

    # ---- frame level ----
    def prepare_frame(self, params, pcm, n):
        fp = to_fo_params(params)
        pcm = np.ascontiguousarray(pcm, np.int32)
        smp = np.zeros((fp.channels, n), np.int32)
        sf = np.zeros(fp.channels, INFO_DTYPE)
        mode = self.L.fo_prepare_frame(C.byref(fp), _p(pcm), n, _p(smp), _p(sf))
        return mode, smp, sf

    def stereo_mode(self, left, right):
        left = np.ascontiguousarray(left, np.int32)
        right = np.ascontiguousarray(right, np.int32)
        return self.L.fo_stereo_mode(_p(left), _p(right), len(left))

    def encode_frame(self, params, frame_number, pcm, n, buf_size=None):
        fp = to_fo_params(params)
        pcm = np.ascontiguousarray(pcm, np.int32)
        if buf_size is None:
            buf_size = 2 * (16 + n * fp.channels * 4 + 64)
        out = np.zeros(buf_size, np.uint8)
        sf = np.zeros(fp.channels, INFO_DTYPE)
        res = np.zeros((fp.channels, n), np.int32)
        verb = C.c_int(0)
        rc = self.L.fo_encode_frame(C.byref(fp), frame_number, _p(pcm), n, _p(out), buf_size,
                                    _p(sf), _p(res), C.byref(verb))
        return rc, out[:max(rc, 0)].copy(), sf, res, verb.value

    def vbs_split(self, pcm, channels, block_size):
        pcm = np.ascontiguousarray(pcm, np.int32)
        sizes = np.zeros(8, np.int32)
        nf = C.c_int(0)
        self.L.fo_vbs_split(_p(pcm), channels, block_size, C.byref(nf), _p(sizes))
        return nf.value, sizes[:nf.value].copy()

    def encode_block(self, params, frame_count, pcm, block_size, buf_size):
        fp = to_fo_params(params)
        pcm = np.ascontiguousarray(pcm, np.int32)
        out = np.zeros(buf_size, np.uint8)
        fc = C.c_uint32(frame_count)
        rc = self.L.fo_encode_block(C.byref(fp), C.byref(fc), _p(pcm), block_size, _p(out), buf_size)
        return rc, out[:max(rc, 0)].copy(), fc.value

    def encode_subframes_batch(self, params, pcm, n, want_residual=True, slot_bytes=0):
        fp = to_fo_params(params)
        pcm = np.ascontiguousarray(pcm, np.int32).reshape(-1, n, fp.channels)
        nframes = pcm.shape[0]
        nsub = nframes * fp.channels
        sf = np.zeros(nsub, INFO_DTYPE)
        res = np.zeros((nframes, fp.channels, n), np.int32) if want_residual else None
        bits = np.zeros((nsub, slot_bytes), np.uint8) if slot_bytes else None
        rc = self.L.fo_encode_subframes_batch(C.byref(fp), _p(pcm), nframes, n, _p(sf), _p(res),
                                              _p(bits), slot_bytes)
        if rc != 0:
            raise RuntimeError("oracle batch failed")
        return {"info": sf, "residual": res, "rice_bits": bits}

    def crc8(self, data):
        d = np.ascontiguousarray(data, np.uint8)
        return self.L.fo_crc8(_p(d), len(d))

    def crc16(self, data):
        d = np.ascontiguousarray(data, np.uint8)
        return self.L.fo_crc16(_p(d), len(d))


DEC_SO = os.path.join(ORACLE_DIR, "libflac_decode.so")


class Decoder:
    """Independent FLAC frame decoder (oracle/flac_decode.c)."""

    def __init__(self):
        if _stale(DEC_SO, [os.path.join(ORACLE_DIR, "flac_decode.c")]):
            subprocess.run(["make", "-C", ORACLE_DIR, "libflac_decode.so"], check=True,
                           stdout=subprocess.DEVNULL)
        self.L = C.CDLL(DEC_SO)
        self.L.fd_decode_frames.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p,
                                            C.c_size_t, C.POINTER(C.c_int), C.c_void_p, C.c_int]
        self.L.fd_decode_frames.restype = C.c_long

    def decode(self, data, channels, bps, max_sample_frames):
        data = np.ascontiguousarray(data, np.uint8)
        pcm = np.zeros((max_sample_frames, channels), np.int32)
        nf = C.c_int(0)
        sizes = np.zeros(65536, np.int32)
        rc = self.L.fd_decode_frames(_p(data), len(data), channels, bps, _p(pcm), max_sample_frames,
                                     C.byref(nf), _p(sizes), len(sizes))
        if rc < 0:
            raise ValueError(f"FLAC decode error {rc}")
        return pcm[:rc], sizes[:nf.value].copy()


class Ref:
    """The real reference functions (only where oracle/_ref was built)."""

    @staticmethod
    def available() -> bool:
        build()
        return os.path.exists(REF_SO)

    def __init__(self):
        build()
        L = self.L = C.CDLL(REF_SO)
        vp, i, u64 = C.c_void_p, C.c_int, C.c_uint64
        L.ref_compute_autocorr.argtypes = [vp, i, i, vp]
        L.ref_compute_lpc_coefs.argtypes = [vp, i, vp, vp]
        L.ref_compute_lpc_coefs_est.argtypes = [vp, i, vp]
        L.ref_compute_lpc_coefs_est.restype = i
        L.ref_quantize_lpc_coefs.argtypes = [vp, i, i, vp, vp]
        L.ref_lpc_calc_coefs.argtypes = [vp, i, i, i, i, vp, vp]
        L.ref_lpc_calc_coefs.restype = i
        L.ref_find_optimal_rice_param.argtypes = [u64, i]
        L.ref_find_optimal_rice_param.restype = i
        L.ref_calc_rice_params.argtypes = [i, i, i, vp, i, i, i, i, vp, vp, vp]
        L.ref_calc_rice_params.restype = C.c_uint32
        L.ref_rice_encode_count.argtypes = [u64, i, i]
        L.ref_rice_encode_count.restype = u64
        L.ref_limit_max_partition_order.argtypes = [i, i, i]
        L.ref_limit_max_partition_order.restype = i
        L.ref_log2i.argtypes = [C.c_uint32]
        L.ref_log2i.restype = i
        L.ref_emit_residual.argtypes = [i, i, vp, i, vp, i, vp, i, C.POINTER(C.c_int64)]
        L.ref_emit_residual.restype = i
        L.ref_bitwriter_run.argtypes = [vp, vp, vp, i, vp, i]
        L.ref_bitwriter_run.restype = i
        L.ref_crc8.argtypes = [vp, C.c_uint32]
        L.ref_crc8.restype = i
        if hasattr(L, "ref_time_hotpath"):
            L.ref_time_hotpath.argtypes = [vp, vp, i, i, i, i, i, i, vp, vp, i, vp]
            L.ref_time_hotpath.restype = C.c_int64
        L.ref_crc16.argtypes = [vp, C.c_uint32]
        L.ref_crc16.restype = i

    def compute_autocorr(self, smp, lag):
        smp = np.ascontiguousarray(smp, np.int32)
        out = np.zeros(MAX_LAGS, np.float64)
        self.L.ref_compute_autocorr(_p(smp), len(smp), lag, _p(out))
        return out

    def compute_lpc_coefs(self, autoc, max_order, ref=None):
        autoc_a = None if autoc is None else np.ascontiguousarray(autoc, np.float64)
        ref_a = None if ref is None else np.ascontiguousarray(ref, np.float64)
        lpc = np.zeros((MAX_ORDER, MAX_ORDER), np.float64)
        self.L.ref_compute_lpc_coefs(_p(autoc_a), max_order, _p(ref_a), _p(lpc))
        return lpc

    def compute_lpc_coefs_est(self, autoc, max_order):
        autoc = np.ascontiguousarray(autoc, np.float64)
        lpc = np.zeros((MAX_ORDER, MAX_ORDER), np.float64)
        est = self.L.ref_compute_lpc_coefs_est(_p(autoc), max_order, _p(lpc))
        return est, lpc

    def quantize_lpc_coefs(self, row, order, precision):
        row = np.array(row, np.float64)
        out = np.zeros(MAX_ORDER, np.int32)
        sh = C.c_int(0)
        self.L.ref_quantize_lpc_coefs(_p(row), order, precision, _p(out), C.byref(sh))
        return out, sh.value

    def lpc_calc_coefs(self, smp, max_order, precision, omethod):
        smp = np.ascontiguousarray(smp, np.int32)
        # the reference leaves untouched rows uninitialised: pre-zero them
        coefs = np.zeros((MAX_ORDER, MAX_ORDER), np.int32)
        shift = np.zeros(MAX_ORDER, np.int32)
        opt = self.L.ref_lpc_calc_coefs(_p(smp), len(smp), max_order, precision, omethod,
                                        _p(coefs), _p(shift))
        return coefs, shift, opt

    def find_optimal_rice_param(self, s, n):
        return self.L.ref_find_optimal_rice_param(int(s) & (2**64 - 1), n)

    def rice_encode_count(self, s, n, k):
        return self.L.ref_rice_encode_count(int(s) & (2**64 - 1), n, k)

    def calc_rice_params(self, lpc, pmin, pmax, res, pred_order, bps, precision):
        res = np.array(res, np.int32)      # the reference takes a non-const pointer
        method, porder = C.c_int(0), C.c_int(0)
        params = np.zeros(MAX_PARTS, np.int32)
        bits = self.L.ref_calc_rice_params(int(lpc), pmin, pmax, _p(res), len(res), pred_order,
                                           bps, precision, C.byref(method), C.byref(porder),
                                           _p(params))
        return bits, method.value, porder.value, params

    def emit_residual(self, method, porder, params, order, res, cap):
        params = np.ascontiguousarray(params, np.int32)
        res = np.ascontiguousarray(res, np.int32)
        out = np.zeros(cap, np.uint8)
        nb = C.c_int64(0)
        rc = self.L.ref_emit_residual(method, porder, _p(params), order, _p(res), len(res),
                                      _p(out), cap, C.byref(nb))
        return rc, nb.value, out

    def bitwriter_run(self, nbits, vals, signed, cap):
        nbits = np.ascontiguousarray(nbits, np.int32)
        vals = np.ascontiguousarray(vals, np.int32)
        signed = np.ascontiguousarray(signed, np.uint8)
        out = np.zeros(cap, np.uint8)
        rc = self.L.ref_bitwriter_run(_p(nbits), _p(vals), _p(signed), len(nbits), _p(out), cap)
        return rc, out

    def time_hotpath(self, smp, obits, max_order, precision, pmin, pmax):
        """bench.py's cpu_baseline leg: seconds per stage (lpc, fir, rice, emit, crc16) of the
        compiled reference stages over prepared subframes smp[nsub][n]; returns (bits, t[5])."""
        smp = np.ascontiguousarray(smp, np.int32)
        nsub, n = smp.shape
        ob = np.ascontiguousarray(obits, np.int32)
        res = np.zeros(n, np.int32)
        cap = 8 * n + 64
        out = np.zeros(cap, np.uint8)
        t = np.zeros(5, np.float64)
        bits = self.L.ref_time_hotpath(_p(smp), _p(ob), nsub, n, max_order, precision, pmin, pmax,
                                       _p(res), _p(out), cap, _p(t))
        return int(bits), t

    def crc8(self, data):
        d = np.ascontiguousarray(data, np.uint8)
        return self.L.ref_crc8(_p(d), len(d))

    def crc16(self, data):
        d = np.ascontiguousarray(data, np.uint8)
        return self.L.ref_crc16(_p(d), len(d))
