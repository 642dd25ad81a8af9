"""flake_amd -- MI355X-native FLAC prediction/entropy path behind libflake's
``flake_encode_frame()`` surface.

This package is only the Python view (ctypes) of two native libraries:

* ``lib/libflakehip.so``  -- gfx950 kernels + the C ABI of ``include/flakehip.h``
* ``lib/libflake_amd.so`` -- the host C layer of ``include/flake_amd.h``

There is no Python or CPU implementation of the path here: if the HIP library
is missing, importing the encoder fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

MAX_ORDER = 32
MAX_PARTS = 256
MAX_LAGS = 33
MAX_BLOCK = 65535

OK, E_GENERIC, E_HIP, E_UNSUPPORTED, E_INVALID, E_NOMEM = 0, -1, -2, -3, -4, -5

SUB_CONSTANT, SUB_VERBATIM, SUB_FIXED, SUB_LPC = 0, 1, 8, 32
CH_NOT_STEREO, CH_LEFT_RIGHT, CH_LEFT_SIDE, CH_RIGHT_SIDE, CH_MID_SIDE = 0, 1, 8, 9, 10

# FLAKE_ORDER_METHOD_* (flake.h:38-46), FLAKE_PREDICTION_* (flake.h:53-57)
OM_MAX, OM_EST, OM_2LEVEL, OM_4LEVEL, OM_8LEVEL, OM_SEARCH, OM_LOG = range(7)
PRED_NONE, PRED_FIXED, PRED_LEVINSON = range(3)
STEREO_INDEPENDENT, STEREO_ESTIMATE = range(2)


class Params(C.Structure):
    """``fhip_params``: the fields of FlakeContext/FlakeEncodeParams the path reads."""
    _fields_ = [(k, C.c_int) for k in (
        "channels", "sample_rate", "bits_per_sample", "block_size", "order_method",
        "stereo_method", "prediction_type", "min_prediction_order", "max_prediction_order",
        "min_partition_order", "max_partition_order", "variable_block_size", "allow_vbs",
        "lpc_precision")]

    def copy(self) -> "Params":
        q = Params()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(Params))
        return q

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


def level_params(level: int, channels: int = 2, bits_per_sample: int = 16,
                 sample_rate: int = 44100, **over) -> Params:
    """Compression-level presets, the table of flake_set_defaults() (encode.c:158-266)."""
    if not 0 <= level <= 12:
        raise ValueError("compression level must be 0..12")
    p = Params(channels=channels, sample_rate=sample_rate, bits_per_sample=bits_per_sample,
               block_size=4096, order_method=OM_EST, stereo_method=STEREO_ESTIMATE,
               prediction_type=PRED_LEVINSON, min_prediction_order=1, max_prediction_order=8,
               min_partition_order=0, max_partition_order=5, variable_block_size=0,
               allow_vbs=0, lpc_precision=15)
    if level <= 2:
        p.block_size = 1152
        p.prediction_type = PRED_FIXED
        p.min_prediction_order = (2, 2, 0)[level]
        p.max_prediction_order = (2, 4, 4)[level]
        p.max_partition_order = 3
        if level == 0:
            p.stereo_method = STEREO_INDEPENDENT
    elif level == 3:
        p.stereo_method = STEREO_INDEPENDENT
        p.max_prediction_order = 6
        p.max_partition_order = 4
    elif level == 4:
        p.max_partition_order = 4
    elif level in (6, 7):
        p.max_partition_order = 6
        if level == 7:
            p.order_method = OM_4LEVEL
    elif level >= 8:
        p.order_method = OM_SEARCH if level in (10, 12) else OM_LOG
        p.max_prediction_order = 32 if level >= 11 else 12
        p.max_partition_order = 6 if level == 8 else 8
        if level >= 11:
            p.block_size = 8192
        if level >= 9:
            p.allow_vbs = 1
            p.variable_block_size = 1
    for k, v in over.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


# numpy view of fhip_subframe_info (1328 bytes)
INFO_DTYPE = np.dtype([
    ("type", "<i4"), ("type_code", "<i4"), ("order", "<i4"), ("shift", "<i4"),
    ("obits", "<i4"), ("wasted", "<i4"), ("rice_method", "<i4"), ("porder", "<i4"),
    ("est_bits", "<u4"), ("ch_mode", "<i4"), ("rice_nbits", "<i4"), ("reserved", "<i4"),
    ("coefs", "<i4", (MAX_ORDER,)), ("rparams", "<i4", (MAX_PARTS,)),
    ("warmup", "<i4", (MAX_ORDER,)),
])
assert INFO_DTYPE.itemsize == 1328


class Batch(C.Structure):
    """``fhip_batch``"""
    _fields_ = [
        ("pcm", C.c_void_p), ("nframes", C.c_int), ("block_size", C.c_int),
        ("info", C.c_void_p), ("residual", C.c_void_p), ("rice_bits", C.c_void_p),
        ("rice_slot_bytes", C.c_int64), ("samples", C.c_void_p), ("autoc", C.c_void_p),
        ("frames", C.c_void_p), ("frame_stride", C.c_int64), ("frame_bytes", C.c_void_p),
        ("first_frame_number", C.c_uint32), ("frame_numbers", C.c_void_p),
    ]


class VbsOut(C.Structure):
    """``fhip_vbs_out`` (device pointers)"""
    _fields_ = [("packed", C.c_void_p), ("packed_cap", C.c_int64), ("frame_bytes", C.c_void_p),
                ("block_bytes", C.c_void_p), ("block_frames", C.c_void_p), ("totals", C.c_void_p)]


class FlakeHipError(RuntimeError):
    def __init__(self, code: int, what: str, detail: str = ""):
        self.code = code
        super().__init__(f"{what}: {code} ({detail})" if detail else f"{what}: {code}")


_lib = None


def load_library() -> C.CDLL:
    """Load libflakehip.so; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("FHIP_LIB") or os.path.join(LIB_DIR, "libflakehip.so")
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -m flake_amd.build` "
            "(the HIP library is the only implementation of this path)")
    lib = C.CDLL(path)
    vp, i, i64 = C.c_void_p, C.c_int, C.c_int64
    sig = {
        "fhip_device_count": (i, []),
        "fhip_frame_stride": (i64, [C.POINTER(Params), i]),
        "fhip_create": (i, [C.POINTER(vp), i, C.POINTER(Params), i]),
        "fhip_destroy": (None, [vp]),
        "fhip_set_stream": (i, [vp, vp]),
        "fhip_sync": (i, [vp]),
        "fhip_strerror": (C.c_char_p, [i]),
        "fhip_last_error": (C.c_char_p, [vp]),
        "fhip_version": (C.c_char_p, []),
        "fhip_encode_subframes_dev": (i, [vp, C.POINTER(Batch)]),
        "fhip_encode_subframes": (i, [vp, C.POINTER(Batch)]),
        "fhip_prepare_ahead": (i, [vp, C.POINTER(Batch)]),
        "fhip_encode_frames_packed": (i, [vp, C.POINTER(Batch), vp, i64, C.POINTER(i64)]),
        "fhip_frames_packed_begin": (i, [vp, C.POINTER(Batch), C.POINTER(i64)]),
        "fhip_frames_packed_fetch": (i, [vp, vp, i64]),
        "fhip_encode_blocks_vbs_packed": (i, [vp, vp, i, i, C.c_uint32, vp, i64, vp, vp, C.POINTER(i64),
                                              C.POINTER(i), C.POINTER(C.c_uint32)]),
        "fhip_encode_blocks_vbs_dev": (i, [vp, vp, i, i, C.c_uint32, C.POINTER(VbsOut)]),
        "fhip_lpc_calc_coefs": (i, [vp, vp, i, i, i, i, i, vp, vp, vp, vp]),
        "fhip_encode_residual": (i, [vp, vp, i, i, vp, vp, vp, i64]),
        "fhip_order_search_bits": (i, [vp, vp, i, i, vp, vp]),
        "fhip_prepare_frames": (i, [vp, vp, i, i, vp, vp]),
        "fhip_calc_rice_params": (i, [vp, vp, i, i, i, i, i, i, i, vp, vp, i64]),
        "fhip_vbs_split": (i, [vp, vp, i, i, vp, vp]),
        "fhip_set_profiling": (i, [vp, i]),
        "fhip_get_kernel_times": (i, [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                      C.POINTER(i), i, i]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


ABI_SYMBOLS = (
    "fhip_frame_stride", "fhip_device_count", "fhip_create", "fhip_destroy", "fhip_set_stream", "fhip_sync",
    "fhip_strerror", "fhip_last_error", "fhip_version", "fhip_encode_subframes_dev",
    "fhip_encode_subframes", "fhip_lpc_calc_coefs", "fhip_encode_residual",
    "fhip_prepare_frames", "fhip_calc_rice_params", "fhip_vbs_split", "fhip_set_profiling",
    "fhip_get_kernel_times", "fhip_prepare_ahead", "fhip_encode_frames_packed",
    "fhip_frames_packed_begin", "fhip_frames_packed_fetch", "fhip_encode_blocks_vbs_packed",
    "fhip_encode_blocks_vbs_dev", "fhip_order_search_bits",
    "fhip_host_alloc", "fhip_host_free", "fhip_host_register", "fhip_host_unregister", "fhip_frames_packed_upload", "fhip_frames_packed_fetch_async", "fhip_frames_packed_fetch_wait",
)


def _ptr(a) -> int | None:
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("arrays handed to the C ABI must be C-contiguous")
        return a.ctypes.data
    return int(a.data_ptr())        # torch tensor


def rice_slot_bytes(p: Params, n: int) -> int:
    """Slot that any residual section of a frame that will not fall back to
    verbatim fits in: the frame's verbatim size (encode.c:521-527), rounded up."""
    bps = p.bits_per_sample
    if p.channels == 2:
        v = 16 + ((n * (bps + bps + 1) + 7) >> 3)
    else:
        v = 16 + ((n * p.channels * bps + 7) >> 3)
    return (v + 3) & ~3


class Encoder:
    """One ``fhip_ctx``: a device, a stream, workspaces for ``max_frames`` frames."""

    def __init__(self, params: Params, max_frames: int, device: int = 0):
        self.lib = load_library()
        self.params = params.copy()
        self.max_frames = int(max_frames)
        h = C.c_void_p()
        rc = self.lib.fhip_create(C.byref(h), int(device), C.byref(self.params), self.max_frames)
        if rc != OK:
            raise FlakeHipError(rc, "fhip_create", self.lib.fhip_strerror(rc).decode())
        self._h = h

    # -- plumbing ---------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            self.lib.fhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int, what: str) -> None:
        if rc != OK:
            raise FlakeHipError(rc, what, self.lib.fhip_last_error(self._h).decode())

    def set_stream(self, stream_handle: int | None) -> None:
        self._check(self.lib.fhip_set_stream(self._h, stream_handle), "fhip_set_stream")

    def sync(self) -> None:
        self._check(self.lib.fhip_sync(self._h), "fhip_sync")

    def set_profiling(self, on: bool) -> None:
        self._check(self.lib.fhip_set_profiling(self._h, int(on)), "fhip_set_profiling")

    def kernel_times(self, reset: bool = True) -> dict:
        names = (C.c_char_p * 8)()
        ms = (C.c_double * 8)()
        cnt = (C.c_int * 8)()
        k = self.lib.fhip_get_kernel_times(self._h, names, ms, cnt, 8, int(reset))
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(k)}

    # -- hot path ---------------------------------------------------------
    def frame_stride(self, block_size: int) -> int:
        return int(self.lib.fhip_frame_stride(C.byref(self.params), block_size))

    def encode_subframes_dev(self, pcm, nframes: int, block_size: int, info, residual=None,
                             rice_bits=None, slot_bytes: int = 0, samples=None, autoc=None,
                             frames=None, frame_stride: int = 0, frame_bytes=None,
                             first_frame_number: int = 0) -> None:
        """Device-resident batch (torch tensors or raw device addresses); async."""
        b = Batch(_ptr(pcm), nframes, block_size, _ptr(info), _ptr(residual), _ptr(rice_bits),
                  slot_bytes, _ptr(samples), _ptr(autoc), _ptr(frames), frame_stride,
                  _ptr(frame_bytes), first_frame_number, None)
        self._check(self.lib.fhip_encode_subframes_dev(self._h, C.byref(b)),
                    "fhip_encode_subframes_dev")

    def encode_blocks_vbs_dev(self, pcm, nblocks: int, block_size: int, packed, packed_cap: int, totals,
                              frame_bytes=None, block_bytes=None, block_frames=None,
                              first_frame_number: int = 0) -> None:
        """Device-resident variable-block-size batch (vbs.c:85-119 per block); async, no host sync."""
        o = VbsOut(_ptr(packed), packed_cap, _ptr(frame_bytes), _ptr(block_bytes), _ptr(block_frames),
                   _ptr(totals))
        self._check(self.lib.fhip_encode_blocks_vbs_dev(self._h, _ptr(pcm), nblocks, block_size,
                                                        first_frame_number, C.byref(o)),
                    "fhip_encode_blocks_vbs_dev")

    def prepare_ahead(self, pcm, nframes: int, block_size: int) -> None:
        """Hint: start the feeder stage (K0) of the NEXT device-resident batch now, beside the
        batch in flight; the next encode_subframes_dev() call must be for this batch."""
        b = Batch(_ptr(pcm), nframes, block_size, None, None, None, 0, None, None, None, 0,
                  None, 0, None)
        self._check(self.lib.fhip_prepare_ahead(self._h, C.byref(b)), "fhip_prepare_ahead")

    def encode_subframes(self, pcm: np.ndarray, block_size: int, want_residual: bool = True,
                         want_bits: bool = True, want_samples: bool = False,
                         want_autoc: bool = False, want_frames: bool = False,
                         first_frame_number: int = 0) -> dict:
        """Host numpy batch: pcm is [nframes][block_size][channels] int32."""
        ch = self.params.channels
        pcm = np.ascontiguousarray(pcm, dtype=np.int32).reshape(-1, block_size, ch)
        nframes = pcm.shape[0]
        nsub = nframes * ch
        out = {"info": np.zeros(nsub, dtype=INFO_DTYPE)}
        slot = rice_slot_bytes(self.params, block_size)
        if want_residual:
            out["residual"] = np.zeros((nframes, ch, block_size), dtype=np.int32)
        if want_bits:
            out["rice_bits"] = np.zeros((nsub, slot), dtype=np.uint8)
        if want_samples:
            out["samples"] = np.zeros((nframes, ch, block_size), dtype=np.int32)
        if want_autoc:
            out["autoc"] = np.zeros((nsub, MAX_LAGS), dtype=np.float64)
        stride = 0
        if want_frames:
            stride = self.frame_stride(block_size)
            out["frames"] = np.zeros((nframes, stride), dtype=np.uint8)
            out["frame_bytes"] = np.zeros(nframes, dtype=np.int32)
        b = Batch(_ptr(pcm), nframes, block_size, _ptr(out["info"]), _ptr(out.get("residual")),
                  _ptr(out.get("rice_bits")), slot, _ptr(out.get("samples")),
                  _ptr(out.get("autoc")), _ptr(out.get("frames")), stride,
                  _ptr(out.get("frame_bytes")), first_frame_number, None)
        self._check(self.lib.fhip_encode_subframes(self._h, C.byref(b)), "fhip_encode_subframes")
        out["slot_bytes"] = slot
        return out

    # -- stage entry points ----------------------------------------------
    def lpc_calc_coefs(self, samples: np.ndarray, max_order: int, precision: int, omethod: int):
        """lpc_calc_coefs() (lpc.c:224-257) over [nsub][n] blocks."""
        samples = np.ascontiguousarray(samples, dtype=np.int32)
        nsub, n = samples.shape
        coefs = np.zeros((nsub, MAX_ORDER, MAX_ORDER), dtype=np.int32)
        shift = np.zeros((nsub, MAX_ORDER), dtype=np.int32)
        opt = np.zeros(nsub, dtype=np.int32)
        autoc = np.zeros((nsub, MAX_LAGS), dtype=np.float64)
        self._check(self.lib.fhip_lpc_calc_coefs(
            self._h, _ptr(samples), nsub, n, max_order, precision, omethod,
            _ptr(coefs), _ptr(shift), _ptr(opt), _ptr(autoc)), "fhip_lpc_calc_coefs")
        return coefs, shift, opt, autoc

    def encode_residual(self, samples: np.ndarray, obits, want_bits: bool = True) -> dict:
        """encode_residual() (optimize.c:124-276) over prepared [nsub][n] blocks."""
        samples = np.ascontiguousarray(samples, dtype=np.int32)
        nsub, n = samples.shape
        info = np.zeros(nsub, dtype=INFO_DTYPE)
        info["obits"] = obits
        res = np.zeros((nsub, n), dtype=np.int32)
        slot = rice_slot_bytes(self.params, n)
        bits = np.zeros((nsub, slot), dtype=np.uint8) if want_bits else None
        self._check(self.lib.fhip_encode_residual(
            self._h, _ptr(samples), nsub, n, _ptr(info), _ptr(res), _ptr(bits), slot),
            "fhip_encode_residual")
        return {"info": info, "residual": res, "rice_bits": bits, "slot_bytes": slot}

    def order_search_bits(self, samples: np.ndarray, obits, magbits=None) -> np.ndarray:
        """The bits[] table of encode_residual()'s LPC order search (optimize.c:201-261) over prepared
        [nsub][n] blocks: [nsub][32] uint32, 0xFFFFFFFF = order not visited / constant block.
        magbits (|x| < 2^magbits per block, optional) lets 16-bit blocks take the packed FIRs."""
        samples = np.ascontiguousarray(samples, dtype=np.int32)
        nsub, n = samples.shape
        info = np.zeros(nsub, dtype=INFO_DTYPE)
        info["obits"] = obits
        if magbits is not None:
            info["reserved"] = (1 + np.asarray(magbits, dtype=np.int64)) << 8
        bits = np.zeros((nsub, 32), dtype=np.uint32)
        self._check(self.lib.fhip_order_search_bits(self._h, _ptr(samples), nsub, n, _ptr(info), _ptr(bits)),
                    "fhip_order_search_bits")
        return bits

    def calc_rice_params(self, residual: np.ndarray, pred_order: int, lpc: bool, bps: int,
                         pmin: int, pmax: int, slot_bytes: int = 0) -> dict:
        """calc_rice_params_lpc/_fixed (rice.c:173-187) + residual emit on given residuals."""
        residual = np.ascontiguousarray(residual, dtype=np.int32)
        nsub, n = residual.shape
        info = np.zeros(nsub, dtype=INFO_DTYPE)
        bits = np.zeros((nsub, slot_bytes), dtype=np.uint8) if slot_bytes else None
        self._check(self.lib.fhip_calc_rice_params(
            self._h, _ptr(residual), nsub, n, pred_order, int(lpc), bps, pmin, pmax,
            _ptr(info), _ptr(bits), slot_bytes), "fhip_calc_rice_params")
        return {"info": info, "rice_bits": bits}

    def vbs_split(self, pcm: np.ndarray, block_size: int):
        """split_frame_v1 (vbs.c:36-83) over [nblocks][block_size][channels] blocks."""
        ch = self.params.channels
        pcm = np.ascontiguousarray(pcm, dtype=np.int32).reshape(-1, block_size, ch)
        nb = pcm.shape[0]
        frames = np.zeros(nb, dtype=np.int32)
        sizes = np.zeros((nb, 8), dtype=np.int32)
        self._check(self.lib.fhip_vbs_split(self._h, _ptr(pcm), nb, block_size, _ptr(frames),
                                            _ptr(sizes)), "fhip_vbs_split")
        return frames, sizes

    def prepare_frames(self, pcm: np.ndarray, block_size: int):
        """copy_samples + channel_decorrelation + remove_wasted_bits (encode.c:541-694)."""
        ch = self.params.channels
        pcm = np.ascontiguousarray(pcm, dtype=np.int32).reshape(-1, block_size, ch)
        nframes = pcm.shape[0]
        smp = np.zeros((nframes, ch, block_size), dtype=np.int32)
        info = np.zeros(nframes * ch, dtype=INFO_DTYPE)
        self._check(self.lib.fhip_prepare_frames(
            self._h, _ptr(pcm), nframes, block_size, _ptr(smp), _ptr(info)),
            "fhip_prepare_frames")
        return smp, info


# ---- host C layer ------------------------------------------------------------

_host = None


def load_host_library() -> C.CDLL:
    global _host
    if _host is not None:
        return _host
    load_library()          # libflake_amd.so links against libflakehip.so
    path = os.path.join(LIB_DIR, "libflake_amd.so")
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `python -m flake_amd.build`")
    lib = C.CDLL(path)
    lib.flake_amd_synth_pcm.restype = None
    lib.flake_amd_synth_pcm.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
    cp = C.POINTER(HostContext)
    lib.flake_amd_set_defaults.argtypes = [C.POINTER(HostParams)]
    lib.flake_amd_validate_params.argtypes = [cp]
    lib.flake_amd_encode_init.argtypes = [cp]
    lib.flake_amd_get_buffer.argtypes = [cp]
    lib.flake_amd_get_buffer.restype = C.c_void_p
    lib.flake_amd_encode_frame.argtypes = [cp, C.c_void_p, C.c_int]
    lib.flake_amd_encode_frames.argtypes = [cp, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_size_t, C.c_void_p]
    lib.flake_amd_encode_frames.restype = C.c_longlong
    lib.flake_amd_pin_buffers.argtypes = [cp, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.flake_amd_pin_buffers.restype = C.c_int
    lib.flake_amd_encode_close.argtypes = [cp]
    lib.flake_amd_encode_close.restype = None
    lib.flake_amd_get_streaminfo.argtypes = [cp, C.POINTER(HostStreaminfo)]
    lib.flake_amd_write_streaminfo.argtypes = [C.POINTER(HostStreaminfo), C.c_void_p]
    lib.flake_amd_write_streaminfo.restype = None
    lib.flake_amd_get_version.restype = C.c_char_p
    lib.flake_amd_last_error.argtypes = [cp]
    lib.flake_amd_last_error.restype = C.c_char_p
    _host = lib
    return lib


class HostParams(C.Structure):
    """``FlakeAmdEncodeParams`` (layout of FlakeEncodeParams, flake.h:59-161)."""
    _fields_ = [(k, C.c_int) for k in (
        "compression", "order_method", "stereo_method", "block_size", "padding_size",
        "min_prediction_order", "max_prediction_order", "prediction_type",
        "min_partition_order", "max_partition_order", "variable_block_size", "allow_vbs")]


class HostContext(C.Structure):
    """``FlakeAmdContext`` (layout of FlakeContext, flake.h:163-215)."""
    _fields_ = [("channels", C.c_int), ("sample_rate", C.c_int), ("bits_per_sample", C.c_int),
                ("samples", C.c_uint), ("params", HostParams), ("header", C.c_void_p),
                ("private_ctx", C.c_void_p)]


class HostStreaminfo(C.Structure):
    _fields_ = [(k, C.c_uint) for k in (
        "min_block_size", "max_block_size", "min_frame_size", "max_frame_size", "sample_rate",
        "channels", "bits_per_sample", "samples")] + [("md5sum", C.c_ubyte * 16)]


class HostEncoder:
    """The host C layer (include/flake_amd.h): libflake's call sequence
    set_defaults -> validate -> encode_init -> encode_frame(s) -> close."""

    def __init__(self, level: int = 5, channels: int = 2, bits_per_sample: int = 16,
                 sample_rate: int = 44100, samples: int = 0, **over):
        self.lib = load_host_library()
        self.ctx = HostContext(channels=channels, sample_rate=sample_rate,
                               bits_per_sample=bits_per_sample, samples=samples)
        self.ctx.params.compression = level
        if self.lib.flake_amd_set_defaults(C.byref(self.ctx.params)) != 0:
            raise ValueError("flake_amd_set_defaults")
        for k, v in over.items():
            if not hasattr(self.ctx.params, k):
                raise AttributeError(k)
            setattr(self.ctx.params, k, v)
        self.subset = self.lib.flake_amd_validate_params(C.byref(self.ctx))
        if self.subset < 0:
            raise ValueError("flake_amd_validate_params rejected the parameters")
        n = self.lib.flake_amd_encode_init(C.byref(self.ctx))
        if n < 0:
            raise FlakeHipError(n, "flake_amd_encode_init")
        self.header = bytes(C.string_at(self.ctx.header, n))
        self.open = True

    def params(self) -> Params:
        hp = self.ctx.params
        return Params(channels=self.ctx.channels, sample_rate=self.ctx.sample_rate,
                      bits_per_sample=self.ctx.bits_per_sample, block_size=hp.block_size,
                      order_method=hp.order_method, stereo_method=hp.stereo_method,
                      prediction_type=hp.prediction_type,
                      min_prediction_order=hp.min_prediction_order,
                      max_prediction_order=hp.max_prediction_order,
                      min_partition_order=hp.min_partition_order,
                      max_partition_order=hp.max_partition_order,
                      variable_block_size=hp.variable_block_size, allow_vbs=hp.allow_vbs,
                      lpc_precision=15)

    def encode_frames(self, pcm: np.ndarray, block_size: int, tail_size: int = 0):
        """pcm: [nblocks*block_size + tail_size][channels] int32.  Returns (bytes, sizes)."""
        ch = self.ctx.channels
        pcm = np.ascontiguousarray(pcm, dtype=np.int32).reshape(-1, ch)
        nblocks = (pcm.shape[0] - tail_size) // block_size
        assert nblocks * block_size + tail_size == pcm.shape[0]
        cap = 64 + pcm.size * 5 + 64 * (nblocks + 1) * 8
        out = np.zeros(cap, dtype=np.uint8)
        sizes = np.zeros(nblocks + (1 if tail_size else 0), dtype=np.int32)
        w = self.lib.flake_amd_encode_frames(C.byref(self.ctx), pcm.ctypes.data, nblocks, block_size,
                                             tail_size, out.ctypes.data, cap, sizes.ctypes.data)
        if w < 0:
            raise FlakeHipError(int(w), "flake_amd_encode_frames",
                                self.lib.flake_amd_last_error(C.byref(self.ctx)).decode())
        return out[:w].copy(), sizes

    def encode_frame(self, pcm: np.ndarray) -> bytes:
        """flake_encode_frame(): one block, result in the library's frame buffer."""
        ch = self.ctx.channels
        pcm = np.ascontiguousarray(pcm, dtype=np.int32).reshape(-1, ch)
        w = self.lib.flake_amd_encode_frame(C.byref(self.ctx), pcm.ctypes.data, pcm.shape[0])
        if w < 0:
            raise FlakeHipError(w, "flake_amd_encode_frame",
                                self.lib.flake_amd_last_error(C.byref(self.ctx)).decode())
        return bytes(C.string_at(self.lib.flake_amd_get_buffer(C.byref(self.ctx)), w))

    def streaminfo(self) -> HostStreaminfo:
        si = HostStreaminfo()
        if self.lib.flake_amd_get_streaminfo(C.byref(self.ctx), C.byref(si)) != 0:
            raise RuntimeError("flake_amd_get_streaminfo")
        return si

    def close(self) -> None:
        if getattr(self, "open", False):
            self.lib.flake_amd_encode_close(C.byref(self.ctx))
            self.open = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def synth_pcm(nframes: int, n: int, channels: int, bps: int, first_frame: int = 0) -> np.ndarray:
    """Deterministic synthetic PCM, [nframes][n][channels] int32."""
    lib = load_host_library()
    out = np.empty((nframes, n, channels), dtype=np.int32)
    lib.flake_amd_synth_pcm(out.ctypes.data, first_frame, nframes, n, channels, bps)
    return out
