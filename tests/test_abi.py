"""The C-ABI shared library: loads here (no GPU) and exports what the header declares."""
import ctypes as C
import os
import re

import flake_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, txt)))


def test_every_declared_symbol_is_exported():
    lib = flake_amd.load_library()
    names = declared("flakehip.h", "fhip_")
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(flake_amd.ABI_SYMBOLS)


def test_host_library_symbols():
    lib = flake_amd.load_host_library()
    for n in declared("flake_amd.h", "flake_amd_"):
        assert hasattr(lib, n), n


def test_struct_layouts_match_header():
    assert C.sizeof(flake_amd.Params) == 14 * 4
    assert flake_amd.INFO_DTYPE.itemsize == 4 * (12 + 32 + 256 + 32)
    assert C.sizeof(flake_amd.Batch) == 8 + 4 + 4 + 8 * 3 + 8 + 8 * 2 + 8 + 8 + 8 + 8 + 8


def test_strerror_and_version():
    lib = flake_amd.load_library()
    assert lib.fhip_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4, -5):
        assert lib.fhip_strerror(code)
    assert b"gfx950" in lib.fhip_version()


def test_create_rejects_what_flake_validate_params_rejects():
    """encode.c:268-373: these never reach the device."""
    lib = flake_amd.load_library()
    bad = [
        dict(channels=0), dict(channels=9), dict(bits_per_sample=3), dict(bits_per_sample=33),
        dict(sample_rate=0), dict(order_method=7), dict(stereo_method=2), dict(block_size=15),
        dict(prediction_type=3), dict(min_prediction_order=9, max_prediction_order=8),
        dict(min_prediction_order=0), dict(max_prediction_order=33),
        dict(min_partition_order=6, max_partition_order=5), dict(max_partition_order=9),
        dict(variable_block_size=1, allow_vbs=0), dict(prediction_type=1, max_prediction_order=5),
    ]
    for kw in bad:
        p = flake_amd.level_params(5, **kw)
        h = C.c_void_p()
        rc = lib.fhip_create(C.byref(h), 0, C.byref(p), 4)
        assert rc == flake_amd.E_INVALID, (kw, rc)
        assert not h.value
    p = flake_amd.level_params(5, block_size=65536)          # above FLAC's and libflake's 65535
    h = C.c_void_p()
    assert lib.fhip_create(C.byref(h), 0, C.byref(p), 4) in (flake_amd.E_INVALID, flake_amd.E_UNSUPPORTED)
    assert not h.value


def test_null_handles_are_errors_not_crashes():
    lib = flake_amd.load_library()
    assert lib.fhip_sync(None) == flake_amd.E_INVALID
    assert lib.fhip_set_profiling(None, 1) == flake_amd.E_INVALID
    lib.fhip_destroy(None)
    b = flake_amd.Batch()
    assert lib.fhip_encode_subframes_dev(None, C.byref(b)) == flake_amd.E_INVALID


def test_level_presets_match_oracle_table(oracle):
    """flake_set_defaults (encode.c:158-266): Python preset table == oracle's."""
    import oraclelib
    for lvl in range(13):
        q = oraclelib.FoParams()
        oracle.L.fo_set_defaults(C.byref(q), lvl)
        p = flake_amd.level_params(lvl)
        for k in ("order_method", "stereo_method", "block_size", "prediction_type",
                  "min_prediction_order", "max_prediction_order", "min_partition_order",
                  "max_partition_order", "variable_block_size", "allow_vbs", "lpc_precision"):
            assert getattr(p, k) == getattr(q, k), (lvl, k)
