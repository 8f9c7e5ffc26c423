"""ORACLE (test infrastructure): ctypes binding of oracle/libsvx_oracle.so (svx_oracle.c)."""
import ctypes
import os

import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsvx_oracle.so")
GAP_DTYPE = np.dtype([("aln", "<u4"), ("op", "<u4"), ("read_pos", "<i4"),
                      ("ref_pos", "<i4"), ("len", "<i4"), ("kind", "<u4")])
_lib = None


def load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(LIB_PATH)
        vp = ctypes.c_void_p
        lib.oracle_rasterize.restype = ctypes.c_int
        lib.oracle_rasterize.argtypes = [vp, ctypes.c_uint32, vp, ctypes.c_int, vp]
        lib.oracle_cigar_scan.restype = ctypes.c_int
        lib.oracle_cigar_scan.argtypes = [vp, vp, vp, ctypes.c_uint32, ctypes.c_int32, vp, ctypes.c_uint64, vp, vp]
        _lib = lib
    return _lib


def rasterize(records, layout="NHWC", mean=(104.0, 117.0, 124.0)):
    """records int32 [n,12] -> float32 [n,227,227,3] or [n,3,227,227]."""
    lib = load()
    rec = np.ascontiguousarray(records, np.int32).reshape(-1, 12)
    n = rec.shape[0]
    shape = (n, 227, 227, 3) if layout == "NHWC" else (n, 3, 227, 227)
    out = np.empty(shape, np.float32)
    m = np.asarray(mean, np.float32)
    rc = lib.oracle_rasterize(rec.ctypes.data, n, out.ctypes.data, 0 if layout == "NHWC" else 1, m.ctypes.data)
    assert rc == 0
    return out


def cigar_scan(cigar, cig_off, ref_start, min_sv, cap=None):
    """-> (gaps structured array, gap_off uint32[n+1], stats int32[n,4])."""
    lib = load()
    cigar = np.ascontiguousarray(cigar, np.uint32)
    cig_off = np.ascontiguousarray(cig_off, np.uint64)
    ref_start = np.ascontiguousarray(ref_start, np.int32)
    n = ref_start.size
    if cap is None:
        cap = max(16, cigar.size)
    gaps = np.empty(cap, GAP_DTYPE)
    stats = np.empty((n, 4), np.int32)
    cnt = ctypes.c_uint64(0)
    rc = lib.oracle_cigar_scan(cigar.ctypes.data, cig_off.ctypes.data, ref_start.ctypes.data, n, int(min_sv),
                               gaps.ctypes.data, cap, ctypes.byref(cnt), stats.ctypes.data)
    assert rc == 0, rc
    gaps = gaps[: cnt.value]
    off = np.zeros(n + 1, np.uint32)
    np.add.at(off, gaps["aln"].astype(np.int64) + 1, 1)
    return gaps, np.cumsum(off, dtype=np.uint32), stats
