"""CPU test: libsvx.so loads and exports every symbol include/svx.h (the product contract) and include/svx_experimental.h
(implementations the default path never calls) declare (no compute)."""
import os
import re

from svision_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_header_symbols():
    header = open(os.path.join(ROOT, "include", "svx.h")).read()
    declared = set(re.findall(r"\b(svx_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), (declared, set(_lib.SYMBOLS))
    exp_header = open(os.path.join(ROOT, "include", "svx_experimental.h")).read()
    exp_body = re.sub(r"/\*.*?\*/", " ", exp_header, flags=re.S)
    experimental = set(re.findall(r"\b(svx_[a-z_0-9]+)\s*\(", exp_body))
    assert experimental == set(_lib.EXPERIMENTAL) and not experimental & declared
    lib = _lib.load()
    for name in declared | experimental:
        assert getattr(lib, name) is not None
    assert lib.svx_version() == 420
    assert lib.svx_strerror(0) == b"ok" and b"capacity" in lib.svx_strerror(-2)
    assert lib.svx_cigar_scan_ws_bytes(0, 0) >= 0 and lib.svx_cigar_scan_ws_bytes(10_000_000, 0) > 40_000_000
    assert lib.svx_cigar_scan_ws_bytes(1000, 1 << 30) - lib.svx_cigar_scan_ws_bytes(1000, 0) == (1 << 30) // 32 + (1 << 30) // 128      # frame records + the map of the frames launch


def test_device_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from svision_amd import kernels
    with pytest.raises(_lib.SvxError):
        kernels.rasterize(torch.zeros((1, 12), dtype=torch.int32))


def test_scan_flags_and_argument_counts_follow_the_header():
    """The ctypes signatures carry as many arguments as the header's declarations (a call with one too few reads garbage), and the
    shape flags kernels.cigar_scan passes are the header's SVX_SCAN_* values."""
    header = open(os.path.join(ROOT, "include", "svx.h")).read()
    flags = {k: int(v) for k, v in re.findall(r"#define\s+(SVX_SCAN_(?:LANES4|LANES8|SHARED|UNSHARED))\s+(\d+)u", header)}
    assert flags == {"SVX_SCAN_LANES4": 1, "SVX_SCAN_LANES8": 2, "SVX_SCAN_SHARED": 4, "SVX_SCAN_UNSHARED": 8}
    src = open(os.path.join(ROOT, "svision_amd", "kernels.py")).read()
    assert '"groups4": 1 | 8, "groups8": 2 | 8, "groups4s": 1 | 4, "groups8s": 2 | 4' in src
    text = re.sub(r"/\*.*?\*/", " ", header + open(os.path.join(ROOT, "include", "svx_experimental.h")).read(), flags=re.S)
    for name, (_res, args) in list(_lib.SYMBOLS.items()) + list(_lib.EXPERIMENTAL.items()):
        m = re.search(r"\b%s\s*\(([^;{]*?)\)\s*;" % name, text, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(args), (name, n, len(args))
