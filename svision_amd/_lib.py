"""ctypes binding of libsvx.so (C ABI declared in include/svx.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C svision_amd/csrc``.  Loading fails loudly when it is missing: the
product has no CPU or PyTorch fallback for these ops.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsvx.so")

SVX_OK = 0
SVX_EINVAL = -1
SVX_ECAPACITY = -2
SVX_ELAUNCH = -3
LAYOUT_NHWC = 0
LAYOUT_NCHW = 1
IMG = 227
GAP_INS = 1
GAP_DEL = 2

# every symbol include/svx.h declares: name -> (restype, argtypes)
_vp, _u32, _i32, _u64, _sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_uint64, ctypes.c_size_t
SYMBOLS = {
    "svx_version": (ctypes.c_int, []),
    "svx_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "svx_crc32c": (_u32, [_vp, _sz]),
    "svx_cigar_scan_ws_bytes": (_sz, [_u32, _u64]),
    "svx_cigar_scan": (ctypes.c_int, [_vp, _vp, _vp, _u32, _u64, _i32, _vp, _u64, _vp, _vp, _vp, _u64, _u32, _vp]),
    "svx_rasterize": (ctypes.c_int, [_vp, _u32, _vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float), _vp]),
    "svx_encode_conv1": (ctypes.c_int, [_vp, _u32, _vp, _vp, _vp, ctypes.c_int, _u32, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_float, _vp, _vp]),
    "svx_alexnet_active_sets": (ctypes.c_int, [_vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "svx_conv2d_same": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, ctypes.c_int, _vp, _vp, _vp, _vp]),
    "svx_fc_ws_bytes": (_sz, [_u32, _u32, _u32]),
    "svx_fc_bias_act": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, ctypes.c_int, _vp]),
    "svx_fc8_softmax": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32, _vp]),
    "svx_bias_relu_pool_lrn": (ctypes.c_int, [_vp, _vp, _vp, _u32, _u32, _u32, _u32, ctypes.c_int, _u32,
                                              ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp]),
    "svx_span_position_distance": (ctypes.c_int, [_vp, _vp, _vp, _u32, _vp, _u64, ctypes.c_double, _vp, _vp]),
    "svx_hash_seeds": (ctypes.c_int, [_vp, _vp, _u32, _vp, _vp, _vp, _u32, _u32, _u32, _vp]),
    "svx_bam_open": (_vp, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]),
    "svx_bam_open_range": (_vp, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, _u64, _u64]),
    "svx_bam_error": (ctypes.c_char_p, []),
    "svx_bam_sizes": (None, [_vp, _vp]),
    "svx_bam_export": (None, [_vp, ctypes.c_int] + [_vp] * 13),
    "svx_bam_seq": (_vp, [_vp]),
    "svx_bam_close": (None, [_vp]),
    "svx_bgzf_inflate_fast_ws_bytes": (_sz, [_u64, _u32]),
    "svx_bgzf_inflate_fast": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32, _u64, _vp, _vp, _vp, _u64, _vp]),
    "svx_bgzf_inflate_fast_on": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32, _u64, _vp, _vp, _vp, _u64, _vp, _vp]),
    "svx_bgzf_crc32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp]),
    "svx_bam_walk_count": (ctypes.c_int, [_vp, _vp, _u32, _vp, _vp]),
    "svx_bam_walk_extract": (ctypes.c_int, [_vp, _vp, _u32, _vp] + [_vp] * 9 + [_u32, _vp]),
    "svx_read_range": (ctypes.c_int, [ctypes.c_char_p, _u64, _u64, _vp, ctypes.c_int]),
    "svx_bgzf_index": (ctypes.c_int64, [_vp, _u64, _u64, _u64, _vp, _vp, _vp, _vp, _vp]),
    "svx_name_ids": (ctypes.c_int64, [_vp, _vp, _u64, _vp, _vp, _vp]),
    "svx_bam_stream_open": (_vp, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int]),
    "svx_bam_stream_next": (_vp, [_vp, _vp]),
    "svx_bam_stream_close": (None, [_vp]),
}
# every symbol include/svx_experimental.h declares: implementations the default path never calls (tests, A/B measurements)
EXPERIMENTAL = {
    "svx_cigar_scan_flat_ws_bytes": (_sz, [_u64]),
    "svx_cigar_scan_flat": (ctypes.c_int, [_vp, _vp, _vp, _u32, _u64, _i32, _vp, _u64, _vp, _vp, _vp, _u64, _vp]),
    "svx_bgzf_inflate": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp]),
    "svx_bgzf_inflate_wave": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp]),
    "svx_bgzf_inflate_lds": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp]),
    "svx_bgzf_inflate_private": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp]),
    "svx_bgzf_inflate_fast_lz": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32, _u64, _vp, _vp, _vp, _u64, ctypes.c_int, _vp, _vp]),
}


class SvxError(RuntimeError):
    pass


class SvxMissing(SvxError):
    """libsvx.so has not been built (as opposed to: built, but wrong)."""


_lib = None


ABI_VERSION = 420                     # SVX_VERSION of include/svx.h this binding was written against


def load():
    """Return the loaded library, binding all prototypes on first use."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SvxMissing(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C svision_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in list(SYMBOLS.items()) + list(EXPERIMENTAL.items()):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.svx_version() != ABI_VERSION:
            raise SvxError(f"{LIB_PATH} reports ABI {lib.svx_version()}, this package binds {ABI_VERSION}: rebuild it (make -C svision_amd/csrc)")
        _lib = lib
    return _lib


def check(code, what):
    if code != SVX_OK:
        msg = load().svx_strerror(code).decode()
        raise SvxError(f"{what}: {msg} ({code})")
