"""Device ops of the hot path: thin wrappers that hand torch-owned device
buffers to the C ABI of libsvx.so on the current HIP stream.

PyTorch is used here only for device memory and streams.
"""
import ctypes

import numpy as np
import torch

from . import _lib

GAP_DTYPE = np.dtype([("aln", "<u4"), ("op", "<u4"), ("read_pos", "<i4"),
                      ("ref_pos", "<i4"), ("len", "<i4"), ("kind", "<u4")])
MEAN = (104.0, 117.0, 124.0)   # reference src/network/create_batch.py:13


# ---- layout plumbing (torch ops, once per model / in tests; include/svx.h describes the layouts) ----
def to_c8(x):
    """NCHW [n,C,H,W] -> C8 [n,C/8,H,W,8] (contiguous)."""
    n, c, h, w = x.shape
    if c % 8:
        raise _lib.SvxError("the C8 layout needs a channel count that is a multiple of 8")
    return x.reshape(n, c // 8, 8, h, w).permute(0, 1, 3, 4, 2).contiguous()


def from_c8(x):
    """C8 [n,C/8,H,W,8] -> NCHW [n,C,H,W] (contiguous)."""
    n, o, h, w, e = x.shape
    return x.permute(0, 1, 4, 2, 3).reshape(n, o * e, h, w).contiguous()


def pack_conv_weights(w_hwio):
    """Checkpoint conv weights HWIO [k,k,cin_g,cout] -> [k,k,cin_g/8,cout,8] (svx_conv2d_same's d_w_packed)."""
    k, k2, cin_g, cout = w_hwio.shape
    if cin_g % 8:
        raise _lib.SvxError("cin/groups must be a multiple of 8")
    return w_hwio.reshape(k, k2, cin_g // 8, 8, cout).permute(0, 1, 2, 4, 3).contiguous()


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_cuda(t, name):
    if not t.is_cuda:
        raise _lib.SvxError(f"{name} must be a device tensor (got {t.device}); the hot path has no CPU fallback")
    if not t.is_contiguous():
        raise _lib.SvxError(f"{name} must be contiguous")


def rasterize(records, layout="NCHW", mean=MEAN, out=None):
    """records: int32 device tensor [n,12] -> float32 [n,3,227,227] (NCHW) or
    [n,227,227,3] (NHWC), mean-subtracted.  See include/svx.h svx_rasterize."""
    lib = _lib.load()
    _require_cuda(records, "records")
    if records.dtype != torch.int32 or records.dim() != 2 or records.shape[1] != 12:
        raise _lib.SvxError("records must be int32 [n,12]")
    n = records.shape[0]
    lay = {"NHWC": _lib.LAYOUT_NHWC, "NCHW": _lib.LAYOUT_NCHW}[layout]
    shape = (n, _lib.IMG, _lib.IMG, 3) if lay == _lib.LAYOUT_NHWC else (n, 3, _lib.IMG, _lib.IMG)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=records.device)
    else:
        _require_cuda(out, "out")
        if out.dtype != torch.float32 or out.numel() != n * 3 * _lib.IMG * _lib.IMG:
            raise _lib.SvxError("out has the wrong dtype/size")
    m = (ctypes.c_float * 3)(*mean)
    rc = lib.svx_rasterize(records.data_ptr(), n, out.data_ptr(), lay, m, _stream_ptr(records.device))
    _lib.check(rc, "svx_rasterize")
    return out


SCAN_FAILED = 0xFFFFFFFF            # SVX_SCAN_FAILED (include/svx.h): d_gap_off[n_aln] when the offsets pass timed out


def check_scan_total(total):
    if total == SCAN_FAILED:
        raise _lib.SvxError("svx_cigar_scan: SVX_SCAN_FAILED (the offsets pass gave up waiting for a tile in front of it, or the CIGAR array holds more words than the caller said)")
    return total


class CigarScanResult:
    """Device-side result of :func:`cigar_scan`."""

    def __init__(self, gaps, gap_off, stats, n_aln, cap):
        self.gaps, self.gap_off, self.stats, self.n_aln, self.cap = gaps, gap_off, stats, n_aln, cap

    def total(self):
        return check_scan_total(int(self.gap_off[self.n_aln].item()) & 0xFFFFFFFF)

    def to_host(self):
        """(gaps structured array sorted by (aln, op), gap_off uint32[n+1], stats int32[n,4])."""
        off = self.gap_off.cpu().numpy().view(np.uint32)
        total = check_scan_total(int(off[self.n_aln]))
        if total > self.cap:
            raise _lib.SvxError(f"cigar_scan: {total} long gaps exceed the capacity {self.cap}")
        raw = self.gaps[: total * 6].cpu().numpy()
        gaps = raw.view(GAP_DTYPE) if total else np.empty(0, GAP_DTYPE)
        return gaps, off, self.stats.cpu().numpy()


FLAT_SCAN_FROM = None               # mean CIGAR words per alignment from which svx_cigar_scan_flat is picked: None = never -- measured (round 5,
                                    # profiles/r05_bench_cigar.json): 600-800 us on the ONT-shaped launch against 520 us of the three-kernel form (a dozen dependent round trips per tile)


def cigar_scan(cigar, cig_off, ref_start, min_sv, gaps_cap=None, mode=None, n_words=None, span_words=None):
    """cigar: int32/uint32 device tensor of packed BAM CIGAR words; cig_off: int64 [n+1];
    ref_start: int32 [n].  See include/svx.h svx_cigar_scan.  ``mode``: "groups" (svx_cigar_scan: four or eight lanes per alignment
    by the launch's mean words per alignment -- "groups4" / "groups8" / "groups4s" / "groups8s" fix the shape (lanes, s: frames shared by the workgroup) --, three launches), "flat" (svx_cigar_scan_flat: one pass over chunks of the flat word array -- ONT /
    assembly-sized alignments), None: by the mean number of words per alignment (SVX_SCAN_MODE overrides).  ``n_words``: an upper
    bound of the offsets' last entry (default: the size of ``cigar``); ``span_words``: the words between the offsets' first and last
    entry where that is less (a window of a larger array), for the choice of the shape."""
    lib = _lib.load()
    for t, nm in ((cigar, "cigar"), (cig_off, "cig_off"), (ref_start, "ref_start")):
        _require_cuda(t, nm)
    if cigar.dtype not in (torch.int32, torch.uint32) or cig_off.dtype != torch.int64 or ref_start.dtype != torch.int32:
        raise _lib.SvxError("cigar must be (u)int32, cig_off int64, ref_start int32")
    n = ref_start.numel()
    if cig_off.numel() != n + 1:
        raise _lib.SvxError("cig_off must have n_aln + 1 entries")
    dev = cigar.device
    auto = gaps_cap is None
    if auto:
        gaps_cap = max(1024, n // 2)
    gaps = torch.empty(gaps_cap * 6, dtype=torch.int32, device=dev)
    gap_off = torch.empty(n + 1, dtype=torch.int32, device=dev)
    stats = torch.empty((n, 4), dtype=torch.int32, device=dev)
    words = int(cigar.numel()) if n_words is None else int(n_words)
    if mode is None:
        import os
        mode = os.environ.get("SVX_SCAN_MODE") or ("flat" if n and FLAT_SCAN_FROM is not None and words >= FLAT_SCAN_FROM * n else "groups")
    if mode == "flat":
        ws_bytes = int(lib.svx_cigar_scan_flat_ws_bytes(words))
        ws = torch.empty(max(8, ws_bytes), dtype=torch.uint8, device=dev)
        rc = lib.svx_cigar_scan_flat(cigar.data_ptr(), cig_off.data_ptr(), ref_start.data_ptr(), n, words, int(min_sv),
                                     gaps.data_ptr(), gaps_cap, gap_off.data_ptr(), stats.data_ptr(), ws.data_ptr(), int(ws.numel()),
                                     _stream_ptr(dev))
        _lib.check(rc, "svx_cigar_scan_flat")
    else:
        ws = torch.empty(max(16, lib.svx_cigar_scan_ws_bytes(n, words)), dtype=torch.uint8, device=dev)
        flags = {"groups": 0, "groups4": 1 | 8, "groups8": 2 | 8, "groups4s": 1 | 4, "groups8s": 2 | 4}[mode]      # SVX_SCAN_LANES4/8 | SVX_SCAN_(UN)SHARED: the count pass's shape, never a result
        if mode == "groups" and span_words is not None and n:        # a window of a larger array: the library's own rule (svx_cigar.hip SHORT_MEAN / SHARE_MEAN) on the window's words
            flags = (1 if span_words <= 256 * n else 2) | (4 if span_words > 1024 * n else 8)
        rc = lib.svx_cigar_scan(cigar.data_ptr(), cig_off.data_ptr(), ref_start.data_ptr(), n, words, int(min_sv),
                                gaps.data_ptr(), gaps_cap, gap_off.data_ptr(), stats.data_ptr(), ws.data_ptr(), int(ws.numel()), flags,
                                _stream_ptr(dev))
        _lib.check(rc, "svx_cigar_scan")
    res = CigarScanResult(gaps, gap_off, stats, n, gaps_cap)
    if auto and res.total() > gaps_cap:       # d_gap_off[n] holds the full count: rerun with the exact capacity
        return cigar_scan(cigar, cig_off, ref_start, min_sv, gaps_cap=res.total(), mode=mode, n_words=n_words, span_words=span_words)
    return res


def bias_relu_pool_lrn(x, bias, lrn=True, radius=2, alpha=2e-05, beta=0.75, k=1.0, active_rows=None, background=None):
    """x: float32 C8 [n,C/8,H,W,8] raw conv output -> relu(x+bias) -> max-pool 3x3/2 -> (LRN) as one kernel, C8 out.
    ``active_rows`` (int32 [n,H] row masks) + ``background`` (C8 [C/8,H,W,8]): pixels whose bit is clear are read from the
    background instead of ``x`` (an active-set convolution that did not write them).  See include/svx.h svx_bias_relu_pool_lrn."""
    lib = _lib.load()
    _require_cuda(x, "x")
    _require_cuda(bias, "bias")
    if x.dtype != torch.float32 or x.dim() != 5 or x.shape[4] != 8 or bias.dtype != torch.float32 or bias.numel() != x.shape[1] * 8:
        raise _lib.SvxError("x must be float32 C8 [n,C/8,H,W,8] and bias float32 [C]")
    n, o, h, w, _e = x.shape
    y = torch.empty((n, o, (h - 3) // 2 + 1, (w - 3) // 2 + 1, 8), dtype=torch.float32, device=x.device)
    if (active_rows is None) != (background is None):
        raise _lib.SvxError("active_rows and background go together")
    if active_rows is not None:
        _require_cuda(active_rows, "active_rows")
        _require_cuda(background, "background")
        if active_rows.dtype != torch.int32 or tuple(active_rows.shape) != (n, h) or background.dtype != torch.float32 \
                or tuple(background.shape)[-4:] != (o, h, w, 8) or background.numel() != o * h * w * 8 \
                or not background.is_contiguous() or not active_rows.is_contiguous():
            raise _lib.SvxError("active_rows must be int32 [n,H] and background float32 C8 [C/8,H,W,8]")
    rc = lib.svx_bias_relu_pool_lrn(x.data_ptr(), bias.data_ptr(), y.data_ptr(), n, o * 8, h, w, 1 if lrn else 0, radius,
                                    alpha, beta, k, active_rows.data_ptr() if active_rows is not None else None,
                                    background.data_ptr() if background is not None else None, _stream_ptr(x.device))
    _lib.check(rc, "svx_bias_relu_pool_lrn")
    return y


def encode_conv1(records, w1_hwio, base, lrn=True, radius=2, alpha=2e-05, beta=0.75, k=1.0, touched=False):
    """records int32 [n,12] -> float32 C8 [n,12,27,27,8]: rasterise + conv1 + relu + pool1 + norm1 in one
    kernel, exploiting the sparsity of the similarity image.  See include/svx.h svx_encode_conv1.
    ``touched=True``: also return the int32 [n,27] row masks of the pooled pixels with a set tap under them."""
    lib = _lib.load()
    for t, nm in ((records, "records"), (w1_hwio, "w1"), (base, "base")):
        _require_cuda(t, nm)
    if records.dtype != torch.int32 or records.dim() != 2 or records.shape[1] != 12:
        raise _lib.SvxError("records must be int32 [n,12]")
    if tuple(w1_hwio.shape) != (11, 11, 3, 96) or w1_hwio.dtype != torch.float32 or base.numel() != 96:
        raise _lib.SvxError("w1 must be float32 HWIO [11,11,3,96] and base float32 [96]")
    n = records.shape[0]
    y = torch.empty((n, 12, 27, 27, 8), dtype=torch.float32, device=records.device)
    mask = torch.empty((n, 27), dtype=torch.int32, device=records.device) if touched else None
    rc = lib.svx_encode_conv1(records.data_ptr(), n, w1_hwio.data_ptr(), base.data_ptr(), y.data_ptr(), 1 if lrn else 0,
                              radius, alpha, beta, k, mask.data_ptr() if touched else None, _stream_ptr(records.device))
    _lib.check(rc, "svx_encode_conv1")
    return (y, mask) if touched else y


def alexnet_active_sets(touched, totals=None, rows=False):
    """touched int32 [n,27] (encode_conv1) -> (list2 [n*729], list3, list4, list5 [n*169], counts [4]) int32 device
    tensors: the output pixels of conv2..conv5 that can differ from the response to an empty image.
    ``totals``: optional int64 device tensor [5] the launch adds its executed pixel counts (conv2..conv5) and image count to.
    See include/svx.h svx_alexnet_active_sets."""
    lib = _lib.load()
    _require_cuda(touched, "touched")
    if totals is not None and (not totals.is_cuda or totals.dtype != torch.int64 or totals.numel() != 5):
        raise _lib.SvxError("totals must be an int64 device tensor [5]")
    if touched.dtype != torch.int32 or touched.dim() != 2 or touched.shape[1] != 27:
        raise _lib.SvxError("touched must be int32 [n,27]")
    n, dev = touched.shape[0], touched.device
    lists = [torch.empty(n * hw, dtype=torch.int32, device=dev) for hw in (729, 169, 169, 169)]
    counts = torch.empty(4, dtype=torch.int32, device=dev)
    ws = torch.empty(max(n, 1) * 4, dtype=torch.int32, device=dev)
    active2 = torch.empty((n, 27), dtype=torch.int32, device=dev) if rows else None
    rc = lib.svx_alexnet_active_sets(touched.data_ptr(), n, lists[0].data_ptr(), lists[1].data_ptr(), lists[2].data_ptr(),
                                     lists[3].data_ptr(), counts.data_ptr(), ws.data_ptr(),
                                     totals.data_ptr() if totals is not None else None,
                                     active2.data_ptr() if rows else None, _stream_ptr(dev))
    _lib.check(rc, "svx_alexnet_active_sets")
    if rows:                                                  # + int32 [n,27] row masks of conv2's active pixels
        return lists[0], lists[1], lists[2], lists[3], counts, active2
    return lists[0], lists[1], lists[2], lists[3], counts


def conv2d_same(x, w_packed, bias=None, groups=1, relu=False, pixels=None, pixel_count=None, out=None, background=None):
    """x float32 C8 [n,Cin/8,H,W,8], w_packed float32 [k,k,Cin/groups/8,Cout,8] (:func:`pack_conv_weights`) -> C8
    [n,Cout/8,H,W,8]: stride-1 SAME convolution on the fp32 matrix cores, optional fused bias + ReLU.
    ``pixels`` / ``pixel_count`` (device int32 permutation of the pixel ids and the number of leading active entries,
    a one-element view): compute only the active output pixels; the others receive ``background`` (C8 [Cout/8,H,W,8]) or,
    without it, keep what ``out`` holds.  See include/svx.h svx_conv2d_same."""
    lib = _lib.load()
    _require_cuda(x, "x")
    _require_cuda(w_packed, "w_packed")
    if x.dtype != torch.float32 or x.dim() != 5 or x.shape[4] != 8 or w_packed.dtype != torch.float32 or w_packed.dim() != 5 or w_packed.shape[4] != 8:
        raise _lib.SvxError("x must be float32 C8 [n,C/8,H,W,8] and w float32 packed [k,k,cin_g/8,cout,8]")
    n, cin8, h, w, _e = x.shape
    cin = cin8 * 8
    k, k2, cin_g8, cout, _e2 = w_packed.shape
    if k != k2 or cin_g8 * 8 * groups != cin:
        raise _lib.SvxError("weight shape %s does not match input %s with %d groups" % (tuple(w_packed.shape), tuple(x.shape), groups))
    if bias is not None:
        _require_cuda(bias, "bias")
    if (pixels is None) != (pixel_count is None) or (pixels is not None and out is None and background is None):
        raise _lib.SvxError("pixels and pixel_count go together, with out or background")
    if background is not None and (pixels is None or tuple(background.shape[-4:]) != (cout // 8, h, w, 8) or not background.is_contiguous()):
        raise _lib.SvxError("background must be a contiguous float32 C8 [Cout/8,H,W,8] tensor and needs pixels")
    y = out if out is not None else torch.empty((n, cout // 8, h, w, 8), dtype=torch.float32, device=x.device)
    if tuple(y.shape) != (n, cout // 8, h, w, 8) or y.dtype != torch.float32 or not y.is_contiguous():
        raise _lib.SvxError("out must be a contiguous float32 C8 [n,Cout/8,H,W,8] tensor")
    rc = lib.svx_conv2d_same(x.data_ptr(), w_packed.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                             n, cin, cout, h, w, k, groups, 1 if relu else 0,
                             pixels.data_ptr() if pixels is not None else None,
                             pixel_count.data_ptr() if pixel_count is not None else None,
                             background.data_ptr() if background is not None else None, _stream_ptr(x.device))
    _lib.check(rc, "svx_conv2d_same")
    return y


def pack_fc_weights(w_out_in):
    """fc weights [out,in] -> [out/32, in/8, 32, 8] (svx_fc_bias_act's d_w_packed)."""
    n, k = w_out_in.shape
    if n % 32 or k % 8:
        raise _lib.SvxError("fc weights need out % 32 == 0 and in % 8 == 0")
    return w_out_in.reshape(n // 32, 32, k // 8, 8).permute(0, 2, 1, 3).contiguous()


def fc_bias_act(x, w_packed, bias, relu=True, out=None, ws=None):
    """x float32 [m,k], w_packed [n/32,k/8,32,8] (:func:`pack_fc_weights`), bias [n] -> act(x @ W^T + bias) [m,n] on the
    fp32 matrix cores (split-K wave tiles + ordered reduction).  See include/svx.h svx_fc_bias_act."""
    lib = _lib.load()
    for t, nm in ((x, "x"), (w_packed, "w_packed"), (bias, "bias")):
        _require_cuda(t, nm)
    if x.dtype != torch.float32 or x.dim() != 2 or w_packed.dim() != 4 or w_packed.shape[2:] != (32, 8) or w_packed.shape[1] * 8 != x.shape[1]:
        raise _lib.SvxError("x must be float32 [m,k] and w packed [n/32,k/8,32,8]")
    m, k = x.shape
    n = w_packed.shape[0] * 32
    if bias.numel() != n:
        raise _lib.SvxError("bias must have n entries")
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    need = lib.svx_fc_ws_bytes(m, n, k)
    if ws is None or ws.numel() * ws.element_size() < need:
        ws = torch.empty(max(need, 16) // 4, dtype=torch.float32, device=x.device)
    rc = lib.svx_fc_bias_act(x.data_ptr(), w_packed.data_ptr(), bias.data_ptr(), out.data_ptr(), ws.data_ptr(), m, n, k, 1 if relu else 0,
                             _stream_ptr(x.device))
    _lib.check(rc, "svx_fc_bias_act")
    return out


def fc8_softmax(x, w_out_in, bias, out=None):
    """x float32 [n,4096], w float32 [5,4096], bias [5] -> packed float32 [n,12] = softmax[5], class, logits[5], 0.
    See include/svx.h svx_fc8_softmax."""
    lib = _lib.load()
    for t, nm in ((x, "x"), (w_out_in, "w"), (bias, "bias")):
        _require_cuda(t, nm)
    if x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != 4096 or tuple(w_out_in.shape) != (5, 4096):
        raise _lib.SvxError("x must be float32 [n,4096] and w float32 [5,4096]")
    n = x.shape[0]
    if out is None:
        out = torch.empty((n, 12), dtype=torch.float32, device=x.device)
    rc = lib.svx_fc8_softmax(x.data_ptr(), w_out_in.data_ptr(), bias.data_ptr(), out.data_ptr(), n, _stream_ptr(x.device))
    _lib.check(rc, "svx_fc8_softmax")
    return out


def span_position_distance(starts, ends, part_off, normalizer=1000.0):
    """Condensed span_position_distance matrices of several partitions in one launch.
    starts / ends: float64 device tensors [N] (partitions concatenated); part_off: int sequence [P+1].
    -> (float64 device tensor [sum n_p (n_p - 1) / 2] in scipy pdist order, out_off numpy uint64 [P+1]).
    See include/svx.h svx_span_position_distance."""
    lib = _lib.load()
    _require_cuda(starts, "starts")
    _require_cuda(ends, "ends")
    if starts.dtype != torch.float64 or ends.dtype != torch.float64 or starts.numel() != ends.numel():
        raise _lib.SvxError("starts / ends must be float64 tensors of one length")
    part = np.ascontiguousarray(part_off, np.uint64)
    if part.ndim != 1 or part.size < 1 or int(part[-1]) != starts.numel() or np.any(np.diff(part.astype(np.int64)) < 0):
        raise _lib.SvxError("part_off must be ascending and end at the number of signatures")
    sizes = np.diff(part.astype(np.int64))
    out_off = np.zeros(part.size, np.uint64)
    out_off[1:] = np.cumsum(sizes * (sizes - 1) // 2)
    total = int(out_off[-1])
    out = torch.empty(total, dtype=torch.float64, device=starts.device)
    if total:
        d_part = torch.from_numpy(part.view(np.int64)).to(starts.device)
        d_off = torch.from_numpy(out_off.view(np.int64)).to(starts.device)
        rc = lib.svx_span_position_distance(starts.data_ptr(), ends.data_ptr(), d_part.data_ptr(), part.size - 1, d_off.data_ptr(), total,
                                            float(normalizer), out.data_ptr(), _stream_ptr(starts.device))
        _lib.check(rc, "svx_span_position_distance")
    return out, out_off


HASH_MAX_X = 2048
_BASE_CODE = np.full(256, 255, np.uint8)
for _i, _c in enumerate(b"ACGTNacgtnRYKMS"):
    _BASE_CODE[_c] = _i
HASH_JOB_DTYPE = np.dtype([("x_off", "<u8"), ("y_off", "<u8"), ("x_len", "<u4"), ("y_len", "<u4"), ("table_off", "<u8"),
                           ("table_slots", "<u4"), ("hit_cap", "<u4"), ("hit_off", "<u8")])


def pack_bases(seq):
    """str / bytes over ACGTN, acgtn (soft-masked references) and RYKMS -> uint8 symbols 0..14, or None when another
    character occurs (those jobs take the host path, whose k-mers are the raw strings)."""
    raw = np.frombuffer(seq.encode() if isinstance(seq, str) else bytes(seq), np.uint8)
    codes = _BASE_CODE[raw]
    return None if (codes == 255).any() else codes


def hash_seeds(jobs, k, window, device):
    """jobs: list of (x_codes, y_codes) uint8 arrays (pack_bases) -> list of (hits_a, hits_b) int32 arrays [n,4]
    = {y position, x position or position in x's reverse complement, match length, forward}, or None for a job whose
    hit lists overflowed.  One launch for the whole batch.  See include/svx.h svx_hash_seeds."""
    lib = _lib.load()
    if not jobs:
        return []
    dev = torch.device(device)
    desc = np.zeros(len(jobs), HASH_JOB_DTYPE)
    parts, off, slots, hit_off = [], 0, 0, 0
    for j, (x, y) in enumerate(jobs):
        if len(x) > HASH_MAX_X:
            raise _lib.SvxError("hash_seeds: piece longer than %d bases" % HASH_MAX_X)
        t = 1
        while t < 8 * max(len(y), 1):
            t <<= 1
        cap = 4 * len(y) + 64
        desc[j] = (off, off + len(x), len(x), len(y), slots, t, cap, hit_off)
        parts += [x, y]
        off += len(x) + len(y)
        slots += t
        hit_off += 2 * cap
    d_bases = torch.from_numpy(np.concatenate(parts + [np.zeros(16, np.uint8)])).to(dev)
    d_jobs = torch.from_numpy(desc.view(np.uint8).copy()).to(dev)
    d_table = torch.empty(slots * 2, dtype=torch.int64, device=dev)
    d_hits = torch.empty(hit_off * 4, dtype=torch.int32, device=dev)
    d_counts = torch.empty(2 * len(jobs), dtype=torch.int32, device=dev)
    rc = lib.svx_hash_seeds(d_bases.data_ptr(), d_jobs.data_ptr(), len(jobs), d_table.data_ptr(), d_hits.data_ptr(), d_counts.data_ptr(),
                            int(k), int(window), HASH_MAX_X, _stream_ptr(dev))
    _lib.check(rc, "svx_hash_seeds")
    counts = d_counts.cpu().numpy().astype(np.int64)
    hits = d_hits.cpu().numpy().reshape(-1, 4)
    out = []
    for j in range(len(jobs)):
        na, nb, cap, ho = int(counts[2 * j]), int(counts[2 * j + 1]), int(desc[j]["hit_cap"]), int(desc[j]["hit_off"])
        out.append(None if na > cap or nb > cap else (hits[ho:ho + na], hits[ho + cap:ho + cap + nb]))
    return out


def bgzf_block_table(raw):
    """BGZF bytes (host, uint8 array: a run of whole blocks) -> (src_off uint64 [n], src_len uint32 [n], isize uint32 [n],
    block_off uint64 [n]): where every block's DEFLATE payload lies, its inflated size, and the block's own offset."""
    raw = np.ascontiguousarray(raw, np.uint8)
    src_off, src_len, isize, block_off = [], [], [], []
    p, n = 0, raw.size
    while p + 18 <= n:
        if not (raw[p] == 0x1f and raw[p + 1] == 0x8b and raw[p + 2] == 8 and raw[p + 3] & 4):
            raise _lib.SvxError("not a BGZF block at offset %d" % p)
        xlen = int(raw[p + 10]) | int(raw[p + 11]) << 8
        q, bsize = p + 12, None
        while q + 4 <= p + 12 + xlen:
            slen = int(raw[q + 2]) | int(raw[q + 3]) << 8
            if raw[q] == 66 and raw[q + 1] == 67:
                bsize = int(raw[q + 4]) | int(raw[q + 5]) << 8
            q += 4 + slen
        if bsize is None or p + bsize + 1 > n:
            break
        end = p + bsize + 1
        block_off.append(p)
        src_off.append(p + 12 + xlen)
        src_len.append(end - 8 - (p + 12 + xlen))
        isize.append(int.from_bytes(raw[end - 4:end].tobytes(), "little"))
        p = end
    return (np.asarray(src_off, np.uint64), np.asarray(src_len, np.uint32), np.asarray(isize, np.uint32), np.asarray(block_off, np.uint64))


WAVE_KERNEL_BELOW = 20_000          # (lane-per-block era) blocks per launch under which the wave-per-block kernel beat the lane-per-block one


def inflate_variant_for(n_blocks):
    """Which inflate implementation a launch of ``n_blocks`` blocks takes: "fast" -- the two-kernel form (svx_inflate2.hip:
    parallel Huffman decoding per block + LZ copies; 0.5 + 0.65 ms per 1,000 blocks, proportional to the launch) -- unless
    SVX_INFLATE_VARIANT names another (lds | private | wave | lane | fast; "auto": the round-3 choice between the lane- and
    the wave-per-block kernel by launch size)."""
    import os
    variant = os.environ.get("SVX_INFLATE_VARIANT", "fast")
    if variant == "auto":
        return "wave" if n_blocks < WAVE_KERNEL_BELOW else "lane"
    return variant


def inflate_workspace(lib, variant, inflated_bytes, n_blocks, device):
    """The workspace tensor a launch of ``variant`` needs (None: none), allocated on the caller's current stream."""
    if not variant.startswith("fast"):
        return None
    return torch.empty(int(lib.svx_bgzf_inflate_fast_ws_bytes(int(inflated_bytes), int(n_blocks))), dtype=torch.uint8, device=device)


def launch_inflate(lib, variant, comp_ptr, src_ptr, len_ptr, dst_ptr, n_blocks, out_ptr, status_ptr, inflated_bytes, device, ws=None, tokens_stream=None):
    """Enqueue one inflate launch on the current stream of ``device``.  ``ws``: the "fast" form's workspace
    (:func:`inflate_workspace`); None: taken from the caching allocator here -- stream-ordered, so it may die with this call.
    ``tokens_stream``: the "fast" form's first kernel goes there (svx_bgzf_inflate_fast_on), the rest stays on the current stream."""
    st = _stream_ptr(device)
    lz = None
    if variant in ("fast-lane", "fast-wave"):                 # the "fast" form with its LZ kernel by name (tests, measurements)
        lz, variant = variant[5:], "fast"
    if variant == "fast":
        d_ws = ws if ws is not None else inflate_workspace(lib, variant, inflated_bytes, n_blocks, device)
        ws_bytes = int(d_ws.numel())
        rc = _launch_fast(lib, comp_ptr, src_ptr, len_ptr, dst_ptr, n_blocks, inflated_bytes, out_ptr, status_ptr, d_ws, ws_bytes, tokens_stream, st, lz)
        _lib.check(rc, "svx_bgzf_inflate (%s)" % variant)
        return
    fn = {"lds": lib.svx_bgzf_inflate_lds, "private": lib.svx_bgzf_inflate_private, "wave": lib.svx_bgzf_inflate_wave,
          "lane": lib.svx_bgzf_inflate}[variant]
    rc = fn(comp_ptr, src_ptr, len_ptr, dst_ptr, n_blocks, out_ptr, status_ptr, st)
    _lib.check(rc, "svx_bgzf_inflate (%s)" % variant)


def _launch_fast(lib, comp_ptr, src_ptr, len_ptr, dst_ptr, n_blocks, inflated_bytes, out_ptr, status_ptr, d_ws, ws_bytes, tokens_stream, st, lz=None):
    if lz is not None:                                        # the LZ kernel by name: an argument of the experimental entry point (svx_experimental.h)
        tok = ctypes.c_void_p(tokens_stream.cuda_stream) if tokens_stream is not None else st
        return lib.svx_bgzf_inflate_fast_lz(comp_ptr, src_ptr, len_ptr, dst_ptr, n_blocks, int(inflated_bytes), out_ptr, status_ptr, d_ws.data_ptr(), ws_bytes,
                                            {"lane": 1, "wave": 2}[lz], tok, st)
    if tokens_stream is not None:
        return lib.svx_bgzf_inflate_fast_on(comp_ptr, src_ptr, len_ptr, dst_ptr, n_blocks, int(inflated_bytes), out_ptr, status_ptr, d_ws.data_ptr(), ws_bytes,
                                            ctypes.c_void_p(tokens_stream.cuda_stream), st)
    return lib.svx_bgzf_inflate_fast(comp_ptr, src_ptr, len_ptr, dst_ptr, n_blocks, int(inflated_bytes), out_ptr, status_ptr, d_ws.data_ptr(), ws_bytes, st)


INFLATE_BAD_CRC = 9                 # SVX_INFLATE_BAD_CRC: the block inflated, but not to the bytes its footer's CRC32 was taken of


def bgzf_crc_wanted():
    """BGZF footers are verified (htslib does, behind pysam's fetch) unless SVX_BGZF_CRC=0."""
    import os
    return os.environ.get("SVX_BGZF_CRC", "1") != "0"


def bgzf_inflate(d_comp, src_off, src_len, isize, wave=None, crc=None):
    """d_comp: uint8 device tensor holding the compressed bytes (16-byte aligned, padded to a multiple of 16); src_off / src_len / isize: host
    arrays of :func:`bgzf_block_table` (or the native reader's).  -> (uint8 device tensor with the inflated stream,
    int32 device tensor [n] status: 0 = ok, 9 = CRC32 mismatch).  ``crc``: verify the blocks' footers (svx_bgzf_crc32; default:
    yes unless SVX_BGZF_CRC=0).  See include/svx.h svx_bgzf_inflate."""
    lib = _lib.load()
    _require_cuda(d_comp, "d_comp")
    if d_comp.dtype != torch.uint8:
        raise _lib.SvxError("d_comp must be a uint8 tensor")
    n = int(len(src_off))
    dev = d_comp.device
    dst = np.zeros(n + 1, np.uint64)
    dst[1:] = np.cumsum(np.asarray(isize, np.uint64))
    total = int(dst[-1])
    d_out = torch.empty(max(total, 4), dtype=torch.uint8, device=dev)
    d_status = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    if n:
        d_src = torch.from_numpy(np.ascontiguousarray(src_off, np.uint64).view(np.int64)).to(dev)
        d_len = torch.from_numpy(np.ascontiguousarray(src_len, np.uint32).view(np.int32)).to(dev)
        d_dst = torch.from_numpy(dst.view(np.int64)).to(dev)
        # the implementations of one contract: wave None = the default for this launch size, True / False the wave- / the
        # lane-per-block kernel, a string one implementation by name ("lds", "private", "wave", "lane", "fast")
        variant = inflate_variant_for(n) if wave is None else wave if isinstance(wave, str) else ("wave" if wave else "lane")
        launch_inflate(lib, variant, d_comp.data_ptr(), d_src.data_ptr(), d_len.data_ptr(), d_dst.data_ptr(), n, d_out.data_ptr(),
                       d_status.data_ptr(), total, dev)
        if bgzf_crc_wanted() if crc is None else crc:
            # the blocks' footers (CRC32 of the inflated bytes) checked on the device: status 9 where one differs
            rc = lib.svx_bgzf_crc32(d_out.data_ptr(), d_dst.data_ptr(), d_comp.data_ptr(), d_src.data_ptr(), d_len.data_ptr(), n,
                                    d_status.data_ptr(), _stream_ptr(dev))
            _lib.check(rc, "svx_bgzf_crc32")
    return d_out[:total], d_status[:n]
