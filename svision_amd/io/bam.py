"""Own BGZF/BAM/FASTA codec (no pysam/htslib in this image or on the GPU box).

Replaces, for the hot path, what the reference obtains from pysam
(SURVEY 8(a')): ``AlignmentFile.fetch`` (run_collection.py:23-26,
classes.py:165-170, genotype.py:22-26), header checks (SVision:141-157),
``FastaFile.fetch`` (analyze_reads.py:42-46).  Records are decoded straight
into a structure-of-arrays :class:`AlignmentTable` whose packed CIGAR words are
uploaded to HBM once and scanned by ``svx_cigar_scan``; per-record Python
objects are never built.

Follows the SAM/BAM specification (SAMv1 section 4): BGZF blocks are gzip members with a
``BC`` extra field; BAM records are little-endian.  CIGARs longer than 65535
ops (``CG:B,I`` tag) are handled by the native decoder and the writer.
"""
import os
import struct
import zlib

import numpy as np

_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
_MAX_BLOCK = 0xFF00

_SEQ_LUT = np.frombuffer(b"=ACMGRSVTWYHKDBN", np.uint8)
_SEQ_CODE = np.zeros(256, np.uint8) + 15
for _i, _c in enumerate(b"=ACMGRSVTWYHKDBN"):
    _SEQ_CODE[_c] = _i
    _SEQ_CODE[ord(chr(_c).lower())] = _i

FLAG_REVERSE = 0x10
FLAG_UNMAPPED = 0x4
FLAG_SECONDARY = 0x100
FLAG_SUPPLEMENTARY = 0x800


def bgzf_decompress(data):
    """Concatenated BGZF blocks -> decompressed bytes."""
    out = []
    pos, n = 0, len(data)
    while pos < n:
        if data[pos:pos + 4] != b"\x1f\x8b\x08\x04":
            raise ValueError("not a BGZF block at offset %d" % pos)
        xlen = struct.unpack_from("<H", data, pos + 10)[0]
        bsize = None
        p, end = pos + 12, pos + 12 + xlen
        while p < end:
            si1, si2, slen = data[p], data[p + 1], struct.unpack_from("<H", data, p + 2)[0]
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", data, p + 4)[0]
            p += 4 + slen
        if bsize is None:
            raise ValueError("BGZF block without BC field")
        cdata = data[pos + 12 + xlen: pos + bsize + 1 - 8]
        out.append(zlib.decompress(cdata, -15))
        pos += bsize + 1
    return b"".join(out)


def bgzf_compress(data, level=1, block_offsets=None):
    """bytes -> BGZF blocks (+ EOF marker); ``block_offsets`` (a list) receives the file offset of every block."""
    out = []
    at = 0
    for i in range(0, len(data), _MAX_BLOCK):
        chunk = data[i:i + _MAX_BLOCK]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        cdata = co.compress(chunk) + co.flush()
        bsize = len(cdata) + 25
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize)
                   + cdata + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
        if block_offsets is not None:
            block_offsets.append(at)
        at += len(out[-1])
    if block_offsets is not None:
        block_offsets.append(at)                            # the EOF block
    out.append(_BGZF_EOF)
    return b"".join(out)


class AlignmentTable:
    """Structure-of-arrays view of a coordinate-sorted set of BAM records.

    tid, pos (int32), flag (uint16), mapq (uint8), l_seq (int32; 0 when SEQ is '*'),
    name_id (int32, first-occurrence order of QNAME), cigar (uint32 packed words),
    cig_off (int64 CSR, n+1).  ``ref_span``/``lead_clip``/``trail_clip`` are filled by
    the device CIGAR scan (:meth:`attach_scan`).
    """

    def __init__(self, references, lengths, tid, pos, flag, mapq, l_seq, name_id, names, cigar, cig_off, header_text="",
                 seq_packed=None, seq_off=None):
        self.references = list(references)
        self.lengths = list(lengths)
        self.tid = np.ascontiguousarray(tid, np.int32)
        self.pos = np.ascontiguousarray(pos, np.int32)
        self.flag = np.ascontiguousarray(flag, np.uint16)
        self.mapq = np.ascontiguousarray(mapq, np.uint8)
        self.l_seq = np.ascontiguousarray(l_seq, np.int32)
        self.name_id = np.ascontiguousarray(name_id, np.int32)
        self.names = names
        self.cigar = np.ascontiguousarray(cigar, np.uint32)
        self.cig_off = np.ascontiguousarray(cig_off, np.int64)
        self.header_text = header_text
        # optional read bases (needed by --hash only): BAM 4-bit codes, two per byte, record i at byte seq_off[i]
        self.seq_packed = seq_packed
        self.seq_off = seq_off
        self.ref_span = None
        self.lead_clip = None
        self.trail_clip = None
        self._tid_bounds = None

    def __len__(self):
        return int(self.tid.size)

    @property
    def sort_order(self):
        for line in self.header_text.split("\n"):
            if line.startswith("@HD"):
                for f in line.split("\t")[1:]:
                    if f.startswith("SO:"):
                        return f[3:]
        return None

    def get_tid(self, name):
        try:
            return self.references.index(name)
        except ValueError:
            return -1

    def query_sequence(self, i):
        """SEQ of record i as stored in the BAM (str), or None when absent ('*') -- pysam's query_sequence."""
        n = int(self.l_seq[i])
        if n == 0 or self.seq_packed is None:
            return None
        o = int(self.seq_off[i])
        raw = np.frombuffer(self.seq_packed, np.uint8, (n + 1) // 2, o)
        codes = np.empty(2 * raw.size, np.uint8)
        codes[0::2] = raw >> 4
        codes[1::2] = raw & 15
        return _SEQ_LUT[codes[:n]].tobytes().decode()

    def subset(self, rows):
        """A new table holding only ``rows`` (ascending record indices); QNAME ids are re-assigned by first occurrence."""
        rows = np.asarray(rows, np.int64)
        n_cig = self.cig_off[rows + 1] - self.cig_off[rows]
        new_off = np.zeros(rows.size + 1, np.int64)
        new_off[1:] = np.cumsum(n_cig)
        idx = np.repeat(self.cig_off[rows] - new_off[:-1], n_cig) + np.arange(int(new_off[-1]))
        uniq, first, inv = np.unique(self.name_id[rows], return_index=True, return_inverse=True)
        order = np.argsort(first, kind="stable")
        rank = np.empty(order.size, np.int64)
        rank[order] = np.arange(order.size)
        return AlignmentTable(self.references, self.lengths, self.tid[rows], self.pos[rows], self.flag[rows], self.mapq[rows],
                              self.l_seq[rows], rank[inv].astype(np.int32), [self.names[int(uniq[j])] for j in order],
                              self.cigar[idx] if idx.size else np.empty(0, np.uint32), new_off, self.header_text,
                              self.seq_packed, None if self.seq_off is None else self.seq_off[rows])

    def ids_of(self, names):
        """name_id values of the given QNAMEs (unknown names are ignored)."""
        if getattr(self, "_name_index", None) is None:
            self._name_index = {nm: i for i, nm in enumerate(self.names)}
        idx = self._name_index
        return np.fromiter((idx[n] for n in names if n in idx), np.int64)

    def attach_scan(self, stats):
        """stats: int32 [n,4] from svx_cigar_scan (ref_span, lead_clip, trail_clip, query_len)."""
        self.ref_span = np.ascontiguousarray(stats[:, 0])
        self.lead_clip = np.ascontiguousarray(stats[:, 1])
        self.trail_clip = np.ascontiguousarray(stats[:, 2])
        self._ref_end = None
        self._max_span = None

    def ref_end(self):
        """htslib bam_endpos: pos + reference span, or pos + 1 for spanless/unmapped records."""
        if getattr(self, "_ref_end", None) is None:
            span = np.where((self.ref_span > 0) & ((self.flag & FLAG_UNMAPPED) == 0), self.ref_span, 1)
            self._ref_end = self.pos.astype(np.int64) + span
            self._ref_end_sorted = {}
        return self._ref_end

    def _bounds(self):
        if self._tid_bounds is None:
            nref = len(self.references)
            # records are sorted by (tid, pos) with unmapped-without-position (tid -1) last
            key = np.where(self.tid < 0, nref, self.tid)
            self._tid_bounds = np.searchsorted(key, np.arange(nref + 1), side="left")
            self._run_max_end = {}
        return self._tid_bounds

    def fetch_range(self, tid, start, end):
        """(first, stop): the records overlapping [start, end) on reference ``tid`` are those rows of [first, stop) whose
        :meth:`ref_end` lies behind ``start`` (what :meth:`fetch` returns as an index array)."""
        b = self._bounds()
        lo, hi = int(b[tid]), int(b[tid + 1])
        if lo == hi:
            return lo, lo
        pos = self.pos
        ref_end = self.ref_end()
        if getattr(self, "_max_span", None) is None:                 # no record reaches farther than this behind its start
            self._max_span = int((ref_end - pos).max()) if len(pos) else 0
        if getattr(self, "_pos_needle", None) is None:               # a needle of the array's own type: no converted copy of `pos`
            info = np.iinfo(pos.dtype)
            self._pos_needle = (pos.dtype.type, int(info.min), int(info.max))
        make, mn, mx = self._pos_needle
        stop = lo + int(pos[lo:hi].searchsorted(make(min(max(int(end), mn), mx)), side="left"))      # pos < end
        first = lo + int(pos[lo:hi].searchsorted(make(min(max(int(start) - self._max_span, mn), mx)), side="left"))
        return first, stop

    def fetch(self, tid, start, end):
        """Indices (file order) of records overlapping the half-open interval
        [start, end) on reference ``tid`` -- pysam ``fetch(contig, start, end)``."""
        first, stop = self.fetch_range(tid, start, end)
        if first >= stop:
            return np.empty(0, np.int64)
        idx = np.arange(first, stop, dtype=np.int64)
        return idx[self.ref_end()[first:stop] > start]

    def count_overlaps(self, tid, starts, ends):
        """Vectorised number of records overlapping each [start, end): the per-cluster
        coverage the reference gets by re-opening the BAM (classes.py:165-170)."""
        b = self._bounds()
        lo, hi = int(b[tid]), int(b[tid + 1])
        pos = self.pos[lo:hi]
        rend_all = self.ref_end()
        if tid not in self._ref_end_sorted:
            self._ref_end_sorted[tid] = np.sort(rend_all[lo:hi])
        rend = self._ref_end_sorted[tid]
        starts = np.asarray(starts, np.int64)
        ends = np.asarray(ends, np.int64)
        n_pos_lt_end = np.searchsorted(pos, ends, side="left")
        n_end_le_start = np.searchsorted(rend, starts, side="right")
        # overlap <=> pos < end and ref_end > start; records with ref_end <= start all have pos < end (end >= start)
        return np.where(ends >= starts, n_pos_lt_end - n_end_le_start, 0)


def concat_tables(tables):
    """Concatenate coordinate-sorted tables of DISJOINT reference sets (e.g. one per chromosome) into one table whose
    references are the inputs' references in order (records stay sorted by (tid, pos))."""
    refs, lens, names = [], [], []
    tid, pos, flag, mapq, l_seq, nid, cigar, off = [], [], [], [], [], [], [], [np.zeros(1, np.int64)]
    words = 0
    for t in tables:
        if t.seq_packed is not None:
            raise ValueError("concat_tables does not carry read bases")
        tid.append(np.where(t.tid >= 0, t.tid + len(refs), t.tid))
        nid.append(t.name_id + len(names))
        refs += t.references
        lens += t.lengths
        names += list(t.names)
        pos.append(t.pos); flag.append(t.flag); mapq.append(t.mapq); l_seq.append(t.l_seq); cigar.append(t.cigar)
        off.append(t.cig_off[1:] + words)
        words += int(t.cig_off[-1])
    cat = np.concatenate
    header = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (r, n) for r, n in zip(refs, lens))
    return AlignmentTable(refs, lens, cat(tid), cat(pos), cat(flag), cat(mapq), cat(l_seq), cat(nid), names, cat(cigar), cat(off),
                          header_text=header)


def read_bai(path):
    """.bai index -> per reference (first virtual offset, last virtual offset) of its records, or None when the
    reference has none (SAMv1 5.2; the pseudo-bin 37450 carries metadata, not chunks)."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:4] != b"BAI\x01":
        raise ValueError("%s is not a BAI index" % path)
    n_ref = struct.unpack_from("<i", raw, 4)[0]
    p = 8
    out = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", raw, p)[0]
        p += 4
        lo, hi = None, None
        for _b in range(n_bin):
            bin_id, n_chunk = struct.unpack_from("<Ii", raw, p)
            p += 8
            for _c in range(n_chunk):
                beg, end = struct.unpack_from("<QQ", raw, p)
                p += 16
                if bin_id != 37450:
                    lo = beg if lo is None else min(lo, beg)
                    hi = end if hi is None else max(hi, end)
        n_intv = struct.unpack_from("<i", raw, p)[0]
        p += 4 + 8 * n_intv
        out.append(None if lo is None else (lo, hi))
    return out


def _table_from_handle(lib, h, with_seq, threads=0, alloc=None):
    """A decoded handle (svx_bam_open / svx_bam_stream_next) -> AlignmentTable; the handle is closed.
    ``alloc(name, dtype, n)``: where the big arrays are created (a shared-memory file for the host helpers)."""
    import ctypes
    if alloc is None:
        def alloc(_name, dtype, n):
            return np.empty(n, dtype)
    try:
        sizes = np.zeros(8, np.uint64)
        lib.svx_bam_sizes(h, sizes.ctypes.data)
        n, nc, nref, _nn, nb, hb, rb, rawb = (int(v) for v in sizes)
        tid, pos, l_seq, name_id = (alloc(k, np.int32, n) for k in ("tid", "pos", "l_seq", "name_id"))
        flag, mapq = alloc("flag", np.uint16, n), alloc("mapq", np.uint8, n)
        cig_off, cigar = alloc("cig_off", np.int64, n + 1), alloc("cigar", np.uint32, nc)
        names, header, ref_names = alloc("names", np.uint8, nb), np.empty(hb, np.uint8), np.empty(rb, np.uint8)
        ref_lens = np.empty(nref, np.int32)
        seq_off = alloc("seq_off", np.int64, n) if with_seq else None
        lib.svx_bam_export(h, int(threads), tid.ctypes.data, pos.ctypes.data, flag.ctypes.data, mapq.ctypes.data,
                           l_seq.ctypes.data, name_id.ctypes.data, cig_off.ctypes.data, cigar.ctypes.data, names.ctypes.data,
                           header.ctypes.data, ref_names.ctypes.data, ref_lens.ctypes.data,
                           seq_off.ctypes.data if with_seq else None)
        seq_packed = None
        if with_seq:
            seq_packed = alloc("seq_packed", np.uint8, rawb)
            if rawb:
                ctypes.memmove(seq_packed.ctypes.data, lib.svx_bam_seq(h), rawb)
    finally:
        lib.svx_bam_close(h)
    name_list = names.tobytes().decode().split("\n")[:-1] if nb else []
    refs = ref_names.tobytes().decode().split("\n")[:-1] if rb else []
    table = AlignmentTable(refs, [int(v) for v in ref_lens], tid, pos, flag, mapq, l_seq, name_id, name_list, cigar, cig_off,
                           header.tobytes().decode(), seq_packed, seq_off)
    table._names_blob = names                                  # the '\n'-joined QNAMEs as they were exported (shared with the helpers)
    return table


def read_bam(path, with_seq=False, threads=0, tids=None, index=None):
    """Decode a BAM file into an :class:`AlignmentTable` with the native multi-threaded decoder of libsvx.so
    (svx_bam_*); ``with_seq``: keep the 4-bit read bases (needed by --hash only).  ``tids`` (with a ``.bai`` next to
    the file or given as ``index``): decode only the byte range holding those references' records -- one rank's
    chromosome shard -- instead of the whole file."""
    from .. import _lib
    lib = _lib.load()
    flags = 1 if with_seq else 0                          # SVX_BAM_KEEP_SEQ
    if tids is not None:
        spans = read_bai(index or (path + ".bai"))
        have = [spans[t] for t in tids if t < len(spans) and spans[t] is not None]
        if have:
            h = lib.svx_bam_open_range(path.encode(), int(threads), flags, min(s[0] for s in have), max(s[1] for s in have))
        else:
            h = lib.svx_bam_open_range(path.encode(), int(threads), flags, 0, 0)
        keep_tids = set(int(t) for t in tids)
    else:
        h = lib.svx_bam_open(path.encode(), int(threads), flags)
        keep_tids = None
    if not h:
        msg = lib.svx_bam_error().decode()
        raise ValueError("%s: %s" % (path, msg))
    table = _table_from_handle(lib, h, with_seq, threads)
    if table.seq_packed is not None:
        table.seq_packed = table.seq_packed.tobytes()
    if keep_tids is not None and len(table) and not set(np.unique(table.tid).tolist()) <= keep_tids:
        table = table.subset(np.flatnonzero(np.isin(table.tid, list(keep_tids))))     # references between two requested ones
    return table


def read_bai_linear(path):
    """.bai -> per reference None or (first virtual offset, last virtual offset, linear index uint64 array): the span of
    its records as in :func:`read_bai` plus the 16 kb linear index (SAMv1 5.1.3: for every 16 kb of the reference the
    virtual offset of the first record overlapping it, 0 where htslib left a gap)."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:4] != b"BAI\x01":
        raise ValueError("%s is not a BAI index" % path)
    n_ref = struct.unpack_from("<i", raw, 4)[0]
    p = 8
    out = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", raw, p)[0]
        p += 4
        lo, hi = None, None
        unpack = struct.unpack_from
        for _b in range(n_bin):                                # (plain ints: a whole-genome index has ~10^5 bins, a NumPy call per bin cost seconds)
            bin_id, n_chunk = unpack("<Ii", raw, p)
            p += 8
            if bin_id != 37450 and n_chunk:
                vals = unpack("<%dQ" % (2 * n_chunk), raw, p)
                b0, e1 = min(vals[0::2]), max(vals[1::2])
                lo = b0 if lo is None or b0 < lo else lo
                hi = e1 if hi is None or e1 > hi else hi
            p += 16 * n_chunk
        n_intv = struct.unpack_from("<i", raw, p)[0]
        p += 4
        linear = np.frombuffer(raw, "<u8", n_intv, p).copy()
        p += 8 * n_intv
        out.append(None if lo is None else (lo, hi, linear))
    return out


def read_bam_header(path):
    """Header text + reference dictionary only (an AlignmentTable without records); no index needed."""
    from .. import _lib
    lib = _lib.load()
    h = lib.svx_bam_open_range(path.encode(), 1, 0, 0, 0)
    if not h:
        raise ValueError("%s: %s" % (path, lib.svx_bam_error().decode()))
    return _table_from_handle(lib, h, False)


def find_index(path):
    """The .bai next to a BAM (``x.bam.bai`` or ``x.bai``), or None."""
    return next((c for c in (path + ".bai", os.path.splitext(path)[0] + ".bai") if os.path.exists(c)), None)


class BamStream:
    """Iterator over a BAM file's reference sequences: yields one :class:`AlignmentTable` per reference that has
    records (file order; all tables carry the whole reference dictionary and the file's ``tid`` numbering; QNAME ids
    count from 0 in every table), each as soon as it is decoded, while the native reader's own threads read and inflate
    the next ones (svx_bam_stream_*).  ``tids`` + a ``.bai``: only those references' byte ranges are read (a rank's
    chromosomes); ``tids`` without an index: the whole file is streamed and the other references are skipped.

    What the reference does window by window through pysam's ``fetch`` (run_collection.py:23-26)."""

    def __init__(self, path, with_seq=False, threads=0, tids=None, index=None, alloc=None):
        from .. import _lib
        self.lib = _lib.load()
        self.path, self.with_seq, self.alloc = path, with_seq, alloc
        self.keep = None if tids is None else set(int(t) for t in tids)
        voffs = np.zeros(0, np.uint64)
        if tids is not None and index is None:
            index = find_index(path)
        if tids is not None and index is not None:
            spans = read_bai(index)
            have = sorted(spans[t] for t in self.keep if t < len(spans) and spans[t] is not None)
            voffs = np.asarray([v for span in have for v in span], np.uint64)
            if not have:                                       # nothing of this rank's in the file: an empty range
                voffs = np.zeros(2, np.uint64)
        self.h = self.lib.svx_bam_stream_open(path.encode(), int(threads), 1 if with_seq else 0,
                                              voffs.ctypes.data if voffs.size else None, int(voffs.size // 2))
        if not self.h:
            raise ValueError("%s: %s" % (path, self.lib.svx_bam_error().decode()))

    def __iter__(self):
        return self

    def __next__(self):
        import ctypes
        while True:
            if self.h is None:
                raise StopIteration
            status = ctypes.c_int(0)
            part = self.lib.svx_bam_stream_next(self.h, ctypes.byref(status))
            if not part:
                err = self.lib.svx_bam_error().decode() if status.value < 0 else None
                self.close()
                if err is not None:
                    raise ValueError("%s: %s" % (self.path, err))
                raise StopIteration
            table = _table_from_handle(self.lib, part, self.with_seq, alloc=self.alloc)
            if len(table) and (self.keep is None or int(table.tid[0]) in self.keep):
                return table

    def close(self):
        if self.h is not None:
            self.lib.svx_bam_stream_close(self.h)
            self.h = None

    def __del__(self):
        self.close()


def read_bam_python(path, with_seq=False):
    """Pure-Python decoder of the same format (zlib + NumPy); kept as the independent check of the native one."""
    with open(path, "rb") as f:
        raw = bgzf_decompress(f.read())
    if raw[:4] != b"BAM\x01":
        raise ValueError("%s is not a BAM file" % path)
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text].split(b"\x00")[0].decode()
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]
    p += 4
    refs, lens = [], []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, p)[0]
        refs.append(raw[p + 4:p + 4 + l_name - 1].decode())
        lens.append(struct.unpack_from("<i", raw, p + 4 + l_name)[0])
        p += 8 + l_name
    # pass 1: record offsets
    offs = []
    n = len(raw)
    unpack_i = struct.Struct("<i").unpack_from
    while p < n:
        offs.append(p)
        p += 4 + unpack_i(raw, p)[0]
    buf = np.frombuffer(raw, np.uint8)
    offs = np.asarray(offs, np.int64)
    nrec = offs.size

    def field(off, dtype, width):
        idx = offs[:, None] + (off + np.arange(width))[None, :]
        return buf[idx].copy().view(dtype).reshape(-1)

    tid = field(4, "<i4", 4)
    pos = field(8, "<i4", 4)
    l_name = field(12, "u1", 1).astype(np.int64)
    mapq = field(13, "u1", 1)
    n_cig = field(16, "<u2", 2).astype(np.int64)
    flag = field(18, "<u2", 2)
    l_seq = field(20, "<i4", 4)
    cig_off = np.zeros(nrec + 1, np.int64)
    cig_off[1:] = np.cumsum(n_cig)
    cig_start = offs + 36 + l_name
    # gather CIGAR words: flat byte index of every word
    total = int(cig_off[-1])
    rec_of_word = np.repeat(np.arange(nrec), n_cig)
    word_in_rec = np.arange(total) - cig_off[rec_of_word]
    byte0 = cig_start[rec_of_word] + 4 * word_in_rec
    cigar = buf[(byte0[:, None] + np.arange(4)[None, :])].copy().view("<u4").reshape(-1) if total else np.empty(0, np.uint32)
    names, name_ids, seen = [], np.empty(nrec, np.int32), {}
    for i in range(nrec):
        o = int(offs[i]) + 36
        nm = raw[o:o + int(l_name[i]) - 1]
        j = seen.get(nm)
        if j is None:
            j = len(names)
            seen[nm] = j
            names.append(nm.decode())
        name_ids[i] = j
    seq_packed = seq_off = None
    if with_seq:
        seq_packed = raw                                   # 4-bit SEQ is used in place from the decompressed file
        seq_off = cig_start + 4 * n_cig
    return AlignmentTable(refs, lens, tid, pos, flag, mapq, l_seq, name_ids, names, cigar, cig_off, text, seq_packed, seq_off)


def _reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def write_bam(path, table, with_seq=True, level=1, index=False):
    """Encode an :class:`AlignmentTable` as a coordinate-sorted BAM (SEQ = N's for
    records with l_seq > 0, '*' otherwise; QUAL 0xFF; no tags)."""
    text = table.header_text or ("@HD\tVN:1.6\tSO:coordinate\n" + "".join(
        "@SQ\tSN:%s\tLN:%d\n" % (r, l) for r, l in zip(table.references, table.lengths)))
    parts = [b"BAM\x01", struct.pack("<i", len(text)), text.encode(), struct.pack("<i", len(table.references))]
    for r, l in zip(table.references, table.lengths):
        parts.append(struct.pack("<i", len(r) + 1) + r.encode() + b"\x00" + struct.pack("<i", l))
    span_ops = np.array([1, 0, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0], np.int64)
    for i in range(len(table)):
        cw = table.cigar[table.cig_off[i]:table.cig_off[i + 1]]
        span = int(((cw >> 4).astype(np.int64) * span_ops[cw & 15]).sum())
        name = table.names[table.name_id[i]].encode() + b"\x00"
        l_seq = int(table.l_seq[i]) if with_seq else 0
        pos = int(table.pos[i])
        aux = b""
        cig_field = cw.astype("<u4").tobytes()
        if cw.size > 65535:                                   # SAMv1 4.2.2: real CIGAR in CG:B,I, placeholder kSmN in the record
            aux = b"CGBI" + struct.pack("<I", cw.size) + cig_field
            cig_field = struct.pack("<II", (l_seq << 4) | 4, (span << 4) | 3)
        body = struct.pack("<iiBBHHHIiii", int(table.tid[i]), pos, len(name), int(table.mapq[i]),
                           _reg2bin(pos, pos + max(span, 1)), len(cig_field) // 4, int(table.flag[i]), l_seq, -1, -1, 0)
        if l_seq and table.seq_packed is not None:
            o = int(table.seq_off[i])
            seq_bytes = bytes(table.seq_packed[o:o + (l_seq + 1) // 2])
        else:
            seq_bytes = b"\xff" * ((l_seq + 1) // 2)       # N's
        body += name + cig_field + seq_bytes + b"\xff" * l_seq + aux
        parts.append(struct.pack("<i", len(body)) + body)
    block_coff = []
    with open(path, "wb") as f:
        f.write(bgzf_compress(b"".join(parts), level, block_coff))
    if index:
        _write_bai(path + ".bai", table, block_coff, [len(x) for x in parts])


def _write_bai(path, table, block_coff, part_sizes):
    """Standard .bai (binning + 16 kb linear index, SAMv1 5.2) for a file written by :func:`write_bam`."""
    def voff(u):                                        # blocks are cut every _MAX_BLOCK uncompressed bytes
        blk = u // _MAX_BLOCK
        if blk >= len(block_coff) - 1:                  # end of data: start of the EOF block
            return block_coff[-1] << 16
        return (block_coff[blk] << 16) | (u - blk * _MAX_BLOCK)

    n_head = len(part_sizes) - len(table)
    u = sum(part_sizes[:n_head])
    span_ops = np.array([1, 0, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0], np.int64)
    bins = [dict() for _ in table.references]
    linear = [dict() for _ in table.references]
    for i in range(len(table)):
        size = part_sizes[n_head + i]
        t = int(table.tid[i])
        if t >= 0:
            cw = table.cigar[table.cig_off[i]:table.cig_off[i + 1]]
            span = max(1, int(((cw >> 4).astype(np.int64) * span_ops[cw & 15]).sum()))
            beg, end = int(table.pos[i]), int(table.pos[i]) + span
            v0, v1 = voff(u), voff(u + size)
            chunks = bins[t].setdefault(_reg2bin(beg, end), [])
            if chunks and chunks[-1][1] == v0:
                chunks[-1][1] = v1
            else:
                chunks.append([v0, v1])
            for w in range(beg >> 14, ((end - 1) >> 14) + 1):
                linear[t].setdefault(w, v0)
        u += size
    out = [b"BAI\x01", struct.pack("<i", len(table.references))]
    for t in range(len(table.references)):
        out.append(struct.pack("<i", len(bins[t])))
        for b, chunks in sorted(bins[t].items()):
            out.append(struct.pack("<Ii", b, len(chunks)) + b"".join(struct.pack("<QQ", c0, c1) for c0, c1 in chunks))
        n_intv = (max(linear[t]) + 1) if linear[t] else 0
        out.append(struct.pack("<i", n_intv))
        last = 0
        for w in range(n_intv):
            last = linear[t].get(w, last)
            out.append(struct.pack("<Q", last))
    with open(path, "wb") as f:
        f.write(b"".join(out))


# ---------------------------------------------------------------------------------------------------------------------
# Fast writer for large synthetic samples (bench.py --from-bam): NumPy-assembled record stream, realistic SEQ / QUAL
# entropy, libdeflate (ctypes) per BGZF block, one independently compressed segment per reference so that the
# references can be encoded in parallel processes and concatenated (BGZF members are independent).
_SPAN_OPS = np.array([1, 0, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0], np.int64)


def reference_spans(table):
    """Reference bases covered by every record (>= 1), from the CIGAR words."""
    n = len(table)
    if n == 0:
        return np.zeros(0, np.int64)
    contrib = (table.cigar >> 4).astype(np.int64) * _SPAN_OPS[table.cigar & 15]
    csum = np.concatenate([[0], np.cumsum(contrib)])
    return np.maximum(1, csum[table.cig_off[1:]] - csum[table.cig_off[:-1]])


def encode_record_stream(table, seq="random", seed=0):
    """The uncompressed BAM record stream of ``table`` -> (uint8 array, int64 offsets [n+1] of the records in it).
    ``seq``: 'random' = uniformly random bases (2 bits of entropy per base, like real reads) and HiFi-like binned
    qualities (7 values, skewed); 'N' = N bases and 0xFF qualities (what :func:`write_bam` writes).  A CIGAR of more than
    65535 operations goes into the record's CG:B,I tag behind the placeholder "<l_seq>S<span>N" (SAMv1 4.2.2)."""
    n = len(table)
    n_real = (table.cig_off[1:] - table.cig_off[:-1]).astype(np.int64)
    long_ = n_real > 65535
    n_cig = np.where(long_, 2, n_real)                        # the words in the record's own CIGAR field
    aux_len = np.where(long_, 8 + 4 * n_real, 0)              # "CG" "B" "I" count words
    names = [table.names[i].encode() + b"\x00" for i in table.name_id]
    l_name = np.fromiter((len(b) for b in names), np.int64, n)
    l_seq = table.l_seq.astype(np.int64)
    sq = (l_seq + 1) // 2
    size = 32 + l_name + 4 * n_cig + sq + l_seq + aux_len
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum(size + 4)
    buf = np.empty(int(off[-1]), np.uint8)
    span = reference_spans(table)
    pos = table.pos.astype(np.int64)
    fixed = np.zeros(n, np.dtype([("block_size", "<i4"), ("tid", "<i4"), ("pos", "<i4"), ("l_name", "u1"), ("mapq", "u1"), ("bin", "<u2"),
                                  ("n_cig", "<u2"), ("flag", "<u2"), ("l_seq", "<i4"), ("next_tid", "<i4"), ("next_pos", "<i4"), ("tlen", "<i4")]))
    fixed["block_size"], fixed["tid"], fixed["pos"], fixed["l_name"], fixed["mapq"] = size, table.tid, table.pos, l_name, table.mapq
    fixed["bin"] = _reg2bin_vec(pos, pos + span)
    fixed["n_cig"], fixed["flag"], fixed["l_seq"], fixed["next_tid"], fixed["next_pos"] = n_cig, table.flag, table.l_seq, -1, -1
    buf[(off[:-1, None] + np.arange(36)).ravel()] = fixed.view(np.uint8).ravel()

    def scatter(starts, lengths, payload):
        tot = int(lengths.sum())
        if tot:
            first = np.zeros(len(lengths), np.int64)
            first[1:] = np.cumsum(lengths)[:-1]
            buf[np.repeat(starts - first, lengths) + np.arange(tot)] = payload
    scatter(off[:-1] + 36, l_name, np.frombuffer(b"".join(names), np.uint8))
    words = table.cigar.astype("<u4")
    if long_.any():                                           # the short records' words in one scatter, the long ones one by one
        short = np.flatnonzero(~long_)
        keep = np.repeat(~long_, n_real)
        scatter((off[:-1] + 36 + l_name)[short], 4 * n_real[short], words[keep].view(np.uint8))
        for i in np.flatnonzero(long_):
            a = int(off[i] + 36 + l_name[i])
            buf[a:a + 8] = np.array([(int(l_seq[i]) << 4) | 4, (int(span[i]) << 4) | 3], "<u4").view(np.uint8)
            t = int(a + 8 + sq[i] + l_seq[i])
            buf[t:t + 4] = np.frombuffer(b"CGBI", np.uint8)
            buf[t + 4:t + 8] = np.array([int(n_real[i])], "<u4").view(np.uint8)
            buf[t + 8:t + 8 + 4 * int(n_real[i])] = words[int(table.cig_off[i]):int(table.cig_off[i + 1])].view(np.uint8)
    else:
        scatter(off[:-1] + 36 + l_name, 4 * n_cig, words.view(np.uint8))
    at = off[:-1] + 36 + l_name + 4 * n_cig
    if seq == "N":
        for i in range(n):
            buf[at[i]:at[i] + sq[i] + l_seq[i]] = 0xFF
    else:
        rng = np.random.default_rng(seed)
        codes = np.array([1, 2, 4, 8], np.uint8)
        pair = ((codes[:, None] << 4) | codes[None, :]).ravel()
        # qualities: 7 bins with probabilities ~ .70 .10 .07 .05 .04 .03 .01 (a 256-entry table indexed by random bytes)
        qlut = np.repeat(np.array([93, 40, 35, 30, 25, 20, 10], np.uint8), [179, 26, 18, 13, 10, 8, 2])
        # a pool of random bases / qualities a few hundred deflate windows (32 KB) long; every record copies a slice from
        # its own pseudo-random offset: for the compressor that is random data (drawing 1.4 G fresh values per
        # chromosome-sized sample would take minutes in NumPy and compress to the same size)
        longest = int(l_seq.max()) if n else 0
        pool_n = max(16 << 20, 4 * longest)
        raw = rng.bit_generator.random_raw((2 * pool_n + 7) // 8).view(np.uint8)
        seq_pool = pair[raw[:pool_n] & 15]
        qual_pool = qlut[raw[pool_n:2 * pool_n]]
        for i in range(n):                                     # two large slice copies per record
            a, k, q = int(at[i]), int(sq[i]), int(l_seq[i])
            o = (i * 2654435761) % (pool_n - longest)
            buf[a:a + k] = seq_pool[o:o + k]
            buf[a + k:a + k + q] = qual_pool[o:o + q]
    return buf, off


def _reg2bin_vec(beg, end):
    end = end - 1
    out = np.zeros(beg.shape, np.int64)
    done = np.zeros(beg.shape, bool)
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        hit = ~done & ((beg >> shift) == (end >> shift))
        out[hit] = base + (beg[hit] >> shift)
        done |= hit
    return out


class _Deflater:
    """Raw deflate of one block: libdeflate through ctypes when the library is on the machine, zlib otherwise."""

    def __init__(self, level=1):
        import ctypes
        self.level, self.ld, self.c = level, None, None
        try:
            ld = ctypes.CDLL("libdeflate.so.0")
            ld.libdeflate_alloc_compressor.restype = ctypes.c_void_p
            ld.libdeflate_alloc_compressor.argtypes = [ctypes.c_int]
            ld.libdeflate_deflate_compress.restype = ctypes.c_size_t
            ld.libdeflate_deflate_compress.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
            self.ld, self.c = ld, ld.libdeflate_alloc_compressor(level)
            self.out = np.empty(_MAX_BLOCK + 1024, np.uint8)
        except OSError:
            pass

    def __call__(self, chunk):
        """chunk: contiguous uint8 array -> compressed bytes."""
        if self.c:
            k = self.ld.libdeflate_deflate_compress(self.c, chunk.ctypes.data, chunk.size, self.out.ctypes.data, self.out.size)
            if k:
                return self.out[:k].tobytes()
        co = zlib.compressobj(self.level, zlib.DEFLATED, -15)
        return co.compress(chunk.tobytes()) + co.flush()


def bgzf_segment(stream, level=1):
    """uint8 array -> (BGZF blocks without the EOF marker as bytes, file offset of every block relative to the segment
    [n_blocks + 1]); blocks are cut every _MAX_BLOCK bytes."""
    deflate = _Deflater(level)
    out, coff, at = [], [], 0
    for i in range(0, int(stream.size), _MAX_BLOCK):
        chunk = stream[i:i + _MAX_BLOCK]
        cdata = deflate(chunk)
        blk = (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cdata) + 25) + cdata
               + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, chunk.size))
        coff.append(at)
        at += len(blk)
        out.append(blk)
    coff.append(at)
    return b"".join(out), np.asarray(coff, np.int64)


def encode_reference_segment(table, seq="random", seed=0, level=1):
    """One reference's records as an independently compressed BGZF segment plus what the index needs:
    -> dict(data=bytes, coff=block offsets, rec_off=record offsets in the uncompressed segment, pos, span, tid)."""
    stream, rec_off = encode_record_stream(table, seq, seed)
    data, coff = bgzf_segment(stream, level)
    return {"data": data, "coff": coff, "rec_off": rec_off, "pos": table.pos.astype(np.int64), "span": reference_spans(table),
            "tid": int(table.tid[0]) if len(table) else -1, "inflated": int(stream.size)}


def write_bam_segments(path, references, lengths, segments, index=True):
    """Header + the segments of :func:`encode_reference_segment` (ascending tid) + EOF marker -> ``path`` (+ ``.bai``)."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (r, l) for r, l in zip(references, lengths))
    head = [b"BAM\x01", struct.pack("<i", len(text)), text.encode(), struct.pack("<i", len(references))]
    for r, l in zip(references, lengths):
        head.append(struct.pack("<i", len(r) + 1) + r.encode() + b"\x00" + struct.pack("<i", l))
    head_bytes = bgzf_compress(b"".join(head), 1)[:-len(_BGZF_EOF)]
    bins = [dict() for _ in references]
    linear = [dict() for _ in references]
    with open(path, "wb") as f:
        f.write(head_bytes)
        base = len(head_bytes)
        for seg in segments:
            n_data = 0
            if seg.get("data") is None and seg.get("data_path"):      # a segment that waits in a file of its own (bench.py: whole-genome jobs)
                import shutil
                with open(seg["data_path"], "rb") as g:
                    shutil.copyfileobj(g, f, 64 << 20)
                n_data = os.path.getsize(seg["data_path"])
                os.remove(seg["data_path"])
            else:
                f.write(seg["data"])
                n_data = len(seg["data"])
            if index and seg["tid"] >= 0:
                t, coff, ro = seg["tid"], seg["coff"] + base, seg["rec_off"]
                blk = ro // _MAX_BLOCK
                v = (coff[np.minimum(blk, len(coff) - 1)] << 16) | np.where(blk < len(coff) - 1, ro - blk * _MAX_BLOCK, 0)
                beg, end = seg["pos"], seg["pos"] + seg["span"]
                rb = _reg2bin_vec(beg, end)
                for i in range(len(beg)):
                    chunks = bins[t].setdefault(int(rb[i]), [])
                    v0, v1 = int(v[i]), int(v[i + 1])
                    if chunks and chunks[-1][1] == v0:
                        chunks[-1][1] = v1
                    else:
                        chunks.append([v0, v1])
                    for w in range(int(beg[i]) >> 14, ((int(end[i]) - 1) >> 14) + 1):
                        linear[t].setdefault(w, v0)
            base += n_data
        f.write(_BGZF_EOF)
    if index:
        _dump_bai(path + ".bai", bins, linear)


def _dump_bai(path, bins, linear):
    out = [b"BAI\x01", struct.pack("<i", len(bins))]
    for t in range(len(bins)):
        out.append(struct.pack("<i", len(bins[t])))
        for b, chunks in sorted(bins[t].items()):
            out.append(struct.pack("<Ii", b, len(chunks)) + b"".join(struct.pack("<QQ", c0, c1) for c0, c1 in chunks))
        n_intv = (max(linear[t]) + 1) if linear[t] else 0
        out.append(struct.pack("<i", n_intv))
        last = 0
        for w in range(n_intv):
            last = linear[t].get(w, last)
            out.append(struct.pack("<Q", last))
    with open(path, "wb") as f:
        f.write(b"".join(out))


def pack_sequence(seq):
    """ASCII bases -> BAM 4-bit packed bytes."""
    codes = _SEQ_CODE[np.frombuffer(seq if isinstance(seq, (bytes, bytearray)) else seq.encode(), np.uint8)]
    if codes.size % 2:
        codes = np.concatenate([codes, np.zeros(1, np.uint8)])
    return ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8).tobytes()


class _LazySequences(dict):
    """name -> bases; a contig of a FASTA file is read (one slice of the memory map, newlines dropped at C speed) the
    first time it is asked for."""

    def __init__(self, loader):
        super().__init__()
        self._loader = loader

    def __missing__(self, name):
        seq = self._loader(name)
        self[name] = seq
        return seq


class Fasta:
    """FASTA with pysam.FastaFile's ``references`` / ``fetch`` / ``get_reference_length``.  A file is memory mapped and
    indexed by its header lines (or its ``.fai``); contigs are materialised on first use, so a rank that works on three
    chromosomes does not parse the other twenty-one."""

    def __init__(self, path=None, sequences=None):
        self.references = []
        self._seq = {}
        self._length = {}
        if sequences is not None:
            for name, seq in sequences.items():
                self.references.append(name)
                self._seq[name] = seq if isinstance(seq, (bytes, bytearray)) else str(seq).encode()
        elif path is not None:
            self._open(path)

    def _open(self, path):
        import mmap
        self._ranges = {}
        size = os.path.getsize(path)
        if size == 0:
            return
        self._file = open(path, "rb")
        self._map = mmap.mmap(self._file.fileno(), 0, access=mmap.ACCESS_READ)
        mm = self._map
        fai = path + ".fai"
        compressed = mm[:2] == b"\x1f\x8b"
        if compressed:
            # gzip / bgzip FASTA (pysam.FastaFile reads bgzip): inflate once into memory and index the text; the .fai of a
            # bgzip file describes the uncompressed text, so it stays valid.  (Random access through a .gzi is not needed:
            # every contig a rank works on is read in full anyway.)
            import gzip
            with gzip.open(path, "rb") as gz:
                mm = self._map = gz.read()
            size = len(mm)
            if size == 0:
                return
        if os.path.exists(fai) and (compressed or os.path.getmtime(fai) >= os.path.getmtime(path)):
            with open(fai) as f:                       # name, length, offset, bases per line, bytes per line (faidx)
                for line in f:
                    p = line.rstrip("\n").split("\t")
                    if len(p) < 5:
                        continue
                    name, length, offset, linebases, linewidth = p[0], int(p[1]), int(p[2]), int(p[3]), int(p[4])
                    n_lines = (length + linebases - 1) // linebases if linebases else 0
                    end = min(size, offset + length + n_lines * max(linewidth - linebases, 0))
                    self.references.append(name)
                    self._ranges[name] = (offset, end)
                    self._length[name] = length
        else:
            pos = 0 if mm[:1] == b">" else mm.find(b"\n>") + 1
            while 0 <= pos < size and mm[pos:pos + 1] == b">":
                eol = mm.find(b"\n", pos)
                eol = size if eol < 0 else eol
                name = mm[pos + 1:eol].split()[0].decode() if eol > pos + 1 else ""
                nxt = mm.find(b"\n>", eol)
                end = size if nxt < 0 else nxt + 1
                self.references.append(name)
                self._ranges[name] = (min(eol + 1, size), end)
                pos = end
        self._seq = _LazySequences(lambda name: self._map[self._ranges[name][0]:self._ranges[name][1]].translate(None, b"\r\n \t"))

    def get_reference_length(self, name):
        if name not in self._length:
            if name in self._seq or not hasattr(self, "_ranges"):
                self._length[name] = len(self._seq[name])
            else:                                       # count the bases without keeping them
                lo, hi = self._ranges[name]
                raw = self._map[lo:hi]
                self._length[name] = len(raw) - raw.count(b"\n") - raw.count(b"\r") - raw.count(b" ") - raw.count(b"\t")
        return self._length[name]

    def fetch(self, name, start, end):
        """0-based half-open; clipped to the contig like htslib faidx."""
        seq = self._seq[name]
        start = max(0, int(start))
        end = min(len(seq), int(end))
        return seq[start:end].decode() if end > start else ""

    def fetch_bytes(self, name, start, end):
        seq = self._seq[name]
        return seq[max(0, int(start)):max(0, min(len(seq), int(end)))]

    def fetch_view(self, name, start, end):
        """:meth:`fetch_bytes` without the copy: a memoryview of the same bytes (len() and integer indexing like bytes)."""
        seq = self._seq[name]
        lo, hi = max(0, int(start)), max(0, min(len(seq), int(end)))
        return memoryview(seq)[lo:hi] if isinstance(seq, (bytes, bytearray)) else seq[lo:hi]


def write_fasta(path, sequences, width=60):
    with open(path, "wb") as f:
        for name, seq in sequences.items():
            seq = seq if isinstance(seq, (bytes, bytearray)) else str(seq).encode()
            f.write(b">" + name.encode() + b"\n")
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width] + b"\n")
    with open(path + ".fai", "w") as f:
        off = 0
        for name, seq in sequences.items():
            off += len(name) + 2
            f.write("%s\t%d\t%d\t%d\t%d\n" % (name, len(seq), off, width, width + 1))
            off += len(seq) + (len(seq) + width - 1) // width
