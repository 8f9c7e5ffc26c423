"""Genotyping of a called site from the reads around it.

Mirror of the reference's ``genotyper`` (src/network/genotype.py:17-72): the first 500
usable alignments within +-1000 bp that are not alt-supporting reads vote for the
reference allele.  The BAM re-fetch per record is replaced by index arithmetic on the
resident :class:`AlignmentTable`.
"""
import numpy as np

try:
    import cython as _cython
    _COMPILED = bool(_cython.compiled)
except ImportError:                                           # interpreted and no Cython on the machine
    _COMPILED = False


_DTYPES = (np.dtype(np.int32), np.dtype(np.int32), np.dtype(np.int32), np.dtype(np.uint16), np.dtype(np.uint8))   # what genotype.pxd declares


def _ref_names(pos, span, ref_end, flag, mapq, name_id, first, stop, s0, alt_ids, min_mapq, mode, start, end, ov, out):
    """The scan of :func:`genotyper` as one pass over the rows [first, stop) (compiled with C types, genotype.pxd; the
    NumPy form below is what runs interpreted).  mode 0: every usable row votes; 1: DEL / INV; 2: INS / DUP; 3: any other
    single type (no row votes).  -> number of name ids written to ``out``."""
    n_alt = alt_ids.shape[0]
    usable = 0
    k = 0
    for r in range(first, stop):
        if ref_end[r] <= s0:
            continue
        if (flag[r] & 0x104) != 0 or mapq[r] < min_mapq:
            continue
        nm = name_id[r]
        is_alt = False
        for a in range(n_alt):
            if alt_ids[a] == nm:
                is_alt = True
                break
        if is_alt:
            continue
        usable += 1
        if usable > 500:                                      # aln_no < 500 (:33-43)
            break
        if mode == 0:
            hit = True
        elif mode == 3:
            hit = False
        else:
            rs = pos[r]
            re = rs + span[r]
            if mode == 1:
                hit = (rs < end - ov and re > end + 100) or (rs < start - 100 and re > start + ov)
            else:
                hit = rs < start - 100 and re > end + 100
        if hit:
            out[k] = nm
            k += 1
    return k


def genotyper(candidate, support_reads, options, sample):
    contig, start, end, svtype = candidate
    table = sample.table
    tid = table.get_tid(contig)
    clen = table.lengths[tid]
    alt = set(support_reads)
    alt_no = len(alt)
    if _COMPILED and (table.pos.dtype, table.ref_span.dtype, table.name_id.dtype, table.flag.dtype, table.mapq.dtype) == _DTYPES:
        s0 = max(0, start - 1000)
        first, stop = table.fetch_range(tid, s0, min(clen, end + 1000))
        if len(svtype) != 1:
            mode = 0
        else:
            mode = 1 if svtype[0] in ("DEL", "INV") else 2 if svtype[0] in ("INS", "DUP") else 3
        out = np.empty(500, np.int32)
        k = _ref_names(table.pos, table.ref_span, table.ref_end(), table.flag, table.mapq, table.name_id, first, stop, s0,
                       np.ascontiguousarray(table.ids_of(alt), np.int64), int(options.min_mapq), mode, start, end,
                       float(min((end - start) / 2, 2000)), out)
        return _call(svtype, len(set(out[:k].tolist())), alt_no, options)
    rows = table.fetch(tid, max(0, start - 1000), min(clen, end + 1000))
    if rows.size:
        names = table.name_id[rows]
        alt_ids = table.ids_of(alt)
        # (np.isin sorts both sides: ~90 us per call for a few hundred names against a few dozen ids)
        usable = ~(names[:, None] == alt_ids[None, :]).any(axis=1) if alt_ids.size <= 64 else ~np.isin(names, alt_ids)
        usable &= ((table.flag[rows] & (0x4 | 0x100)) == 0) & (table.mapq[rows] >= options.min_mapq)
        rows = rows[usable][:500]                             # aln_no < 500 (:33-43)
    if rows.size == 0:
        ref_names = np.empty(0, np.int32)
    elif len(svtype) == 1:
        rs = table.pos[rows].astype(np.int64)
        re = rs + table.ref_span[rows]
        if svtype[0] in ("DEL", "INV"):
            ov = min((end - start) / 2, 2000)
            hit = ((rs < end - ov) & (re > end + 100)) | ((rs < start - 100) & (re > start + ov))
        else:
            hit = np.zeros(rows.size, bool)
        if svtype[0] in ("INS", "DUP"):
            hit = hit | ((rs < start - 100) & (re > end + 100))
        ref_names = table.name_id[rows][hit]
    else:
        ref_names = table.name_id[rows]
    return _call(svtype, len(set(ref_names.tolist())), alt_no, options)


def _call(svtype, ref_no, alt_no, options):
    gt = "./."
    if len(svtype) != 1:
        return gt, ref_no, alt_no
    if alt_no + ref_no >= options.min_gt_depth:
        ratio = alt_no / (alt_no + ref_no)
        if ratio >= options.homo_thresh:
            gt = "1/1"
        elif ratio >= options.hete_thresh:
            gt = "0/1"
        else:
            gt = "0/0"
    return gt, ref_no, alt_no
