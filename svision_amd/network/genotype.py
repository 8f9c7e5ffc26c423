"""Genotyping of a called site from the reads around it.

Mirror of the reference's ``genotyper`` (src/network/genotype.py:17-72): the first 500
usable alignments within +-1000 bp that are not alt-supporting reads vote for the
reference allele.  The BAM re-fetch per record is replaced by index arithmetic on the
resident :class:`AlignmentTable`.
"""
import numpy as np


def genotyper(candidate, support_reads, options, sample):
    contig, start, end, svtype = candidate
    table = sample.table
    tid = table.get_tid(contig)
    clen = table.lengths[tid]
    rows = table.fetch(tid, max(0, start - 1000), min(clen, end + 1000))
    alt = set(support_reads)
    alt_no = len(alt)
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
    ref_no = len(set(ref_names.tolist()))
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
