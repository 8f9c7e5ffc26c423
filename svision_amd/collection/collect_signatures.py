"""Window-level signature collection: alignments -> reads -> segments -> Signatures.

Host-side mirror of the reference's ``analyze_alignments``
(src/collection/collect_signatures.py:114-310).  The per-alignment work that is
proportional to CIGAR length already happened on the GPU (``svx_cigar_scan``:
long gaps + reference span + clip lengths for every record of the sample), so
this module only touches reads that can yield a signature: reads with two or
more usable alignments in the window, or with a long in-CIGAR gap.  All other
reads produce exactly one segment in the reference (:226-227) and are skipped
with vectorised masks.
"""
import numpy as np

from .analyze_reads import analyze_between_aligns, analyze_gap, analyze_inside_align
from .classes import by_read_pos
from .graph import build_graph


def _emit(ctx, cur, nxt, helpers, next_is_last):
    """Type the junction cur -> nxt of one read (on private copies) and keep its Signature, if any."""
    signatures, want_graph, whole_seq, chrom_of, fetch_ref, sample, options, qname = ctx
    cur, nxt = cur.copy(), nxt.copy()
    graph = None
    if want_graph:                                            # before analyze_gap shifts / trims anything (:236-237, :298-302)
        graph = build_graph(cur, nxt, helpers, options.min_sv_size, whole_seq, chrom_of, sample.fetch_ref_str, qname, next_is_last)
    sig = analyze_gap(cur, nxt, chrom_of, fetch_ref, options, qname, helpers)
    if sig is not None:
        sig.set_graph(graph)
        signatures.append(sig)


def analyze_alignments(rows, sample, options, part_num=0):
    """rows: record indices of one fetch window, in file order -> list[Signature] in
    read-first-occurrence order (the order of the reference's ``reads_dict``)."""
    table = sample.table
    min_mapq = 0 if options.contig else options.min_mapq
    rows = np.asarray(rows, np.int64)
    if rows.size == 0:
        return []
    flag = table.flag[rows]
    n_cig = table.cig_off[rows + 1] - table.cig_off[rows]
    keep = (n_cig > 0) & ((flag & (0x4 | 0x100)) == 0) & (table.mapq[rows] >= min_mapq)   # :131-139
    rows = rows[keep]
    if rows.size == 0:
        return []
    name = table.name_id[rows]
    gap_cnt = (sample.gap_off[rows + 1] - sample.gap_off[rows]).astype(np.int64)
    # per read: number of records and of long gaps in this window
    uniq, first_idx, inv = np.unique(name, return_index=True, return_inverse=True)
    n_rec = np.bincount(inv, minlength=uniq.size)
    n_gap = np.bincount(inv, weights=gap_cnt, minlength=uniq.size)
    cand = (n_rec >= 2) | (n_gap > 0)
    if not cand.any():
        return []
    order = np.argsort(first_idx[cand], kind="stable")      # read insertion order (:152-162)
    cand_ids = np.flatnonzero(cand)[order]
    # rows of each candidate read, in file order
    sel = cand[inv]
    sel_rows, sel_inv = rows[sel], inv[sel]
    by_read = np.argsort(sel_inv, kind="stable")
    sel_rows, sel_inv = sel_rows[by_read], sel_inv[by_read]
    bounds = np.searchsorted(sel_inv, cand_ids, side="left"), np.searchsorted(sel_inv, cand_ids, side="right")

    chrom_of = sample.chrom_of
    fetch_ref = getattr(sample, "fetch_ref_view", None) or sample.fetch_ref
    signatures = []
    # the fields the per-read analysis reads, as Python values (one vectorised pass instead of NumPy scalar accesses per read)
    row_list = sel_rows.tolist()
    cols = dict(zip(row_list, zip(table.flag[sel_rows].tolist(), table.lead_clip[sel_rows].tolist(),
                                  table.trail_clip[sel_rows].tolist(), table.pos[sel_rows].tolist(),
                                  table.ref_span[sel_rows].tolist(), table.tid[sel_rows].tolist(),
                                  table.mapq[sel_rows].tolist(), table.l_seq[sel_rows].tolist())))
    names = table.names
    want_graph = getattr(options, "graph", False)
    for rid, lo, hi in zip(uniq[cand_ids].tolist(), bounds[0].tolist(), bounds[1].tolist()):
        primary = -1
        supp = []
        for a in row_list[lo:hi]:                            # the last non-supplementary record wins (:172-178)
            if cols[a][0] & 0x800:
                supp.append(a)
            else:
                primary = a
        if primary < 0:
            continue
        if cols[primary][7] == 0:
            # SEQ '*' on the primary: the reference slices None (analyze_reads.py:667) and the window fails
            raise TypeError("'NoneType' object is not subscriptable")
        qname = names[rid]
        majors, minors = analyze_between_aligns(primary, supp, table, options, sample, cols)
        segs = list(minors)
        for seg in majors:                                    # :201-216
            pieces, helpers = analyze_inside_align(seg, sample.gaps_of(seg.aln), options, sample)
            if pieces is None:
                segs.append(seg)
            else:
                segs.extend(pieces)
                segs.extend(helpers)
        segs.sort(key=by_read_pos)
        n = len(segs)
        if n < 2:
            continue

        whole_seq = table.query_sequence(primary) if want_graph else None
        ctx = (signatures, want_graph, whole_seq, chrom_of, fetch_ref, sample, options, qname)
        if n == 2:
            _emit(ctx, segs[0], segs[1], (), True)
            continue
        if segs[0].is_reverse:                                # :250-261
            _emit(ctx, segs[0], segs[1], (), True)
        if segs[-1].is_reverse:                               # :263-274
            _emit(ctx, segs[-2], segs[-1], (), True)
        main_idx = [i for i, s in enumerate(segs) if s.type == "main"]
        for p in range(len(main_idx) - 1):                    # :287-308
            i, j = main_idx[p], main_idx[p + 1]
            if segs[j].q_start - segs[i].q_end >= -25:
                _emit(ctx, segs[i], segs[j], segs[i + 1:j], p == len(main_idx) - 2)
    return signatures
