"""Per-read alignment analysis: split-read segments, in-CIGAR long gaps, gap typing.

Host-side mirror of the reference's ``src/collection/analyze_reads.py``:
``analyze_between_aligns`` :619-801, ``analyze_inside_align`` :804-970 (its CIGAR
walk :828-853 runs on the GPU -- ``svx_cigar_scan`` -- and only the handful of
long gaps per alignment reach this module), ``analyze_gap`` :155-615,
``trim_segs`` :82-152, ``shift_left`` :12-39, ``cal_overlap_ratio`` :49-80.
The ``--hash`` branches (:731-790, :898-929) re-align unmapped / inserted read pieces with
the k-mer aligner of :mod:`svision_amd.segmentplot` and add the hits as helper segments.
"""
from ..segmentplot.run_hash_lineplot import hashplot_unmapped
from .classes import _TABLE, Seg, Signature, by_read_pos


def shift_left(ref_seq, ref_start, target_start, target_end):
    """Left-align the interval [target_start, target_end) inside ``ref_seq`` (which
    starts at ``ref_start``) while the base before it equals its last base (:12-39)."""
    rel_s = target_start - ref_start
    rel_e = target_end - ref_start
    n = len(ref_seq)
    if rel_s >= n or rel_e >= n:
        return target_start, target_end
    k = 0
    while target_start - ref_start > 0 and ref_seq[rel_s - k - 1] == ref_seq[rel_e - k]:
        k += 1
        target_start -= 1
        target_end -= 1
    return target_start, target_end


def cal_overlap_ratio(base, target, left_most, right_most):
    """Fraction of ``base``'s reference span covered by ``target`` (:49-80)."""
    if base.same_value(target):                               # `==` on dicts upstream (:53): a duplicate record counts as itself
        return 0
    if base.ref_start < left_most or base.ref_end > right_most:
        return 1.0
    span = base.ref_end - base.ref_start
    if base.ref_start >= target.ref_start and base.ref_end <= target.ref_end:
        return 1.0
    if base.ref_end >= target.ref_end > base.ref_start and target.ref_start < base.ref_start:
        return (target.ref_end - base.ref_start) / span
    if base.ref_end < target.ref_start < base.ref_start and target.ref_end > base.ref_end:
        return (base.ref_end - target.ref_start) / span
    return 0


def trim_segs(segs, first, last):
    """Clamp / extend segments to +-2 x gap around the breakpoint, in place (:82-152)."""
    gap = max(last.q_start - first.q_end, last.ref_start - first.ref_end)
    left_most = first.ref_end - gap * 2
    right_most = last.ref_start + gap * 2
    for seg in segs:
        if seg.same_value(first):                             # dict `==` upstream (:102,120,126)
            if seg.ref_start < left_most:
                seg.q_start += left_most - seg.ref_start
                seg.ref_start = left_most
            elif seg.ref_start > left_most:
                grow = seg.ref_start - left_most
                seg.ref_start = left_most
                seg.q_end += grow
                for other in segs:
                    if not other.same_value(first):
                        other.q_start += grow
                        other.q_end += grow
        elif seg.same_value(last):
            if seg.ref_end > right_most:
                seg.q_end -= seg.ref_end - right_most
                seg.ref_end = right_most
            elif seg.ref_end < right_most:
                seg.q_end += right_most - seg.ref_end
                seg.ref_end = right_most
        else:
            length = seg.q_end - seg.q_start
            if seg.ref_start < left_most:
                seg.ref_start = left_most
                seg.ref_end = left_most + length
            if seg.ref_end > right_most:
                seg.ref_end = right_most
                seg.ref_start = right_most - length


def _among(seg, others):
    """``seg in others`` with the by-value comparison of segments."""
    for h in others:
        if seg.same_value(h):
            return True
    return False


def _signature(chrom, qname, sig_type, first_bkp, segs, helpers, trim_first, trim_last, mechanism="None",
               extend_end=0):
    """Shared tail of every analyze_gap branch: breakpoints of the helper segments in read
    order, extreme coordinates, trim, Signature."""
    bkps = [first_bkp]
    left, right = first_bkp[0], first_bkp[1]
    for seg in segs:
        if _among(seg, helpers):                              # `align in help_aligns` (:225): by value
            bkps.append([seg.ref_start, seg.ref_end, seg.ref_end - seg.ref_start])
            if seg.ref_start < left:
                left = seg.ref_start
            if seg.ref_end > right:
                right = seg.ref_end
    trim_segs(segs, trim_first, trim_last)
    return Signature(chrom, left, right + extend_end, sig_type, qname, segs, bkps, mechanism)


def _gap_bkp(anchor_end, next_start, length_if_open, length_if_closed=1):
    """[[start, end, len]] of the main breakpoint: closed (next starts at/before the anchor
    end) -> one-base interval."""
    if next_start <= anchor_end:
        return [anchor_end, anchor_end + 1, length_if_closed]
    return [anchor_end, next_start, length_if_open]


def analyze_gap(cur, nxt, chrom_of, fetch_ref, options, qname, help_segs=()):
    """Type the junction between two major segments ``cur`` -> ``nxt`` of one read, with the
    segments lying between them on the read as helpers (:155-615).  Returns a Signature or
    None.  ``cur``/``nxt`` are private copies; helper segments are mutated in place, as in
    the reference."""
    if cur.ref_id != nxt.ref_id:
        return None
    helpers = list(help_segs)
    covered = list(helpers)
    chrom = chrom_of(cur.ref_id)
    min_sv, max_sv = options.min_sv_size, options.max_sv_size

    if cur.is_reverse == nxt.is_reverse:
        lo = min(cur.ref_start, cur.ref_end, nxt.ref_start, nxt.ref_end)
        hi = max(cur.ref_start, cur.ref_end, nxt.ref_start, nxt.ref_end)
        if lo < 0:
            # the reference fetches [lo, hi) here whether it is needed or not (:182) and pysam refuses a negative start;
            # run_detect then drops the whole window (run_collection.py:44-47)
            raise ValueError("start out of range (%d)" % lo)
        ref_seq = None
        for seg in helpers:                                   # :184-193 forward helpers are left-shifted
            if not seg.is_reverse:
                if ref_seq is None:
                    ref_seq = fetch_ref(chrom, lo, hi)
                seg.ref_start, seg.ref_end = shift_left(ref_seq, lo, seg.ref_start, seg.ref_end)
        d_read = nxt.q_start - cur.q_end
        d_ref = nxt.ref_start - cur.ref_end

        if d_ref >= -min_sv:
            diff = d_read - d_ref
            if diff >= min_sv:                                # INS :207-246
                covered += [cur, nxt]
                segs = sorted(covered, key=by_read_pos)
                bkp = _gap_bkp(cur.ref_end, nxt.ref_start, abs(d_read), abs(d_read) + abs(d_ref))
                return _signature(chrom, qname, "sigGap", bkp, segs, helpers, cur, nxt,
                                  extend_end=diff if not helpers else 0)
            if -max_sv <= diff <= -min_sv:                    # DEL :248-315
                if ref_seq is None:
                    ref_seq = fetch_ref(chrom, lo, hi)
                s, e = shift_left(ref_seq, lo, cur.ref_end, nxt.ref_start)
                cur.ref_end = s + 1
                nxt.ref_start = e
                covered += [cur, nxt]
                segs = sorted(covered, key=by_read_pos)
                bkp = _gap_bkp(cur.ref_end, nxt.ref_start, nxt.ref_start - cur.ref_end)
                if helpers:
                    mechanism = "None"
                elif d_read > 10:
                    mechanism = "MMBIR+%d" % d_read
                elif d_read >= -2:
                    mechanism = ("NHEJ+%d" if d_read >= 0 else "NHEJ%d") % d_read
                elif d_read >= -20:
                    mechanism = "AltEJ%d" % d_read
                else:
                    mechanism = "NAHR%d" % d_read
                return _signature(chrom, qname, "sigGap", bkp, segs, helpers, cur, nxt, mechanism)
            # neither: only a junction bridged by helper segments is reported (INV-like) :317-352
            if helpers:
                covered += [cur, nxt]
                segs = sorted(covered, key=by_read_pos)
                bkp = _gap_bkp(cur.ref_end, nxt.ref_start, nxt.ref_start - cur.ref_end)
                sig = _signature(chrom, qname, "sigGap", bkp, segs, helpers, cur, nxt)
                if nxt.ref_start - cur.ref_end > 0:
                    return sig
            return None

        # reference overlap beyond min_sv: duplication :355-424
        dup_len = abs(d_ref)
        dup = Seg(nxt.q_start, nxt.q_start + dup_len, nxt.ref_start, nxt.ref_start + dup_len,
                  cur.ref_id, cur.is_reverse, qual=cur.qual)
        rest = Seg(nxt.q_start + dup_len + 1, nxt.q_end, nxt.ref_start + dup_len + 1, nxt.ref_end,
                   cur.ref_id, cur.is_reverse, qual=cur.qual)
        if rest.q_end < rest.q_start:
            rest.q_end = dup.q_end + dup_len
            rest.ref_end = dup.ref_end + dup_len
        covered += [cur, dup, rest]
        segs = sorted(covered, key=by_read_pos)
        blen = abs(d_read) + abs(d_ref)
        bkp = _gap_bkp(cur.ref_end, rest.ref_start, blen, blen)
        return _signature(chrom, qname, "sigDup", bkp, segs, helpers + [dup], cur, rest)

    # opposite strands: only when nothing lies between them; a forward stand-in segment is
    # synthesised so that the pair can be drawn :427-615
    covered += [cur, nxt]
    if helpers:
        return None
    if not cur.is_reverse:                                    # forward then reverse
        new_len = cur.q_end - cur.q_start
        if nxt.ref_end <= cur.ref_end:
            added = Seg(nxt.q_end, nxt.q_end + new_len, cur.ref_end, cur.ref_end + new_len,
                        cur.ref_id, cur.is_reverse, qual=cur.qual)
        else:
            fixed = max(nxt.ref_end - cur.ref_end, nxt.q_end - cur.q_end)
            added = Seg(cur.q_end + fixed, cur.q_end + fixed + new_len, nxt.ref_end, nxt.ref_end + new_len,
                        cur.ref_id, cur.is_reverse, qual=cur.qual)
        covered.append(added)
        segs = sorted(covered, key=by_read_pos)
        bkp = _gap_bkp(cur.ref_end, added.ref_start, added.ref_start - cur.ref_end)
        return _signature(chrom, qname, "sigUncovered", bkp, segs, [nxt], cur, added)
    # reverse then forward
    new_len = nxt.q_end - nxt.q_start
    if cur.ref_start >= nxt.ref_start:
        added = Seg(0, new_len, nxt.ref_start - new_len, nxt.ref_start - 1, cur.ref_id, nxt.is_reverse, qual=cur.qual)
        shift = new_len
    else:
        fixed = max(nxt.ref_start - cur.ref_start, nxt.q_start - cur.q_start)
        added = Seg(0, new_len, nxt.ref_start - fixed - new_len, nxt.ref_start - fixed - 1,
                    cur.ref_id, nxt.is_reverse, qual=cur.qual)
        shift = new_len + abs((nxt.ref_start - cur.ref_start) - (nxt.q_start - cur.q_start))
    for seg in covered:
        seg.q_start += shift
        seg.q_end += shift
    covered.append(added)
    segs = sorted(covered, key=by_read_pos)
    # added.ref_end < nxt.ref_start always, so the reference's malformed closed-gap list (:545,593) is unreachable
    assert nxt.ref_start > added.ref_end
    bkp = [added.ref_end, nxt.ref_start, nxt.ref_start - added.ref_end]
    return _signature(chrom, qname, "sigUncovered", bkp, segs, [cur], added, nxt)


def _hash_hits_to_segs(hits, read_offset, ref_offset, like, read_seq):
    """Aligner hits (piece / window coordinates) -> helper segments in read / reference coordinates
    (:757-773, :910-929)."""
    out = []
    for h in hits:
        fwd = h.forward()
        q0 = (h.xStart() if fwd else h.xEnd()) + read_offset
        q1 = (h.xEnd() if fwd else h.xStart()) + read_offset
        seg = Seg(q0, q1, h.yStart() + ref_offset, h.yEnd() + ref_offset, like.ref_id, not fwd,
                  like.is_supplementary, "other", like.qual, -1)
        seg.read_seq = read_seq
        out.append(seg)
    return out


def _fields(table, a):
    """(flag, leading clip, trailing clip, pos, reference span, tid, mapq, l_seq) of record ``a`` as Python ints."""
    return (int(table.flag[a]), int(table.lead_clip[a]), int(table.trail_clip[a]), int(table.pos[a]), int(table.ref_span[a]),
            int(table.tid[a]), int(table.mapq[a]), int(table.l_seq[a]))


def analyze_between_aligns(primary, supplementary, table, options, sample=None, cols=None):
    """Primary + supplementary alignments of one read -> (major, minor) segments (:619-801).

    ``primary``/``supplementary`` are record indices into ``table`` (an AlignmentTable with the
    device scan attached).  Query coordinates are expressed on the primary's strand.  ``cols``: {record index:
    :func:`_fields` tuple} prepared by the caller for the records of a whole window."""
    if not options.contig and len(supplementary) > 4:
        return [], []
    _TABLE[0] = table
    first = cols[primary] if cols is not None else _fields(table, primary)
    p_rev = bool(first[0] & 0x10)
    qlen = first[7]                                        # supplementary records inherit the primary's SEQ
    majors, minors, same_strand = [], [], []
    keep_seq = options.hash or getattr(options, "graph", False)           # the bases of every segment (--hash re-aligns them, --graph prints them)
    whole_seq = table.query_sequence(primary) if keep_seq else None
    for a in [primary] + list(supplementary):
        flag, lead, trail, r0, span, tid, mapq, _l = cols[a] if cols is not None else _fields(table, a)
        a_rev = bool(flag & 0x10)
        if a_rev != p_rev:
            q_start, q_end = trail, qlen - lead              # qlen - query_alignment_end, qlen - query_alignment_start
        else:
            q_start, q_end = lead, qlen - trail
        seg = Seg(q_start, q_end, r0, r0 + span, tid, a_rev != p_rev, bool(flag & 0x800), qual=mapq, aln=int(a))
        if keep_seq:
            seg.read_seq = whole_seq[q_start:q_end]           # :667 (TypeError on SEQ '*', as upstream)
        if seg.is_reverse:
            seg.type = "other"
            minors.append(seg)
        else:
            same_strand.append(seg)
    if len(same_strand) == 1:
        same_strand[0].type = "main"
        return same_strand, minors                            # (:685-691 returns before the --hash block)
    ordered = sorted(same_strand, key=by_read_pos)
    left_most, right_most = ordered[0].ref_start, ordered[0].ref_end
    for base in ordered:
        if base.ref_start < left_most:
            left_most = base.ref_start
        if base.ref_end > right_most:
            right_most = base.ref_end
    last = len(ordered) - 1
    for i, base in enumerate(ordered):
        covered = False
        if 0 < i < last:
            for target in ordered:
                if cal_overlap_ratio(base, target, left_most, right_most) >= 0.8:
                    covered = True
                    break
        if covered:
            base.type = "other"
            minors.append(base)
        else:
            base.type = "main"
            majors.append(base)
    if options.hash:
        _hash_between(majors, minors, options, sample)
    return majors, minors


def _hash_between(majors, minors, options, sample):
    """--hash: re-align the unmapped read piece between two adjacent main segments (:731-790).
    Upstream indexes the sorted segment list with the loop counter rather than with the main
    segment's own position (:744-745) and slices the segment's own bases with whole-read
    coordinates (:761-763); both are kept."""
    ordered = sorted(majors + minors, key=by_read_pos)
    main_idx = [i for i, s in enumerate(ordered) if s.type == "main"]
    for i in range(len(main_idx) - 1):
        if main_idx[i + 1] - main_idx[i] != 1:
            continue
        cur, nxt = ordered[i], ordered[i + 1]
        d_read = nxt.q_start - cur.q_end
        if d_read < options.min_sv_size:
            continue
        d_ref = nxt.ref_start - cur.ref_end
        if not (d_ref >= -options.min_sv_size and abs(d_read - d_ref) >= options.min_sv_size):
            continue
        read_start, read_end = cur.q_end, nxt.q_start
        piece = cur.read_seq[read_start:read_end]
        ref_start = min(cur.ref_start, nxt.ref_start)
        ref_end = max(cur.ref_end, nxt.ref_end)
        ref_seq = sample.fetch_ref_str(sample.chrom_of(cur.ref_id), ref_start, ref_end)
        if len(piece) < options.max_hash_len:
            _m, hits = hashplot_unmapped(ref_seq, piece, options.k_size, options.min_accept)
            minors.extend(_hash_hits_to_segs(hits, read_start, ref_start, cur, piece))


def _piece(out, seg, q0, q1, r0, r1):
    """A main segment cut out of alignment ``seg`` between two of its long gaps (:932-948)."""
    new = Seg(q0, q1, r0, r1, seg.ref_id, False, seg.is_supplementary, "main", seg.qual, seg.aln, derived=True)
    if seg.read_seq is not None:                              # :943 the piece's own bases (read by --graph)
        new.read_seq = seg.read_seq[q0 - seg.q_start:q1 - seg.q_start]
    out.append(new)


def analyze_inside_align(seg, gaps, options=None, sample=None):
    """Split one major segment at its long CIGAR gaps (:857-948).  ``gaps`` are this
    alignment's SvxGap records (kind, read_pos, ref_pos, len in op order) from the device scan;
    returns (new major segments, helper segments from --hash) or (None, None) when the alignment
    holds no long gap."""
    if len(gaps) == 0:
        return None, None
    out = []
    vrp = seg.q_start
    rows = gaps.tolist()                                   # (aln, op, read_pos, ref_pos, len, kind) tuples of kernels.GAP_DTYPE
    first_ref = rows[0][3]
    m = first_ref - seg.ref_start
    _piece(out, seg, vrp, vrp + m, seg.ref_start, first_ref - 1)
    vrp += m
    prev_end = None
    for _aln, _op, _read_pos, ref_pos, length, kind in rows:
        if prev_end is not None:
            m = ref_pos - prev_end
            _piece(out, seg, vrp + 1, vrp + m + 1, prev_end, ref_pos)
            vrp += m
        if kind == 1:
            vrp += length
            prev_end = ref_pos
        else:
            prev_end = ref_pos + length
    m = seg.ref_end - prev_end
    _piece(out, seg, vrp + 1, vrp + m + 1, prev_end, seg.ref_end)
    helpers = []
    if options is not None and options.hash:                   # :898-929 re-align every long insertion
        ref_seq = None
        for g in gaps:
            if int(g["kind"]) != 1:
                continue
            read_pos, length = int(g["read_pos"]), int(g["len"])
            ins = seg.read_seq[read_pos - seg.q_start:read_pos + length - seg.q_start]
            if ref_seq is None or True:                        # upstream re-fetches per insertion (same span)
                ref_seq = sample.fetch_ref_str(sample.chrom_of(seg.ref_id), seg.ref_start, seg.ref_end)
            if len(ins) < options.max_hash_len:
                _m, hits = hashplot_unmapped(ref_seq, ins, options.k_size, options.min_accept)
                hs = _hash_hits_to_segs(hits, read_pos, seg.ref_start, seg, "")
                helpers.extend(hs)
    return out, helpers
