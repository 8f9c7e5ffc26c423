"""ORACLE (test infrastructure, not product): per-alignment CIGAR / segment scan.

CPU restatement of ``analyze_inside_align``
(/root/reference/src/collection/analyze_reads.py:804-970, default path without
``--hash``) plus the alignment-level quantities the reference obtains from
pysam (``reference_end``, ``query_alignment_start/end``; SURVEY 8(a')).

CIGAR ops use BAM codes: M=0 I=1 D=2 N=3 S=4 H=5 P=6 '='=7 X=8.  The reference
rewrites H to S before parsing (collect_signatures.py:91), so H behaves as S.
"""
import re

OPS = 'MIDNSHP=X'
_CIG_RE = re.compile(r'(\d+)([MIDNSHP=X])')


def parse_cigar(cigar):
    """CIGAR text -> [(op_code, length)]."""
    return [(OPS.index(o), int(n)) for n, o in _CIG_RE.findall(cigar)]


def pack_cigar(ops):
    """[(op, len)] -> list of BAM u32 words (len << 4 | op)."""
    return [(n << 4) | o for o, n in ops]


def scan_long_gaps(ops, ref_start, min_sv):
    """analyze_reads.py:828-853: walk the CIGAR, return long gaps as
    (op_index, kind, read_pos, ref_pos, length) with kind 1='I', 2='D'."""
    read_pos, ref_pos = 0, ref_start
    gaps = []
    for i, (op, n) in enumerate(ops):
        if op in (3, 4, 5):            # N, S (and H rewritten to S): read only  (:831-832)
            read_pos += n
        elif op == 1:                  # I (:834-839)
            if n >= min_sv:
                gaps.append((i, 1, read_pos, ref_pos, n))
            read_pos += n
        elif op == 2:                  # D (:841-844)
            if n >= min_sv:
                gaps.append((i, 2, read_pos, ref_pos, n))
            ref_pos += n
        elif op in (0, 7, 8):          # M = X (:846-848)
            ref_pos += n
            read_pos += n
    return gaps


def alignment_stats(ops):
    """pysam-derived per-alignment quantities (SURVEY 8(a')):
    (ref_span, lead_clip, trail_clip, query_len) where reference_end =
    pos + ref_span (M,D,N,=,X), query_alignment_start = lead_clip (leading S/H),
    query_alignment_end = query_len - trail_clip, query_len = sum(M,I,S,H,=,X)."""
    ref_span = sum(n for o, n in ops if o in (0, 2, 3, 7, 8))
    qlen = sum(n for o, n in ops if o in (0, 1, 4, 5, 7, 8))
    lead = 0
    for o, n in ops:
        if o in (4, 5):
            lead += n
        else:
            break
    trail = 0
    for o, n in reversed(ops):
        if o in (4, 5):
            trail += n
        else:
            break
    if lead == qlen:                  # all-clip CIGAR: do not double count
        trail = 0
    return ref_span, lead, trail, qlen


def major_segments(gaps, q_start, ref_start, ref_end):
    """analyze_reads.py:857-895: major segments [q0, q1, r0, r1] between long
    gaps, from a *virtual* read position that advances by the reference span of
    each piece (+ insertion lengths).  Returns None when there is no long gap."""
    if not gaps:
        return None
    segs = []
    vrp = q_start
    first_ref = gaps[0][3]
    m = first_ref - ref_start
    segs.append([vrp, vrp + m, ref_start, first_ref - 1])
    vrp += m
    for g, nxt in zip(gaps[:-1], gaps[1:]):
        if g[1] == 1:
            vrp += g[4]
        cur_ref_end = g[3] + (g[4] if g[1] == 2 else 0)
        m = nxt[3] - cur_ref_end
        segs.append([vrp + 1, vrp + m + 1, cur_ref_end, nxt[3]])
        vrp += m
    g = gaps[-1]
    if g[1] == 1:
        vrp += g[4]
    last_ref_end = g[3] + (g[4] if g[1] == 2 else 0)
    m = ref_end - last_ref_end
    segs.append([vrp + 1, vrp + m + 1, last_ref_end, ref_end])
    return segs


def inside_align(cigar_ops, q_start, ref_start, ref_end, min_sv):
    """Full analyze_inside_align restatement: segments or None."""
    return major_segments(scan_long_gaps(cigar_ops, ref_start, min_sv), q_start, ref_start, ref_end)
