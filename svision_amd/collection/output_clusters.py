"""Candidate sites -> segment-pair records (the encode -> CNN boundary).

Host-side mirror of the reference's ``src/collection/output_clusters.py``:
``writer_cluster_to_file`` :30-89, ``proc_one_cluster`` :93-123, ``proc_one_sig``
:125-216, ``linearOrNot`` :11-26, ``cal_non_linear`` :218-249.  Besides the
reference's 23-column TSV text, every pair is also kept as a 12-int record so the
rasteriser can consume a packed int32 array without re-parsing text.
"""
import os

from .graph import write_cluster_graphs

from ..segmentplot.classes import cord_to_segments


def linearOrNot(a, b):
    """True when two segments continue each other: same strand and 0.7 < dRef/dRead < 1.5 (:11-26)."""
    d_ref = b.yStart() - a.yEnd()
    d_read = b.xStart() - a.xEnd()
    if d_read == 0:
        d_read = 1
    if a.forward() != b.forward():
        return False
    ratio = d_ref / d_read
    return not (ratio >= 1.5 or ratio <= 0.7)


def cal_non_linear(segs):
    """Length-weighted distance from the diagonal, normalised by the reference span (:218-249)."""
    ys, total = [], 0
    for s in segs:
        ys.append(s.yStart())
        ys.append(s.yEnd())
        total += abs((s.xStart() + s.xEnd()) / 2 - (s.yStart() + s.yEnd()) / 2) * s.length()
    span = max(ys) - min(ys)
    if span == 0:
        return -1
    return int(total / span)


class PairLine:
    """One TSV line = one similarity image."""
    __slots__ = ("region", "seg1", "seg2", "read_len", "ref_len", "tag", "sub", "qname", "sig_type",
                 "bkp", "score", "forward", "mechanism")

    def __init__(self, region, seg1, seg2, read_len, ref_len, tag, sub, qname, sig_type, bkp, score, forward, mechanism):
        self.region, self.seg1, self.seg2 = region, seg1, seg2
        self.read_len, self.ref_len, self.tag, self.sub = read_len, ref_len, tag, sub
        self.qname, self.sig_type, self.bkp, self.score = qname, sig_type, bkp, score
        self.forward, self.mechanism = forward, mechanism

    def record(self):
        """12 ints: TSV columns 1..12 (svx_rasterize input)."""
        return self.seg1.fields() + self.seg2.fields() + (self.read_len, self.ref_len)

    def label(self):
        """The label string BatchGenerator derives from the TSV line (create_batch.py:45-49)."""
        return "svision".join([self.tag, self.region, self.qname, self.sig_type, str(self.bkp[0]), str(self.bkp[1]),
                               str(self.score), self.forward, self.mechanism, str(self.bkp[2])])

    def text(self):
        return "\t".join([self.region, self.seg1.toString(), self.seg2.toString(), str(self.read_len), str(self.ref_len),
                          self.tag, str(self.sub), self.qname, self.sig_type, str(self.bkp[0]), str(self.bkp[1]),
                          str(self.score), self.forward, self.mechanism, str(self.bkp[2])]) + "\n"


def proc_one_sig(cluster_region, sig, sig_cnt, options=None):
    """All non-collinear segment pairs of one signature (:125-216); -1 when it has no extent."""
    ref_len, read_len, main_cords, other_cords = sig.get_segs_cords()
    mains = cord_to_segments(main_cords)
    others = cord_to_segments(other_cords)
    score = cal_non_linear(mains + others)
    if score == -1:
        return -1
    lines = []
    sub = 0
    for a, b in zip(mains[:-1], mains[1:]):                   # main x main (:176-182)
        sub += 1
        if not linearOrNot(a, b):
            lines.append(PairLine(cluster_region, a, b, read_len, ref_len, "%dm" % sig_cnt, sub, sig.qname, sig.type,
                                  sig.bkps[0], score, "True", sig.mechanism))
    for a in mains:                                           # main x other (:189-209)
        for i, b in enumerate(others):
            sub += 1
            if not linearOrNot(a, b):
                fwd = "False" if (a.forward() is False or b.forward() is False) else "True"
                lines.append(PairLine(cluster_region, a, b, read_len, ref_len, str(sig_cnt), sub, sig.qname, sig.type,
                                      sig.bkps[i + 1], score, fwd, sig.mechanism))
    return lines


def proc_one_cluster(cluster, options=None):
    region = cluster.region()
    lines = []
    for cnt, sig in enumerate(cluster.get_signatures(), start=1):
        got = proc_one_sig(region, sig, cnt, options)
        if got != -1:
            lines.extend(got)
    return cluster, lines


def iter_pair_lines(clusters, options):
    """Site filter of writer_cluster_to_file (:46-51) + per-cluster pair extraction, one list of PairLines per kept
    cluster, in cluster order (the streaming pipeline hands a window's lines on while later clusters are still worked on)."""
    for cl in clusters:
        if int(cl.cend) - int(cl.cstart) > options.max_sv_size:
            continue
        if cl.read_num >= options.min_support:
            yield proc_one_cluster(cl, options)[1]
            if getattr(options, "graph", False) is True:      # :57-67
                write_cluster_graphs(cl, options)


def collect_pair_lines(clusters, options):
    """All PairLines of a window's clusters, in TSV order."""
    out = []
    for lines in iter_pair_lines(clusters, options):
        out.extend(lines)
    return out


def writer_cluster_to_file(clusters, chrom, part_num, options):
    """Write ``segments/{chrom}.segments.{part}.bed`` (:84-89); returns the PairLines."""
    lines = collect_pair_lines(clusters, options)
    path = os.path.join(options.out_path, "segments", "%s.segments.%s.bed" % (chrom, part_num))
    with open(path, "w") as f:
        for ln in lines:
            f.write(ln.text())
    return lines
