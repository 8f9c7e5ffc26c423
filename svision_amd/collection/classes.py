"""Signature / Cluster / segment records of the collection step.

Host-side mirror of the reference's ``src/collection/classes.py`` (Signature
:7-117, Cluster :122-177) and of the segment dicts built in
``src/collection/analyze_reads.py:658-668,932-948``.  A segment is a small
mutable object instead of a dict; the reference's dict-equality tests
(``seg == first_seg``, ``align in help_aligns``) only ever hold for the same
object (help/main/synthetic segments differ in ``type`` or strand), so identity
comparison is used.
"""


_TABLE = [None]                   # AlignmentTable the `aln` indices point into (set by analyze_between_aligns)
BY_VALUE = [True]                 # tests/test_dup_golden.py switches it off to show that its fixture needs the by-value comparison


class Seg:
    """One aligned piece of a read in read/reference coordinates.  Compiled as an extension type (classes.pxd: C integer
    fields); the same source runs interpreted when the host build is absent."""
    __slots__ = ("q_start", "q_end", "ref_start", "ref_end", "ref_id", "is_reverse",
                 "is_supplementary", "type", "qual", "aln", "read_seq", "derived")

    def __init__(self, q_start, q_end, ref_start, ref_end, ref_id, is_reverse,
                 is_supplementary=False, type=None, qual=0, aln=-1, derived=False):
        self.q_start, self.q_end = q_start, q_end
        self.ref_start, self.ref_end = ref_start, ref_end
        self.ref_id = ref_id
        self.is_reverse = is_reverse
        self.is_supplementary = is_supplementary
        self.type = type
        self.qual = qual
        self.aln = aln            # index of the source alignment in the AlignmentTable (-1: synthetic)
        self.read_seq = None      # bases of this piece of the read (--hash only)
        self.derived = derived    # piece cut out of an alignment (upstream: cigarstring '') rather than a whole record

    def copy(self):
        c = Seg(self.q_start, self.q_end, self.ref_start, self.ref_end, self.ref_id, self.is_reverse,
                self.is_supplementary, self.type, self.qual, self.aln, self.derived)
        c.read_seq = self.read_seq
        return c

    def same_value(self, o):
        """Upstream segments are dicts and are compared BY VALUE (`==`, `in`: analyze_reads.py:53,102,126,225,...), which
        matters when a BAM holds the same record twice: all fields equal, including the `type` key being set or not
        (None here), the CIGAR string of whole records ('' for pieces cut out of one) and, implicitly, the read."""
        if self is o:
            return True
        if not BY_VALUE[0]:
            return False
        if (self.q_start, self.q_end, self.ref_start, self.ref_end, self.ref_id, self.is_reverse, self.is_supplementary,
                self.type, self.qual, self.derived, self.aln < 0) != \
                (o.q_start, o.q_end, o.ref_start, o.ref_end, o.ref_id, o.is_reverse, o.is_supplementary,
                 o.type, o.qual, o.derived, o.aln < 0):
            return False
        if self.derived or self.aln < 0 or self.aln == o.aln:
            return True
        t = _TABLE[0]
        if t is None:
            return False
        a, b = t.cigar[t.cig_off[self.aln]:t.cig_off[self.aln + 1]], t.cigar[t.cig_off[o.aln]:t.cig_off[o.aln + 1]]
        return a.size == b.size and bool((a == b).all())

    def __repr__(self):
        return "Seg(q=%d-%d ref=%d-%d rev=%s %s)" % (self.q_start, self.q_end, self.ref_start, self.ref_end,
                                                    self.is_reverse, self.type)


def by_read_pos(seg):
    return (seg.q_start, seg.q_end)


class Signature:
    """Abnormal-alignment signature of one read (reference classes.py:7-117)."""

    def __init__(self, contig, tstart, tend, type, qname, sorted_aligns, all_bkps, mechanism):
        self.contig = contig
        self.tstart = tstart
        self.tend = tend
        self.qname = qname
        self.type = type
        self.bkps = all_bkps
        self.sorted_aligns = sorted_aligns
        self.mechanism = mechanism
        self.graph = None

    def get_source(self):
        return (self.contig, self.tstart, self.tend)

    def get_key(self):
        return (self.contig, (self.tstart + self.tend) // 2)

    def position_distance_to(self, other):
        """classes.py:35-44: min of start / end / centre distances (inf across contigs)."""
        if self.contig != other.contig:
            return float("inf")
        c1 = (self.tstart + self.tend) // 2
        c2 = (other.tstart + other.tend) // 2
        return min(abs(self.tstart - other.tstart), abs(self.tend - other.tend), abs(c1 - c2))

    def set_graph(self, graph):
        self.graph = graph

    def get_segs_cords(self):
        """classes.py:72-117: rebase every segment on the first one (in place) and split
        them into main (first, last) and other coordinate triples."""
        return _segs_cords(self.sorted_aligns)


def _segs_cords(segs):
    """Body of :meth:`Signature.get_segs_cords` (a module-level function so that the compiled module can type it)."""
    first = segs[0]
    q0, r0 = first.q_start, first.ref_start
    main, other = [], []
    last = len(segs) - 1
    i = 0
    for s in segs:
        s.ref_start -= r0
        s.ref_end -= r0
        s.q_start -= q0
        s.q_end -= q0
        if i == 0 or i == last:
            main.append([[s.q_start, s.q_end], [s.ref_start, s.ref_end], 0])
        elif s.is_reverse:
            other.append([[s.q_end, s.q_start], [s.ref_start, s.ref_end], 1])
        else:
            other.append([[s.q_start, s.q_end], [s.ref_start, s.ref_end], 0])
        i += 1
    return s.ref_end, s.q_end, main, other


class Cluster:
    """Candidate SV site: signatures merged by hierarchical clustering (classes.py:122-177).
    ``coverage`` (all records overlapping [int(cstart), int(cend))) is filled by the caller
    in one vectorised pass instead of a BAM re-open per cluster (classes.py:165-170)."""

    def __init__(self, sigs):
        self.sigs = self.signatures = sigs
        self.contig = sigs[0].contig
        self.read_num = len(sigs)
        self.cstart = sum(s.tstart for s in sigs) / len(sigs)
        self.cend = sum(s.tend for s in sigs) / len(sigs)
        self.abandon = 1 if (self.cstart < 0 or self.cend < 0 or self.cstart > self.cend) else 0
        self.coverage = 0

    def get_signatures(self):
        return self.signatures

    def region(self):
        """``chr+start+end+coverage`` site key (output_clusters.py:105-106)."""
        return "%s+%d+%d+%d" % (self.contig, int(self.cstart), int(self.cend), self.coverage)
