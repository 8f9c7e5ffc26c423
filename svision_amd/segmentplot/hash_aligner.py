"""k-mer seed-and-extend re-aligner for unmapped / inserted read pieces (``--hash``).

Mirror of the reference's ``HashAligner`` (src/segmentplot/hash_aligner.py:7-398): exact k-mer seeds
(k = 10) of the reference window looked up in a table of the piece's k-mers on both strands,
ungapped extension until the first mismatch / 'N' / end, hits of at least ``windowSize`` bases kept
unless they coincide with a self-repeat of the reference window, greedy merge of collinear hits.
Quirks kept on purpose (they shape which duplications are reported):
the k-mer loops stop at ``len - (k + 1)`` (:150,162,176); the extension counts the mismatching base
and never reaches the last base (``>= len - 1``, :44-62); seeds whose previous base also matches are
skipped (:196,204); with ``repeat_thresh`` hits a reference k-mer is only recorded as "avoid".
"""
from .classes import Segment

_COMPLEMENT = {"A": "T", "T": "A", "C": "G", "G": "C"}


def reverse_complement(bases):
    """classes.py:21-39: anything that is not upper-case ACGT becomes 'N'."""
    return "".join(_COMPLEMENT.get(b, "N") for b in reversed(bases))


class HashAligner:
    def __init__(self, k, windowSize, mismatchNum, repeat_thresh):
        self.k, self.windowSize, self.mismatchNum, self.repeat_thresh = k, windowSize, mismatchNum, repeat_thresh
        self.segments = []
        self.selfDiffSegs = []
        self.compareDiffSegs = None
        self.avoid_kmers = []
        self.hashvalues = []
        self.y_hashvalues = None

    # -- accessors of the reference class ---------------------------------------------------------
    def getSegments(self): return self.segments
    def getSelfDiffSegs(self): return self.selfDiffSegs
    def getHashValues(self): return self.hashvalues
    def getAvoidKmer(self): return self.avoid_kmers

    def run(self, x_bases, y_bases, compareDiffSegs=None, y_hashvalue=None, avoid_kmers_from_ref=None):
        """x: the piece to place (or the window itself for the self pass), y: the reference window."""
        self.ref_length = len(y_bases)
        self.compareDiffSegs = compareDiffSegs
        self.y_hashvalues = y_hashvalue
        self._align(x_bases, y_bases, avoid_kmers_from_ref)

    def _extend(self, x, y, x_pos, y_pos):
        """Ungapped extension from a seed (:37-62 / :81-99): returns the match length."""
        n = self.k
        mismatch = 0
        x_last, y_last = len(x) - 1, len(y) - 1
        while mismatch <= self.mismatchNum:
            if x_pos + n >= x_last or y_pos + n >= y_last:
                break
            a, b = x[x_pos + n], y[y_pos + n]
            if a == "N" or b == "N":
                break
            if a != b:
                mismatch += 1
            n += 1
        return n

    def _keep(self, seg):
        if self.compareDiffSegs is None:                       # self pass: every hit, off-diagonal ones as "diff"
            self.segments.append(seg)
            if self._off_diagonal(seg):
                self.selfDiffSegs.append(seg)
        elif not self._explained_by_self_repeat(seg):
            self.segments.append(seg)

    def _seed(self, x, rx, y, positions, i):
        for pos in positions:
            if pos >= 0:
                if pos > 0 and i > 0 and x[pos - 1] == y[i - 1]:
                    continue                                   # already covered by the previous k-mer
                n = self._extend(x, y, pos, i)
                if n >= self.windowSize:
                    self._keep(Segment(pos, i, n, True, 0))
            else:
                rp = -1 - pos
                if rp > 0 and i > 0 and rx[rp - 1] == y[i - 1]:
                    continue
                n = self._extend(rx, y, rp, i)
                if n >= self.windowSize:
                    self._keep(Segment((len(rx) - 1) - rp, i, n, False, 0))

    def _align(self, x, y, avoid_kmers_from_ref):
        k = self.k
        rx = reverse_complement(x)
        table = {}
        for i in range(0, len(x) - (k + 1)):
            table.setdefault(x[i:i + k], []).append(i)
        for i in range(0, len(rx) - (k + 1)):
            table.setdefault(rx[i:i + k], []).append(-1 - i)
        if self.y_hashvalues is None:
            self.hashvalues = []
            for i in range(0, len(y) - (k + 1)):
                kmer = y[i:i + k]
                self.hashvalues.append(kmer)
                hits = table.get(kmer)
                if hits is None:
                    continue
                if len(hits) >= self.repeat_thresh:
                    self.avoid_kmers.append(kmer)
                else:
                    self._seed(x, rx, y, hits, i)
        else:
            avoid = set(avoid_kmers_from_ref)                  # the reference tests list membership
            for i, kmer in enumerate(self.y_hashvalues):
                hits = table.get(kmer)
                if hits is not None and kmer not in avoid:
                    self._seed(x, rx, y, hits, i)

    # -- merge of collinear hits (:241-293) ----------------------------------------------------------
    def getMergeSegments(self):
        segs = self.segments
        cur = 1
        while cur < len(segs):
            s = segs[cur]
            merged = False
            for cand in segs[:cur]:
                if self.linearOrNot(cand, s):
                    if s.forward() is True:
                        cand.setxEnd(max(s.xEnd(), cand.xEnd()))
                    elif s.forward() is False:
                        cand.setxEnd(min(s.xEnd(), cand.xEnd()))
                    cand.setyEnd(max(s.yEnd(), cand.yEnd()))
                    cand.setLength(abs(cand.length()) + abs(s.xEnd() - cand.xEnd()))
                    segs.remove(s)
                    merged = True
                    break
            if not merged:
                cur += 1
        self.segments = [s for s in segs if (s.yEnd() - s.yStart()) >= 20]
        return self.segments

    def linearOrNot(self, i, j):
        """:296-328: same strand, start-offset ratio in [0.8, 1.2], not too far apart, merged slope ~ +-1."""
        if i.forward() != j.forward():
            return False
        dy = abs(float(i.yStart() - j.yStart()))
        diff = 5 if dy == 0 else abs(float(i.xStart() - j.xStart())) / dy
        if diff > 1.2 or diff < 0.8:
            return False
        max_dis = (i.length() + j.length()) * 1.5
        if abs(i.xEnd() - j.xStart()) > max_dis and abs(i.yEnd() - j.yStart()) > max_dis:
            return False
        run = float(j.xEnd() - i.xStart())
        if run == 0:
            run = 0.0001
        slope = float(j.yEnd() - i.yStart()) / run
        return not abs(abs(slope) - 1) > 0.2

    def _explained_by_self_repeat(self, seg):
        """compareWithDiffSegs (:331-349)."""
        for d in self.compareDiffSegs:
            if (abs(seg.yStart() - d.yStart()) <= 5 and seg.yEnd() <= d.yEnd()) \
                    or (abs(seg.yEnd() - d.yEnd()) <= 5 and seg.yStart() >= d.yStart()):
                return True
        return False

    @staticmethod
    def _off_diagonal(seg):
        """calDiffForRef (:351-363)."""
        diff2 = float(seg.xEnd()) / float(seg.yEnd())
        diff3 = (float(seg.xStart() + seg.xEnd()) / 2.0) / (float(seg.yStart() + seg.yEnd()) / 2.0)
        return diff2 != 1 or diff3 != 1
