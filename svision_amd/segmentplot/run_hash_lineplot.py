"""``hashplot_unmapped``: place an unmapped / inserted read piece on its local reference window.

Mirror of the reference's src/segmentplot/run_hash_lineplot.py (``hashplot_unmapped`` :52-85,
``select_longest`` :8-33).  ``cord_to_segments`` lives in :mod:`.classes`.
"""
from .classes import Segment, cord_to_segments  # noqa: F401  (re-exported like upstream)
from .hash_aligner import HashAligner


def select_longest(segments):
    """Longest hit(s) per strand, forward ones first (:8-33)."""
    best = {True: [], False: []}
    for seg in segments:
        bucket = best[seg.forward() == True]      # noqa: E712  (None counts as reverse, as upstream)
        span = abs(seg.xEnd() - seg.xStart())
        if not bucket or span > abs(bucket[0].xEnd() - bucket[0].xStart()):
            bucket[:] = [seg]
        elif span == abs(bucket[0].xEnd() - bucket[0].xStart()):
            bucket.append(seg)
    return best[True] + best[False]


def hashplot_unmapped(ref, seq, k, min_accept):
    """-> (None, segments): self-align the window to learn its repeats, then place ``seq`` (:52-85)."""
    repeat_thresh = 2
    self_pass = HashAligner(k, min_accept, 0, repeat_thresh)
    self_pass.run(ref, ref)
    placer = HashAligner(k, min_accept, 0, repeat_thresh)
    placer.run(seq, ref, self_pass.getSelfDiffSegs(), self_pass.getHashValues(), self_pass.getAvoidKmer())
    merged = placer.getMergeSegments()
    if len(merged) >= 2:
        merged = select_longest(merged)
    return None, merged
