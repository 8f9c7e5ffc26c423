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


DEVICE = None        # torch.device of the process that owns a GPU (set by Sample.from_table); None in the forked host helpers


def hashplot_unmapped(ref, seq, k, min_accept):
    """-> (None, segments): self-align the window to learn its repeats, then place ``seq`` (:52-85).
    In the GPU-owning process the two seed-and-extend passes run on the device (``svx_hash_seeds``); the hit lists
    are replayed through the same order-dependent host steps.  Sequences outside the alphabet ACGTN (upstream's
    k-mers are raw strings), pieces longer than the kernel's table, k > 13 and overflowing hit lists take the host
    passes below, which are also what the helper processes (no GPU) run."""
    if DEVICE is not None:
        got = hashplot_unmapped_batch([(ref, seq)], k, min_accept, DEVICE)[0]
        if got is not None:
            return None, got
    return None, _hashplot_host(ref, seq, k, min_accept)


def _hashplot_host(ref, seq, k, min_accept):
    repeat_thresh = 2
    self_pass = HashAligner(k, min_accept, 0, repeat_thresh)
    self_pass.run(ref, ref)
    placer = HashAligner(k, min_accept, 0, repeat_thresh)
    placer.run(seq, ref, self_pass.getSelfDiffSegs(), self_pass.getHashValues(), self_pass.getAvoidKmer())
    merged = placer.getMergeSegments()
    if len(merged) >= 2:
        merged = select_longest(merged)
    return merged


def hashplot_unmapped_batch(pairs, k, min_accept, device):
    """[(ref, seq), ...] -> [segments or None]: every pair's seed-and-extend passes in ONE device launch.
    None = the pair cannot go through the device kernel (see :func:`hashplot_unmapped`)."""
    from .. import kernels
    out = [None] * len(pairs)
    jobs, where = [], []
    if not (2 <= k <= 13):
        return out
    for n, (ref, seq) in enumerate(pairs):
        if len(seq) > kernels.HASH_MAX_X:
            continue
        x, y = kernels.pack_bases(seq), kernels.pack_bases(ref)
        if x is None or y is None:
            continue
        jobs.append((x, y))
        where.append(n)
    for n, res, (x, y) in zip(where, kernels.hash_seeds(jobs, k, min_accept, device), jobs):
        if res is None:
            continue
        hits_a, hits_b = res
        self_pass = HashAligner(k, min_accept, 0, 2)
        self_pass.compareDiffSegs = None
        for i, pos, length, fwd in hits_a.tolist():          # the self pass: its off-diagonal hits are the window's repeats
            self_pass._keep(Segment(pos, i, length, True, 0) if fwd else Segment((len(y) - 1) - pos, i, length, False, 0))
        placer = HashAligner(k, min_accept, 0, 2)
        placer.compareDiffSegs = self_pass.getSelfDiffSegs()
        for i, pos, length, fwd in hits_b.tolist():
            placer._keep(Segment(pos, i, length, True, 0) if fwd else Segment((len(x) - 1) - pos, i, length, False, 0))
        merged = placer.getMergeSegments()
        if len(merged) >= 2:
            merged = select_longest(merged)
        out[n] = merged
    return out
