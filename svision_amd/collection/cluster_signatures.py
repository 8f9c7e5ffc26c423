"""Partition + hierarchical clustering of signatures into candidate SV sites.

Mirror of the reference's ``src/collection/cluster_signatures.py``:
``signature_partition`` :51-66, ``cluster_partitions`` :68-130,
``span_position_distance`` :132-141.  The O(n^2) Python-callback ``pdist`` of the
reference is replaced by the same IEEE-double arithmetic: on the device
(``svx_span_position_distance``, the condensed matrices of all partitions of a window in
one launch) in the process that owns the GPU, with NumPy on the condensed index pairs in
the forked host helpers (which never touch the GPU); both are bit-identical to the
callback.  The condensed matrix then goes through the very same SciPy
``linkage(method="average")`` / ``fcluster(criterion="distance")`` calls, so merge order,
labels and therefore site coordinates are unchanged.
"""
import logging

import numpy as np
from scipy.cluster.hierarchy import fcluster, linkage

from .classes import Cluster

try:                                                          # SciPy's own compiled cores, without the per-call validation wrappers
    from scipy.cluster import _hierarchy as _hc
    from scipy.cluster.hierarchy import _LINKAGE_METHODS
    _AVERAGE = _LINKAGE_METHODS["average"]
    if not (hasattr(_hc, "nn_chain") and hasattr(_hc, "cluster_dist")):
        _hc = None
except Exception:                                             # noqa: BLE001 -- another SciPy layout: the public functions below
    _hc = None


try:
    import cython as _cython
    _COMPILED = bool(_cython.compiled)
except ImportError:                                           # interpreted and no Cython on the machine
    _COMPILED = False
_NAN, _INF = float("nan"), float("inf")


def average_linkage_labels(dist, n, threshold):
    """fcluster(linkage(dist, "average"), threshold, "distance") for a condensed matrix of ``n`` observations.  The same two
    compiled routines the public functions end in (scipy/cluster/hierarchy.py: ``_hierarchy.nn_chain`` for method "average",
    ``_hierarchy.cluster_dist`` for criterion "distance"), called directly: a window clusters ~75 small partitions and the
    wrappers' array-API plumbing and ``is_valid_linkage`` passes cost more than the clustering (≈25 of 90 ms per window in
    the build container's profile).  Anything unusual -- non-finite distances, another SciPy -- takes the public path, which
    raises what upstream would see."""
    if _hc is None or not np.isfinite(dist).all() or dist.size != n * (n - 1) // 2:
        return fcluster(linkage(dist, method="average"), threshold, criterion="distance")
    z = _hc.nn_chain(np.ascontiguousarray(dist, np.float64), int(n), _AVERAGE)
    labels = np.zeros(n, dtype="i")
    _hc.cluster_dist(np.asarray(z), labels, float(threshold), int(n))
    return labels


def partition_and_cluster(signatures, chrom, sample, options):
    partitions = signature_partition(signatures, options)
    return cluster_partitions(partitions, chrom, sample, options)


def signature_partition(signatures, options):
    """Greedy split of the key-sorted signatures (:51-66).  A partition is only closed
    once it holds MORE than min_support signatures, and only such partitions are kept."""
    ordered = sorted(signatures, key=lambda s: s.get_key())
    partitions, cur = [], []
    for sig in ordered:
        if len(cur) > options.min_support and cur[-1].position_distance_to(sig) > options.patition_max_distance:
            partitions.append(cur)
            cur = []
        cur.append(sig)
    if len(cur) > options.min_support:
        partitions.append(cur)
    return partitions


def span_position_distance_condensed(starts, ends, normalizer=1000):
    """Condensed pairwise matrix of span_position_distance (:132-141) in pdist order."""
    s = np.ascontiguousarray(starts, np.float64)              # (columns of a 2-D array come in strided)
    e = np.ascontiguousarray(ends, np.float64)
    if _COMPILED:                                             # the same IEEE operations pair by pair, as C loops (cluster_signatures.pxd)
        out = np.empty(len(s) * (len(s) - 1) // 2, np.float64)
        _condensed_loops(s, e, float(normalizer), out)
        return out
    n = len(starts)
    i, j = np.triu_indices(n, k=1)
    span = e - s
    centre = np.floor_divide(s + e, 2)
    pos = np.minimum(np.minimum(np.abs(s[i] - s[j]), np.abs(e[i] - e[j])), np.abs(centre[i] - centre[j])) / normalizer
    with np.errstate(invalid="ignore", divide="ignore"):
        spd = np.abs(span[i] - span[j]) / np.maximum(span[i], span[j])
    return pos + spd


def _condensed_loops(s, e, normalizer, out):
    """out[k] = span_position_distance of the k-th pair (i < j, row-major): min(|ds|, |de|, |dcentre|) / normalizer +
    |dspan| / max(span).  A zero ``max(span)`` gives what IEEE division gives NumPy: nan for 0 / 0, inf otherwise."""
    n = s.shape[0]
    k = 0
    for i in range(n):
        si, ei = s[i], e[i]
        spi = ei - si
        ci = (si + ei) // 2.0
        for j in range(i + 1, n):
            sj, ej = s[j], e[j]
            spj = ej - sj
            cj = (sj + ej) // 2.0
            pos = abs(si - sj)
            d = abs(ei - ej)
            if d < pos:
                pos = d
            d = abs(ci - cj)
            if d < pos:
                pos = d
            num = abs(spi - spj)
            den = spi if spi > spj else spj
            if den != 0.0:
                spd = num / den
            elif num == 0.0:
                spd = _NAN
            else:
                spd = _INF
            out[k] = pos / normalizer + spd
            k += 1


def condensed_distances(parts, sample):
    """[condensed span_position_distance matrix (float64) of every partition in ``parts``] -- one device launch for
    all of them when the sample lives on a GPU in this process, NumPy otherwise."""
    if not parts:
        return []
    if getattr(sample, "device_buffers", None) is not None:
        import torch
        from .. import kernels
        dev = sample.device_buffers[0].device
        sizes = np.array([len(p) for p in parts], np.int64)
        off = np.zeros(sizes.size + 1, np.int64)
        off[1:] = np.cumsum(sizes)
        starts = torch.tensor([s.tstart for p in parts for s in p], dtype=torch.float64, device=dev)
        ends = torch.tensor([s.tend for p in parts for s in p], dtype=torch.float64, device=dev)
        out, out_off = kernels.span_position_distance(starts, ends, off)
        host = out.cpu().numpy()
        return [host[int(out_off[i]):int(out_off[i + 1])] for i in range(len(parts))]
    return [span_position_distance_condensed([s.tstart for s in p], [s.tend for s in p]) for p in parts]


def cluster_partitions(partitions, chrom, sample, options):
    clusters = []
    kept = []
    for part in partitions:
        if len(part) > 100000:                                # :80-85
            logging.warning("Partition size large than 100,000, ranging from %s:%d-%d", chrom, part[0].tstart, part[-1].tstart)
            continue
        kept.append(part)
    dists = iter(condensed_distances([p for p in kept if len(p) > 1], sample))
    for part in kept:
        if len(part) == 1:
            groups = [part]
        else:
            labels = average_linkage_labels(next(dists), len(part), options.cluster_max_distance)
            groups = [[] for _ in range(int(labels.max()))]
            for sig, lab in zip(part, labels):
                groups[lab - 1].append(sig)
        for sigs in groups:
            cl = Cluster(sigs)
            if cl.abandon == 0:
                clusters.append(cl)
    if clusters:                                              # coverage: one vectorised pass (classes.py:165-170)
        tid = sample.table.get_tid(clusters[0].contig)
        cov = sample.table.count_overlaps(tid, [int(c.cstart) for c in clusters], [int(c.cend) for c in clusters])
        for c, v in zip(clusters, cov):
            c.coverage = int(v)
    return sorted(clusters, key=lambda c: (c.contig, (c.cstart + c.cend) / 2))
