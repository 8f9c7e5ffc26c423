"""Seeded synthetic inputs shared by the CPU and GPU parity tests."""
import numpy as np


def random_records(n, seed=0, hostile=True):
    """[n,12] int32 segment-pair records: realistic SVision geometry plus (hostile)
    out-of-range, zero/negative-length and steep cases that exercise clipLine."""
    rng = np.random.default_rng(seed)
    rec = np.zeros((n, 12), np.int64)
    for i in range(n):
        mode = rng.integers(0, 6) if hostile else 0
        span = int(rng.choice([30, 200, 227, 228, 1000, 5000, 40000, 1000000]))
        read_len = int(rng.integers(1, span + 1))
        ref_len = int(rng.integers(1, span + 1))
        for k in (0, 5):
            fwd = int(rng.integers(0, 2)) if mode != 0 else int(rng.random() < 0.8)
            y0 = int(rng.integers(0, ref_len + 1))
            y1 = int(rng.integers(y0, ref_len + 2))
            x0 = int(rng.integers(0, read_len + 1))
            if mode == 1:      # far out of range on both axes
                x0 = int(rng.integers(-2 * span, 3 * span))
                y0 = int(rng.integers(-span, 2 * span)); y1 = y0 + int(rng.integers(0, 2 * span))
            elif mode == 2:    # degenerate lengths 0 / 1 / negative
                y1 = y0 + int(rng.integers(-2, 3))
            elif mode == 3:    # reverse segment running off the top
                fwd = 0; x0 = int(rng.integers(0, max(1, read_len // 4)))
            elif mode == 4:    # small image, ratio clamps to 1
                pass
            rec[i, k:k + 5] = (x0, x0 + (y1 - y0), y0, y1, fwd)
        if mode == 4:
            read_len = int(rng.integers(1, 227)); ref_len = int(rng.integers(1, 227))
        rec[i, 10], rec[i, 11] = read_len, ref_len
    return rec.astype(np.int32)


_OPS = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8])


def random_cigars(n_aln, seed=0, mean_ops=200, long_gap_rate=0.01, max_ops=None, lognormal_sigma=None):
    """Packed CIGAR words + CSR offsets + ref_start for n_aln synthetic alignments
    (HiFi-like: =/X/I/D soup, optional S/H clips, occasional long I/D, rare N/P)."""
    rng = np.random.default_rng(seed)
    if lognormal_sigma is None:
        n_ops = np.maximum(1, rng.poisson(mean_ops, n_aln)).astype(np.int64)
    else:                                                     # ONT-like: op count follows the read length (median mean_ops)
        n_ops = np.maximum(1, rng.lognormal(np.log(mean_ops), lognormal_sigma, n_aln)).astype(np.int64)
    if max_ops is not None:
        n_ops = np.minimum(n_ops, max_ops)
    n_ops[rng.random(n_aln) < 0.02] = 1
    off = np.zeros(n_aln + 1, np.uint64)
    off[1:] = np.cumsum(n_ops)
    total = int(off[-1])
    kind = rng.choice(_OPS, size=total, p=[0.25, 0.12, 0.12, 0.002, 0.0, 0.0, 0.003, 0.4, 0.105])
    lens = rng.integers(1, 40, total)
    small = (kind == 1) | (kind == 2)
    lens[small] = rng.integers(1, 12, int(small.sum()))
    big = small & (rng.random(total) < long_gap_rate)
    lens[big] = rng.integers(30, 5000, int(big.sum()))
    matches = (kind == 0) | (kind == 7)
    lens[matches] = rng.integers(1, 3000, int(matches.sum()))
    # clips at the ends of some alignments
    starts = off[:-1].astype(np.int64)
    ends = off[1:].astype(np.int64) - 1
    for idx, p in ((starts, 0.4), (ends, 0.4)):
        sel = idx[(rng.random(n_aln) < p) & (n_ops >= 3)]
        kind[sel] = rng.choice([4, 5], sel.size)
        lens[sel] = rng.integers(1, 20000, sel.size)
    sel2 = starts[(rng.random(n_aln) < 0.05) & (n_ops >= 5)]
    kind[sel2] = 5; kind[sel2 + 1] = 4                        # H then S
    cigar = ((lens.astype(np.uint64) << np.uint64(4)) | kind.astype(np.uint64)).astype(np.uint32)
    ref_start = rng.integers(0, 200_000_000, n_aln).astype(np.int32)
    return cigar, off, ref_start
