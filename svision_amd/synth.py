"""Synthetic long-read workloads (no real BAM ships with the reference; the demo
BAM is listed in /root/reference/.MISSING_LARGE_BLOBS).

Generates a random reference, plants structural variants on two haplotypes and
"sequences + aligns" reads analytically: every read is cut from a donor
haplotype, and its alignments (primary + supplementary, CIGARs with =/X/I/D
noise, long I/D for in-read SVs, S/H clips) are written directly as an
:class:`svision_amd.io.bam.AlignmentTable`, i.e. what an aligner would emit
(SURVEY 8(d) synthetic inputs).  Deterministic for a given seed.
"""
from dataclasses import dataclass, field

import numpy as np

from .io.bam import AlignmentTable, FLAG_REVERSE, FLAG_SUPPLEMENTARY

_OP = {"M": 0, "I": 1, "D": 2, "N": 3, "S": 4, "H": 5, "P": 6, "=": 7, "X": 8}


@dataclass
class SimConfig:
    contigs: list = field(default_factory=lambda: [("chr21", 46_709_983)])
    coverage: float = 30.0
    read_len_mean: float = 15_000.0
    read_len_sd: float = 2_000.0
    lognormal: bool = False          # ONT-like read lengths (read_len_mean = median)
    lognormal_sigma: float = 0.9
    err_rate: float = 0.005          # small events (X / 1-3 bp I / 1-3 bp D) per reference base
    sv_spacing: float = 110_000.0    # mean distance between planted SVs
    sv_min: int = 50
    sv_max: int = 10_000
    inline_max: int = 2_000          # SVs up to this size stay inside one alignment as I/D ops
    sv_min_gap: int = 25_000         # minimum distance between neighbouring planted SVs (+ 3 x size)
    het_frac: float = 0.5
    sv_mix: tuple = (("DEL", 0.45), ("INS", 0.40), ("INV", 0.05), ("DUP", 0.05), ("dDUP", 0.03), ("DELINV", 0.02))
    seed: int = 1


def make_genome(cfg):
    rng = np.random.default_rng(cfg.seed)
    lut = np.frombuffer(b"ACGT", np.uint8)
    return {name: lut[rng.integers(0, 4, length, dtype=np.uint8)].tobytes() for name, length in cfg.contigs}


def plant_svs(cfg):
    """-> {contig: [dict(type,pos,len,src,gt)]} sorted by pos, non-overlapping."""
    rng = np.random.default_rng(cfg.seed + 1)
    kinds = [k for k, _ in cfg.sv_mix]
    probs = np.array([p for _, p in cfg.sv_mix], float)
    probs /= probs.sum()
    out = {}
    for name, length in cfg.contigs:
        svs = []
        pos = 30_000 + int(rng.exponential(cfg.sv_spacing))
        while pos < length - 60_000:
            kind = str(rng.choice(kinds, p=probs))
            size = int(np.exp(rng.uniform(np.log(cfg.sv_min), np.log(cfg.sv_max))))
            gt = (1, 1) if rng.random() >= cfg.het_frac else ((1, 0) if rng.random() < 0.5 else (0, 1))
            sv = {"type": kind, "pos": pos, "len": size, "src": -1, "gt": gt}
            if kind == "dDUP":
                sv["src"] = pos - int(rng.integers(3 * size + 2_000, 3 * size + 12_000))
            elif kind in ("cINS", "rcINS"):        # insertion whose bases are a (reverse-complemented) nearby copy
                sv["src"] = pos - size - int(rng.integers(0, 1_500))
            svs.append(sv)
            pos += 3 * size + cfg.sv_min_gap + int(rng.exponential(cfg.sv_spacing))
        out[name] = svs
    return out


def _haplotype_pieces(length, svs, hap):
    """Donor haplotype as pieces (ref_lo, ref_hi, strand(+1/-1), novel) in donor order."""
    pieces = []
    c = 0
    for sv in svs:
        if not sv["gt"][hap]:
            continue
        p, n, t = sv["pos"], sv["len"], sv["type"]
        if t == "DEL":
            pieces.append((c, p, 1, 0)); c = p + n
        elif t == "INS":
            pieces.append((c, p, 1, 0)); pieces.append((0, n, 1, 1)); c = p
        elif t == "INV":
            pieces.append((c, p, 1, 0)); pieces.append((p, p + n, -1, 0)); c = p + n
        elif t == "DUP":
            pieces.append((c, p + n, 1, 0)); pieces.append((p, p + n, 1, 0)); c = p + n
        elif t == "dDUP":
            pieces.append((c, p, 1, 0)); pieces.append((sv["src"], sv["src"] + n, 1, 0)); c = p
        elif t in ("cINS", "rcINS"):
            pieces.append((c, p, 1, 0)); pieces.append((sv["src"], sv["src"] + n, 1 if t == "cINS" else -1, 2)); c = p
        elif t == "DELINV":
            d = n // 2
            pieces.append((c, p, 1, 0)); pieces.append((p + d, p + d + n, -1, 0)); c = p + d + n
    pieces.append((c, length, 1, 0))
    arr = np.asarray(pieces, np.int64)
    plen = arr[:, 1] - arr[:, 0]
    dstart = np.zeros(arr.shape[0] + 1, np.int64)
    dstart[1:] = np.cumsum(plen)
    return arr, dstart


def _noisy_block(rng, n, err_rate):
    """CIGAR (ops, lens) in reference order for an aligned block spanning n reference
    bases, with small X/I/D events; returns (ops, lens, read_len)."""
    k = int(rng.poisson(n * err_rate)) if n >= 16 else 0
    slots = (n - 8) // 4
    k = min(k, max(0, slots))
    if k == 0:
        return np.array([7], np.int64), np.array([n], np.int64), n
    pos = (np.sort(rng.choice(slots, k, replace=False)) + 1) * 4
    typ = rng.choice(3, k, p=[0.6, 0.2, 0.2])               # 0 X, 1 I, 2 D
    elen = np.where(typ == 0, 1, rng.integers(1, 4, k))
    rcons = np.where(typ == 1, 0, elen)
    prev_end = np.concatenate([[0], pos[:-1] + rcons[:-1]])
    runs = pos - prev_end
    ops = np.empty(2 * k + 1, np.int64)
    lens = np.empty(2 * k + 1, np.int64)
    ops[0::2] = 7
    lens[0:-1:2] = runs
    lens[-1] = n - (pos[-1] + rcons[-1])
    ops[1::2] = np.where(typ == 0, 8, typ)                   # X=8, I=1, D=2
    lens[1::2] = elen
    read_len = n - int(rcons.sum()) + int(elen[typ != 2].sum())
    return ops, lens, read_len


_COMP = np.zeros(256, np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def _revcomp(arr):
    return _COMP[arr[::-1]]


def _block_bases(rng, ref_bases, ops, lens):
    """Read bases (reference order) of one aligned block given its noisy CIGAR."""
    out = []
    r = 0
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for op, n in zip(ops.tolist(), lens.tolist()):
        if op == 7:                                   # =
            out.append(ref_bases[r:r + n]); r += n
        elif op == 8:                                 # X: any other base
            sub = ref_bases[r:r + n].copy()
            for j in range(n):
                sub[j] = acgt[(int(np.searchsorted(acgt, sub[j])) + int(rng.integers(1, 4))) % 4]
            out.append(sub); r += n
        elif op == 1:                                 # I
            out.append(acgt[rng.integers(0, 4, n)])
        elif op == 2:                                 # D
            r += n
    return np.concatenate(out) if out else np.empty(0, np.uint8)


def simulate(cfg, with_genome=True, with_seq=False, svs=None):
    """-> (AlignmentTable, genome dict or None, svs dict).  ``with_seq``: also synthesise the read bases
    (primary records carry SEQ, as minimap2 writes them; needed for --hash only)."""
    rng = np.random.default_rng(cfg.seed + 2)
    svs = plant_svs(cfg) if svs is None else svs          # svs: {contig: [dict(type,pos,len,src,gt)]} planted by hand
    genome = make_genome(cfg) if (with_genome or with_seq) else None
    recs = []          # (tid, pos, flag, mapq, l_seq, name_id, ops, lens[, seq])
    names = []
    for tid, (cname, clen) in enumerate(cfg.contigs):
        for hap in (0, 1):
            pieces, dstart = _haplotype_pieces(clen, svs[cname], hap)
            dlen = int(dstart[-1])
            n_reads = int(cfg.coverage / 2.0 * clen / cfg.read_len_mean)
            if cfg.lognormal:
                lens_r = np.exp(rng.normal(np.log(cfg.read_len_mean), cfg.lognormal_sigma, n_reads))
            else:
                lens_r = rng.normal(cfg.read_len_mean, cfg.read_len_sd, n_reads)
            lens_r = np.clip(lens_r, 1_000, dlen // 2).astype(np.int64)
            starts = (rng.random(n_reads) * (dlen - lens_r)).astype(np.int64)
            rev = rng.random(n_reads) < 0.5
            first_piece = np.searchsorted(dstart, starts, side="right") - 1
            last_piece = np.searchsorted(dstart, starts + lens_r, side="left") - 1
            for r in range(n_reads):
                a, b = int(starts[r]), int(starts[r] + lens_r[r])
                blocks = []    # ('M', ref_lo, ref_hi, strand) / ('N', len)
                for pi in range(int(first_piece[r]), int(last_piece[r]) + 1):
                    lo, hi, strand, novel = (int(v) for v in pieces[pi])
                    d0, d1 = int(dstart[pi]), int(dstart[pi + 1])
                    s, e = max(a, d0), min(b, d1)
                    if e <= s:
                        continue
                    if novel == 2:             # unaligned insertion carrying a copy of ref[lo:hi] (strand sign)
                        blocks.append(("N", e - s, lo + (s - d0) if strand > 0 else hi - (e - d0), strand))
                    elif novel:
                        blocks.append(("N", e - s))
                    elif strand > 0:
                        blocks.append(("M", lo + (s - d0), lo + (e - d0), 1))
                    else:                      # donor walks the piece right-to-left
                        blocks.append(("M", hi - (e - d0), hi - (s - d0), -1))
                name_id = len(names)
                names.append("%s_h%d_r%d" % (cname, hap, r))
                _emit_read(rng, cfg, tid, blocks, bool(rev[r]), name_id, recs,
                           np.frombuffer(genome[cname], np.uint8) if with_seq else None)
    recs.sort(key=lambda t: (t[0], t[1]))
    n = len(recs)
    tid = np.fromiter((t[0] for t in recs), np.int32, n)
    pos = np.fromiter((t[1] for t in recs), np.int32, n)
    flag = np.fromiter((t[2] for t in recs), np.uint16, n)
    mapq = np.fromiter((t[3] for t in recs), np.uint8, n)
    l_seq = np.fromiter((t[4] for t in recs), np.int32, n)
    nid = np.fromiter((t[5] for t in recs), np.int32, n)
    n_ops = np.fromiter((t[6].size for t in recs), np.int64, n)
    cig_off = np.zeros(n + 1, np.int64)
    cig_off[1:] = np.cumsum(n_ops)
    ops = np.concatenate([t[6] for t in recs]) if n else np.empty(0, np.int64)
    lens = np.concatenate([t[7] for t in recs]) if n else np.empty(0, np.int64)
    cigar = ((lens.astype(np.uint64) << np.uint64(4)) | ops.astype(np.uint64)).astype(np.uint32)
    # QNAME ids must follow first occurrence in file order (as a BAM reader would assign them)
    remap = {}
    new_names = []
    for i in range(n):
        j = int(nid[i])
        if j not in remap:
            remap[j] = len(new_names)
            new_names.append(names[j])
        nid[i] = remap[j]
    refs = [c for c, _ in cfg.contigs]
    lens_c = [l for _, l in cfg.contigs]
    seq_packed = seq_off = None
    if with_seq:
        from .io.bam import pack_sequence
        chunks, seq_off, o = [], np.zeros(n, np.int64), 0
        for i, t in enumerate(recs):
            seq_off[i] = o
            if t[8] is not None:
                pk = pack_sequence(t[8].tobytes())
                chunks.append(pk)
                o += len(pk)
        seq_packed = b"".join(chunks)
    return AlignmentTable(refs, lens_c, tid, pos, flag, mapq, l_seq, nid, new_names, cigar, cig_off,
                          seq_packed=seq_packed, seq_off=seq_off), genome, svs


def _emit_read(rng, cfg, tid, blocks, read_rev, name_id, recs, ref_bases=None):
    """Group a read's blocks into alignments and append BAM-like records."""
    # 1. noisy CIGAR per aligned block, read extents
    items = []      # [kind, ref_lo, ref_hi, strand, ops, lens, read_len]
    pieces = []     # read bases per block, in donor-forward read orientation
    for blk in blocks:
        if blk[0] == "N":
            items.append(["N", 0, 0, 0, None, None, blk[1]])
            if ref_bases is not None:
                if len(blk) == 4:              # copy insertion
                    src = ref_bases[blk[2]:blk[2] + blk[1]]
                    pieces.append(src.copy() if blk[3] > 0 else _revcomp(src))
                else:
                    pieces.append(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, blk[1])])
        else:
            ops, lens, rl = _noisy_block(rng, blk[2] - blk[1], cfg.err_rate)
            items.append(["M", blk[1], blk[2], blk[3], ops, lens, rl])
            if ref_bases is not None:
                bases = _block_bases(rng, ref_bases[blk[1]:blk[2]], ops, lens)
                pieces.append(bases if blk[3] > 0 else _revcomp(bases))
    read_fwd = np.concatenate(pieces) if (ref_bases is not None and pieces) else None
    total = sum(it[6] for it in items)
    # 2. merge collinear forward blocks separated by small gaps into one alignment
    alns = []       # [q_lo, q_hi, strand, ref_lo, ops, lens]
    q = 0
    i = 0
    while i < len(items):
        it = items[i]
        if it[0] == "N":
            q += it[6]; i += 1
            continue
        q_lo, ops, lens = q, [it[4]], [it[5]]
        ref_lo, ref_hi, strand = it[1], it[2], it[3]
        q += it[6]; i += 1
        while strand > 0 and i < len(items):
            ins = 0
            j = i
            if items[j][0] == "N" and items[j][6] <= cfg.inline_max and j + 1 < len(items):
                ins = items[j][6]; j += 1
            nxt = items[j]
            if nxt[0] != "M" or nxt[3] <= 0:
                break
            gap = nxt[1] - ref_hi
            if gap < 0 or gap > cfg.inline_max or (gap == 0 and ins == 0):
                break
            if ins:
                ops.append(np.array([1], np.int64)); lens.append(np.array([ins], np.int64)); q += ins
            if gap:
                ops.append(np.array([2], np.int64)); lens.append(np.array([gap], np.int64))
            ops.append(nxt[4]); lens.append(nxt[5])
            ref_hi = nxt[2]; q += nxt[6]; i = j + 1
        alns.append([q_lo, q, strand, ref_lo, np.concatenate(ops), np.concatenate(lens)])
    if not alns:
        return
    primary = max(range(len(alns)), key=lambda k: alns[k][1] - alns[k][0])
    for k, (q_lo, q_hi, strand, ref_lo, ops, lens) in enumerate(alns):
        if read_rev:                               # coordinates on the read as sequenced
            q_lo, q_hi = total - q_hi, total - q_lo
        bam_rev = (strand < 0) != read_rev
        lead, trail = (total - q_hi, q_lo) if bam_rev else (q_lo, total - q_hi)
        clip = 4 if k == primary else 5            # S on the primary, H on supplementary (minimap2 style)
        pre = ([clip], [lead]) if lead else ([], [])
        post = ([clip], [trail]) if trail else ([], [])
        o = np.concatenate([np.asarray(pre[0], np.int64), ops, np.asarray(post[0], np.int64)])
        l = np.concatenate([np.asarray(pre[1], np.int64), lens, np.asarray(post[1], np.int64)])
        flag = (FLAG_REVERSE if bam_rev else 0) | (0 if k == primary else FLAG_SUPPLEMENTARY)
        seq = None
        if read_fwd is not None and k == primary:
            as_sequenced = _revcomp(read_fwd) if read_rev else read_fwd
            seq = _revcomp(as_sequenced) if bam_rev else as_sequenced      # BAM stores SEQ on the reference strand
        recs.append((tid, ref_lo, flag, 60, total if k == primary else 0, name_id, o, l, seq))
