"""Both ingest engines on BAM + BAI files written by tests/htslike.py -- a writer that shares no code with the product's own
(struct + zlib, htslib's conventions: flush-before-record blocks of 0xFF00 bytes with records larger than a block
straddling, or blocks cut wherever 0xFF00 bytes end; EOF block; reg2bin; the 37450 pseudo-bin; back-filled linear index;
n_no_coor; aux tags of every type with a multi-KB MM / ML pair; SEQ = *; a CG-tag CIGAR; a multi-sample header; deflate levels
1, 6 and 9).  What the readers return is compared with what the test itself put in, field for field
(/root/reference/src/collection/collect_signatures.py:131-155 reads these fields through pysam)."""
import os

import numpy as np
import pytest

from svision_amd.io import bam
from tests import htslike

REFS = [("chrA", 400_000), ("chrEmpty", 50_000), ("chrB", 250_000), ("chrC", 120_000)]


def _records(seed=5):
    rng = np.random.default_rng(seed)
    recs = []
    serial = 0

    def cigar_for(n_ops, clip=True):
        ops = []
        if clip and rng.random() < 0.5:
            ops.append((int(rng.integers(1, 400)), "S" if rng.random() < 0.7 else "H"))
        for k in range(n_ops):
            op = "M" if k % 2 == 0 else "IDNX=P"[int(rng.integers(0, 6))]
            ops.append((int(rng.integers(1, 60 if op != "N" else 2000)), op))
        if clip and rng.random() < 0.5:
            ops.append((int(rng.integers(1, 400)), "S"))
        return ops

    def tags_for(i):
        t = [("NM", "i", int(rng.integers(0, 1 << 20))), ("AS", "C", int(rng.integers(0, 255))), ("XA", "A", "q"), ("Xc", "c", -5),
             ("Xs", "s", -3000), ("XS", "S", 60000), ("XI", "I", 4000000000), ("Xf", "f", 1.5), ("RG", "Z", "rg%d" % (1 + i % 2)),
             ("XH", "H", "1AE301")]
        if i % 3 == 0:
            t += [("Bc", "Bc", [-1, 2, -3]), ("BC", "BC", list(range(40))), ("Bs", "Bs", [-300, 300]), ("BS", "BS", [65535]),
                  ("Bi", "Bi", [-70000, 70000]), ("BI", "BI", [1, 2, 3]), ("Bf", "Bf", [0.25, -0.5])]
        if i % 11 == 0:                                            # base modifications: a multi-KB MM:Z / ML:B,C pair
            n = 3000
            t += [("MM", "Z", "C+m," + ",".join(str(int(v)) for v in rng.integers(0, 9, n)) + ";"), ("ML", "BC", [int(v) for v in rng.integers(0, 256, n)])]
        return t
    for tid, (_name, length) in enumerate(REFS):
        if tid == 1:
            continue
        n = {0: 140, 2: 90, 3: 40}[tid]
        pos = np.sort(rng.integers(0, length - 60_000, n))
        pos[:3] = pos[0]                                            # equal positions: order of the file must be kept
        for j, p in enumerate(pos):
            i = serial
            serial += 1
            n_ops = int(rng.integers(1, 40))
            flag = int(rng.choice([0, 16, 2048, 2064]))
            rec = dict(tid=tid, pos=int(p), qname="read/%05d" % (i - (1 if flag & 2048 and i else 0)), flag=flag, mapq=int(rng.integers(0, 61)),
                       cigar=cigar_for(n_ops), tags=tags_for(i))
            if tid == 0 and j == 20:                               # a CIGAR of more than 65,535 operations: CG tag
                rec["cigar"] = [(1, "M"), (1, "I")] * 35_000 + [(7, "M")]
                rec["flag"] = 0
            if tid == 2 and j == 10:                               # a record larger than a BGZF block: straddles under every policy
                rec["cigar"] = [(70_000, "M")]
            if j % 13 == 5:                                        # secondary alignment, SEQ and QUAL absent
                rec["flag"] = 256 | (flag & 16)
                rec["seq"], rec["qual"] = "*", None
            elif j % 17 == 3:                                      # unmapped read placed with its mate: no CIGAR
                rec["flag"], rec["cigar"], rec["mapq"] = 4 | 1 | 8, [], 0
                rec["seq"], rec["qual"] = "".join("ACGT"[v] for v in rng.integers(0, 4, 150)), bytes(rng.integers(0, 42, 150).astype(np.uint8))
            else:
                lq = htslike.query_len([c for c in rec["cigar"] if c[1] != "H"])
                rec["seq"] = "".join("ACGTN"[v] for v in rng.integers(0, 5, lq))
                rec["qual"] = bytes(rng.integers(0, 42, lq).astype(np.uint8)) if j % 7 else None   # QUAL absent: 0xFF bytes
            recs.append(rec)
    for k in range(9):                                             # the unmapped reads behind the last reference
        recs.append(dict(tid=-1, pos=-1, qname="unmapped/%d" % k, flag=4, mapq=0, cigar=[], seq="ACGT" * 30, qual=bytes([30] * 120), tags=[("RG", "Z", "rg1")]))
    return recs


def _expected(recs, tid):
    mine = [r for r in recs if r["tid"] == tid]
    words = [np.asarray([n << 4 | htslike.OPS.index(op) for n, op in r["cigar"]], np.uint32) for r in mine]
    off = np.zeros(len(mine) + 1, np.int64)
    off[1:] = np.cumsum([len(w) for w in words])
    return dict(pos=np.asarray([r["pos"] for r in mine], np.int32), flag=np.asarray([r["flag"] for r in mine], np.uint16),
                mapq=np.asarray([r["mapq"] for r in mine], np.uint8), l_seq=np.asarray([0 if r["seq"] == "*" else len(r["seq"]) for r in mine], np.int32),
                cigar=np.concatenate(words) if words else np.zeros(0, np.uint32), cig_off=off, qnames=[r["qname"] for r in mine])


def _check(table, want, what):
    assert len(table) == len(want["pos"]), what
    for f in ("pos", "flag", "mapq", "l_seq", "cig_off"):
        assert np.array_equal(np.asarray(getattr(table, f)), want[f]), (what, f)
    assert np.array_equal(np.asarray(table.cigar[:]), want["cigar"]), (what, "cigar")
    assert [table.names[i] for i in table.name_id] == want["qnames"], (what, "qname")


CASES = [(1, "htslib"), (6, "htslib"), (9, "stream"), (6, "stream")]


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("htslike")
    recs = _records()
    out = {}
    for level, policy in CASES:
        path = str(d / ("l%d_%s.bam" % (level, policy)))
        out[(level, policy)] = (path, htslike.write_bam(path, REFS, recs, level=level, policy=policy))
    return recs, out


@pytest.mark.parametrize("case", CASES)
def test_host_readers_on_a_foreign_bam(files, case):
    recs, out = files
    path, voffs = out[case]
    head = bam.read_bam_header(path)
    assert head.references == [n for n, _l in REFS] and head.lengths == [l for _n, l in REFS]
    assert "SM:sampleB" in head.header_text and head.header_text.startswith("@HD")
    whole = bam.read_bam(path)                                    # no index: the whole file, unmapped tail included
    assert int((np.asarray(whole.tid) < 0).sum()) == 9
    for tid in (0, 2, 3):
        want = _expected(recs, tid)
        _check(whole.subset(np.flatnonzero(np.asarray(whole.tid) == tid)), want, "whole file, tid %d" % tid)
        _check(bam.read_bam(path, tids=[tid]), want, "byte range of tid %d through the .bai" % tid)
    parts = list(bam.BamStream(path, threads=3, tids=[0, 1, 2, 3], index=path + ".bai"))
    assert [int(p.tid[0]) for p in parts] == [0, 2, 3]
    for p in parts:
        _check(p, _expected(recs, int(p.tid[0])), "stream part")
    # the index itself: spans and the back-filled linear index are the record offsets the writer reports
    lin = bam.read_bai_linear(path + ".bai")
    assert lin[1] is None
    first = {t: next(i for i, r in enumerate(recs) if r["tid"] == t) for t in (0, 2, 3)}
    last = {t: max(i for i, r in enumerate(recs) if r["tid"] == t) for t in (0, 2, 3)}
    for t in (0, 2, 3):
        lo, hi, linear = lin[t]
        assert lo == voffs[first[t]] and hi == voffs[last[t] + 1]
        assert set(int(v) for v in linear) <= set(voffs)          # every linear entry is a record's own start


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_device_engine_on_a_foreign_bam(files, case):
    """The device decoder (svx_bgzf_inflate + svx_bgzf_crc32 + svx_bam_walk_*) on the same files: the tables the test put in,
    and the packed CIGARs it leaves in HBM are the same words."""
    import svision_amd.ingest_gpu as ig
    recs, out = files
    path, _voffs = out[case]
    head = bam.read_bam_header(path)
    saved = ig.FIRST_GROUP_BYTES, ig.PIPE_GROUP_BYTES
    try:
        for first, later in ((1 << 10, 1 << 10), (1 << 40, 1 << 40)):      # every chromosome its own launch / all in one
            ig.FIRST_GROUP_BYTES, ig.PIPE_GROUP_BYTES = first, later
            dec = ig.DeviceDecoder(path, path + ".bai", head.references, head.lengths, head.header_text, "cuda:0", threads=3)
            assert dec.usable([0, 1, 2, 3])
            got = []
            for finish, (d_cigar, d_off, d_pos) in dec.parts_pipelined([0, 1, 2, 3]):
                tb = finish()
                ig.spill_cigar(tb)
                tid = int(tb.tid[0])
                got.append(tid)
                want = _expected(recs, tid)
                _check(tb, want, "device engine, tid %d" % tid)
                assert np.array_equal(d_off.cpu().numpy(), want["cig_off"]) and np.array_equal(d_pos.cpu().numpy(), want["pos"])
                assert np.array_equal(d_cigar.cpu().numpy().view(np.uint32)[:want["cigar"].size], want["cigar"])
            assert got == [0, 2, 3]
    finally:
        ig.FIRST_GROUP_BYTES, ig.PIPE_GROUP_BYTES = saved
