"""Unit-level parity vs randomised cases pushed through the reference's own functions
(tests/golden/make_fuzz_fixture.py): analyze_gap (every branch, with helpers and left-shift on
repeats), analyze_inside_align, refine_type, the per-site vote, linearOrNot / cal_non_linear."""
import copy
import json
import os
import collections

import numpy as np
import pytest

from oracle import cigar_ref
from svision_amd.collection import analyze_reads as ar
from svision_amd.collection import output_clusters as oc
from svision_amd.collection.classes import Seg
from svision_amd.network.output import refine_type
from svision_amd.network.predict import Predict
from svision_amd.segmentplot.classes import Segment
from tests import helpers


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(helpers.GOLDEN, "fuzz_small.expected.json")) as f:
        return json.load(f)


def _seg(v, typ):
    return Seg(v[0], v[1], v[2], v[3], v[5], v[4], False, typ, 60)


def test_analyze_gap_fuzz(fx):
    ref, ref_start = fx["ref"].encode(), fx["ref_start"]
    fetch = lambda chrom, s, e: ref[max(0, s - ref_start):max(0, e - ref_start)]
    opts = helpers.default_options()
    kinds = collections.Counter()
    for case in fx["gap"]:
        cur, nxt = _seg(case["cur"], "main"), _seg(case["nxt"], "main")
        helps = [_seg(h, "other") for h in case["help"]]
        try:
            sig = ar.analyze_gap(cur, nxt, lambda tid: "chr%d" % tid, fetch, opts, "r", helps)
            err = None
        except Exception as e:      # noqa: BLE001
            sig, err = None, type(e).__name__
        assert err == case["err"]
        if case["sig"] is None:
            assert sig is None
        else:
            got = [sig.type, sig.tstart, sig.tend, sig.bkps, sig.mechanism,
                   [[a.q_start, a.q_end, a.ref_start, a.ref_end, bool(a.is_reverse)] for a in sig.sorted_aligns]]
            assert got == case["sig"]
            kinds[sig.type + ("+help" if helps else "")] += 1
        assert [[h.ref_start, h.ref_end] for h in helps] == case["help_after"]
    assert len(kinds) >= 5 and sum(kinds.values()) == 433


def test_inside_align_fuzz(fx):
    for case in fx["inside"]:
        ops = cigar_ref.parse_cigar(case["cigar"])
        gaps = cigar_ref.scan_long_gaps(ops, case["ref_start"], 50)
        arr = np.array([(0, g[0], g[2], g[3], g[4], g[1]) for g in gaps],
                       dtype=[("aln", "<u4"), ("op", "<u4"), ("read_pos", "<i4"), ("ref_pos", "<i4"), ("len", "<i4"), ("kind", "<u4")])
        seg = Seg(case["q_start"], 0, case["ref_start"], case["ref_end"], 0, False, False, "main", 60, 0)
        got, _helpers = ar.analyze_inside_align(seg, arr)
        exp = case["segs"]
        assert (got is None) == (exp is None)
        if got is not None:
            assert [[s.q_start, s.q_end, s.ref_start, s.ref_end] for s in got] == exp
            assert all(s.type == "main" and not s.is_reverse for s in got)


def test_refine_type_fuzz(fx):
    opts = helpers.default_options()
    for case in fx["refine"]:
        t, b = refine_type(copy.deepcopy(case["types"]), copy.deepcopy(case["bkps"]), opts)
        assert list(t) == case["out_types"] and [list(x) for x in b] == case["out_bkps"]


def test_vote_fuzz(fx):
    p = Predict("chr", None)
    for case in fx["vote"]:
        reads = {k: {int(c): v for c, v in d.items()} for k, d in case["reads"].items()}
        got = p.get_region_potential_svtypes(reads)
        assert [[t, list(ids), [list(x) for x in bk]] for t, ids, bk in got] == case["out"]


def test_linear_and_score_fuzz(fx):
    for case in fx["linear"]:
        a, b = Segment(*case["a"]), Segment(*case["b"])
        assert oc.linearOrNot(a, b) == case["linear"]
        assert oc.cal_non_linear([a, b]) == case["score"]


def test_negative_reference_start_drops_the_window(tmp_path):
    """Found by tools/diff_ref.py (seed 5039): the reference fetches [min, max) of the two segments' reference
    coordinates in analyze_gap whether it needs the bases or not (analyze_reads.py:182); pysam refuses a negative start,
    run_detect turns the exception into its error string and the window writes nothing (run_collection.py:44-47)."""
    from svision_amd import pipeline
    opts = helpers.default_options()
    cur = Seg(0, 5000, -12, 4988, 0, False, False, "main", 60)
    nxt = Seg(5300, 9000, 4990, 8690, 0, False, False, "main", 60)
    with pytest.raises(ValueError):
        ar.analyze_gap(cur, nxt, lambda tid: "chr0", lambda c, s, e: b"A" * (e - s), opts, "r", [])

    class Boom:
        table = None
    orig = pipeline.detect_window
    pipeline.detect_window = lambda *a, **k: (_ for _ in ()).throw(ValueError("start out of range (-12)"))
    try:
        assert pipeline._collect_lines(Boom(), opts, "chr0", 0, 1000) == []
    finally:
        pipeline.detect_window = orig
