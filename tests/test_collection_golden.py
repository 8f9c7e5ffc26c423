"""Host collection logic vs golden vectors produced by the reference itself
(tests/golden/make_collect_fixture.py ran /root/reference's run_detect on collect_small.bam)."""
import json
import os

import pytest

from svision_amd.collection.run_collection import detect_window
from svision_amd.collection.output_clusters import collect_pair_lines
from tests import helpers


@pytest.fixture(scope="module")
def expected():
    with open(os.path.join(helpers.GOLDEN, "collect_small.expected.json")) as f:
        return json.load(f)


def _check_windows(sample_factory, expected):
    n_sig = n_lines = 0
    for w in expected["windows"]:
        sample = sample_factory()           # fresh: segments are mutated in place by the collection step
        opts = helpers.default_options(min_support=w["min_support"])
        sigs, clusters = detect_window(opts, sample, w["chrom"], w["start"], w["end"], w["part"])
        got = [[s.type, s.tstart, s.tend, s.qname, s.bkps, s.mechanism,
                [[a.q_start, a.q_end, a.ref_start, a.ref_end, bool(a.is_reverse)] for a in s.sorted_aligns]] for s in sigs]
        assert len(got) == len(w["signatures"])
        for g, e in zip(got, w["signatures"]):
            assert g == e
        cl = [[c.contig, c.cstart, c.cend, c.read_num, c.coverage, [s.qname for s in c.signatures]] for c in clusters]
        assert cl == w["clusters"]
        tsv = "".join(p.text() for p in collect_pair_lines(clusters, opts))
        assert tsv == w["tsv"]
        n_sig += len(got)
        n_lines += tsv.count("\n")
    assert n_sig == 776 and n_lines == 1214


def test_collection_matches_reference_cpu(expected, oracle_lib):
    _check_windows(lambda: helpers.golden_sample(50), expected)


@pytest.mark.gpu
def test_collection_matches_reference_gpu(expected):
    _check_windows(lambda: helpers.golden_sample(50, device="cuda:0"), expected)
