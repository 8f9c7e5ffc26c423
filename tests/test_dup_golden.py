"""Duplicate BAM records: upstream compares segment dicts by value, the product mirrors it with Seg.same_value
(tests/golden/make_dup_fixture.py ran the reference's run_detect on a BAM holding ~40 % of its supplementary records
and 8 % of all records twice)."""
import json
import os

import pytest

from svision_amd.collection.output_clusters import collect_pair_lines
from svision_amd.collection.run_collection import detect_window
from tests import helpers


@pytest.fixture(scope="module")
def expected():
    with open(os.path.join(helpers.GOLDEN, "dup_small.expected.json")) as f:
        return json.load(f)


def _run(expected, device, identity_only=False):
    lines = 0
    mismatches = 0
    for case in expected["cases"]:
        opts = helpers.default_options(**case["options"])
        for w in case["windows"]:
            sample = helpers.golden_sample(opts.min_sv_size, device=device, name="dup_small")
            _sigs, clusters = detect_window(opts, sample, w["chrom"], w["start"], w["end"], w["part"])
            got = "".join(p.text() for p in collect_pair_lines(clusters, opts))
            if identity_only:
                mismatches += got != w["tsv"]
            else:
                assert got == w["tsv"]
            lines += got.count("\n")
    return lines, mismatches


def test_duplicate_records_follow_the_reference_cpu(expected, oracle_lib):
    assert expected["duplicated_supplementary"] > 20
    lines, _ = _run(expected, None)
    assert lines > 100


def test_identity_comparison_would_differ(expected, oracle_lib):
    """The fixture really exercises the by-value comparison: with object identity the TSV changes."""
    from svision_amd.collection import classes
    classes.BY_VALUE[0] = False                               # Seg is an extension type when compiled: its methods cannot be patched
    try:
        _lines, mismatches = _run(expected, None, identity_only=True)
    finally:
        classes.BY_VALUE[0] = True
    assert mismatches > 0


@pytest.mark.gpu
def test_duplicate_records_follow_the_reference_gpu(expected):
    _run(expected, "cuda:0")
