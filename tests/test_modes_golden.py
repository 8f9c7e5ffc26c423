"""--contig mode (BASELINE configs[4]) and an ONT-like sample (configs[3] stand-in: long noisy reads, several SVs per
read, > 4 supplementary alignments, low-MAPQ / secondary / unmapped records, reads seen by two windows) vs the TSV
the reference's run_detect wrote (tests/golden/make_modes_fixture.py)."""
import json
import os

import pytest

from svision_amd.collection.output_clusters import collect_pair_lines
from svision_amd.collection.run_collection import detect_window
from svision_amd.io import bam
from svision_amd.sample import Sample
from tests import helpers


@pytest.fixture(scope="module")
def expected():
    with open(os.path.join(helpers.GOLDEN, "modes.expected.json")) as f:
        return json.load(f)


def _run(case, bam_name, fasta_name, device):
    fasta = helpers.load_golden_fasta(fasta_name)
    total = 0
    for w in case["windows"]:
        table = bam.read_bam(os.path.join(helpers.GOLDEN, bam_name))
        if device is None:
            sample = Sample.with_scan(table, fasta, 50, helpers.oracle_scan(table, 50))
        else:
            sample = Sample.from_table(table, fasta, 50, device)
        opts = helpers.default_options(**case["options"])
        _sigs, clusters = detect_window(opts, sample, w["chrom"], w["start"], w["end"])
        tsv = "".join(p.text() for p in collect_pair_lines(clusters, opts))
        assert tsv == w["tsv"]
        total += tsv.count("\n")
    return total


def test_contig_mode_cpu(expected, oracle_lib):
    assert _run(expected["contig"], "collect_small.bam", "collect_small.fa.gz", None) == 616


def test_ont_like_cpu(expected, oracle_lib):
    assert _run(expected["ont"], "ont_small.bam", "ont_small.fa.gz", None) == 241


@pytest.mark.gpu
def test_modes_gpu(expected):
    assert _run(expected["contig"], "collect_small.bam", "collect_small.fa.gz", "cuda:0") == 616
    assert _run(expected["ont"], "ont_small.bam", "ont_small.fa.gz", "cuda:0") == 241
