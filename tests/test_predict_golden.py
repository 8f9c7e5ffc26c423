"""Vote / VCF / merge logic vs golden vectors produced by the reference's own Predict.run,
write_results_to_vcf, genotyper and merge_split_vcfs (tests/golden/make_predict_fixture.py),
with the CNN outputs injected from the fixture."""
import json
import os

import numpy as np
import pytest

from svision_amd.network import output
from svision_amd.network.create_batch import BatchGenerator
from svision_amd.network.predict import Predict
from tests import helpers


@pytest.fixture(scope="module")
def expected():
    with open(os.path.join(helpers.GOLDEN, "predict_small.expected.json")) as f:
        return json.load(f)


class Injected:
    needs_images = False

    def __init__(self, classes, probs, batch):
        self.classes = np.asarray(classes, np.int64)
        self.probs = np.asarray(probs, np.uint32).view(np.float32).reshape(-1, 5)
        self.batch, self.i = batch, 0

    def __call__(self, _images):
        sl = slice(self.i * self.batch, (self.i + 1) * self.batch)
        self.i += 1
        return None, self.classes[sl], self.probs[sl]


def test_predict_and_merge_match_reference(expected, oracle_lib, tmp_path):
    sample = helpers.golden_sample(50)
    for case in expected["cases"]:
        opts = helpers.default_options(min_support=case["min_support"], batch_size=case["batch_size"], qname=case["qname"],
                                       sample="HGtest", out_path=str(tmp_path), source_version="1.4")
        pred_dir = tmp_path / ("pred_%d_%d" % (case["min_support"], case["batch_size"]))
        pred_dir.mkdir()
        for chrom in case["chrom_order"]:
            c = case["chroms"][chrom]
            bed = tmp_path / (chrom + ".all.bed")
            bed.write_text(c["tsv"])
            gen = BatchGenerator(str(bed), nb_classes=5, batch_size=case["batch_size"], device="cpu")
            assert gen.labels == c["labels"] and gen.images == c["data"]
            prefix = str(pred_dir / ("%s.predict.s%d" % (chrom, case["min_support"])))
            Predict(chrom, str(bed)).run(prefix, opts, classifier=Injected(c["classes"], c["probs"], case["batch_size"]), sample=sample)
            assert open(prefix + ".vcf").read() == c["vcf"]
            assert open(prefix + ".score.txt").read() == c["score"]
        scores = output.cal_scores_max_min(str(pred_dir))
        mx, mn = np.max(scores), np.min(scores)
        assert float(mx) == case["max_score"] and float(mn) == case["min_score"]
        merged = str(tmp_path / "merged.vcf")
        output.merge_split_vcfs(str(pred_dir), merged, mx, mn, case["chrom_order"], opts, fasta=sample.fasta)
        assert open(merged).read() == case["merged_vcf"]
