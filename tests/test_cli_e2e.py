"""End-to-end driver (svision_amd.cli.run) on the golden sample: tasking, per-window collection,
`cat`, prediction, score range, merge -- single process and 2-rank gloo -- must reproduce the
merged VCF the REFERENCE pipeline produced (tests/golden/predict_small.expected.json, case 0:
-s 3 --window_size 150000 --batch_size 128).  CNN outputs are injected from the fixture; the
device kernels are exercised by the -m gpu twin (test_gpu_pipeline.py)."""
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest

from svision_amd import cli
from tests import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case():
    with open(os.path.join(helpers.GOLDEN, "predict_small.expected.json")) as f:
        return json.load(f)["cases"][0]


class ChromInjected:
    """Feeds each chromosome's stored (argmax, softmax) batches in call order."""
    needs_images = False

    def __init__(self, case, chroms):
        self.queue = []
        b = case["batch_size"]
        for c in chroms:
            cls = np.asarray(case["chroms"][c]["classes"], np.int64)
            pr = np.asarray(case["chroms"][c]["probs"], np.uint32).view(np.float32).reshape(-1, 5)
            for i in range(0, len(cls), b):
                self.queue.append((cls[i:i + b], pr[i:i + b]))

    def __call__(self, _images):
        cls, pr = self.queue.pop(0)
        return None, cls, pr


def make_options(out_path, case):
    return cli.parse_arguments(["-o", out_path, "-b", "/virtual/sample.bam", "-m", "/virtual/model.ckpt", "-g", "/virtual/genome.fa",
                                "-n", "HGtest", "-s", str(case["min_support"]), "--window_size", "150000",
                                "--batch_size", str(case["batch_size"])])


def run_rank(out_path):
    """Body shared by the single-process test and the gloo workers."""
    from svision_amd import dist as sdist
    case = _case()
    rank, ws = sdist.init_from_env()
    sample = helpers.golden_sample(50)
    opts = make_options(out_path, case)
    chroms = case["chrom_order"]
    lengths = dict(zip(sample.table.references, sample.table.lengths))
    mine = sdist.shard_chromosomes(chroms, [lengths[c] for c in chroms], ws)[rank]
    return cli.run(opts, sample=sample, classifier=ChromInjected(case, mine)), case


def test_cli_single_process(oracle_lib, tmp_path):
    merged, case = run_rank(str(tmp_path))
    assert open(merged).read() == case["merged_vcf"]
    assert not os.path.exists(os.path.join(str(tmp_path), "segments"))          # cleaned up unless --debug


def test_cli_two_ranks_gloo(oracle_lib, tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", os.path.join(ROOT, "tests", "gloo_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    merged = os.path.join(str(tmp_path), "HGtest.svision.s3.vcf")
    assert open(merged).read() == _case()["merged_vcf"]


def test_tasking_matches_reference_quirks():
    refs, lens = ["chr1", "chr2", "chrUn"], [25_000_000, 9_000_000, 500]
    o = types.SimpleNamespace(chrom=None, contig=False, window_size=10_000_000)
    t = cli.build_tasks(o, refs, lens, ["chr1", "chr2"])
    assert t == {"chr1": [[0, 10_000_000], [10_000_000, 20_000_000], [20_000_000, 25_000_000]], "chr2": [[0, 9_000_000]]}
    o = types.SimpleNamespace(chrom=None, contig=True, window_size=10_000_000)
    assert cli.build_tasks(o, refs, lens, ["chr1", "chr2"]) == {"chr1": [[0, 25_000_000]], "chr2": [[0, 9_000_000]]}
    o = types.SimpleNamespace(chrom="chr1:15000000-40000000", contig=False, window_size=10_000_000)
    # SVision:225-232: windows restart at 0
    assert cli.build_tasks(o, refs, lens, ["chr1", "chr2"]) == {"chr1": [[0, 10_000_000], [10_000_000, 20_000_000], [20_000_000, 25_000_001]]}
    o = types.SimpleNamespace(chrom="chr2:100-5000", contig=False, window_size=10_000_000)
    assert cli.build_tasks(o, refs, lens, ["chr1", "chr2"]) == {"chr2": [[100, 5000]]}


def test_shard_chromosomes_lpt():
    from svision_amd import dist as sdist
    chroms = ["chr1", "chr2", "chr3", "chr4", "chr5"]
    lens = [248, 242, 198, 190, 181]
    shards = sdist.shard_chromosomes(chroms, lens, 2)
    assert sorted(c for s in shards for c in s) == sorted(chroms)
    assert shards == [["chr1", "chr4", "chr5"], ["chr2", "chr3"]]           # LPT, task order kept inside a rank
    loads = [sum(lens[chroms.index(c)] for c in s) for s in shards]
    assert max(loads) - min(loads) <= max(lens)
    assert sdist.shard_chromosomes(chroms, lens, 8)[5:] == [[], [], []]
    assert sdist.exchange_score_range([]) == (None, None)
    mx, mn = sdist.exchange_score_range([3.5, 1.25, 9.0])
    assert (float(mx), float(mn)) == (9.0, 1.25)
    assert sdist.gather_texts({"a": "x"}) == {"a": "x"}
