"""BASELINE configs[2] (whole genome, chromosomes sharded over the ranks, one RCCL exchange) under -m gpu:

* a 24-contig miniature of GRCh38 (every primary chromosome at 1/200 of its length, HiFi-like reads, SVs every ~30 kb)
  from a BAM file through the command line with helper processes -- PooledHotPath, four batches per launch, three
  streams, windows of all 24 chromosomes in flight -- against the file-based driver (run_detect -> TSV -> Predict.run,
  one batch per launch) on the C oracle's scan;
* the same command line with its cross-rank exchange (score-range all_reduce + record gather, svision_amd/dist.py) on
  the `nccl` backend -- RCCL -- as a one-rank group (SVX_FORCE_DIST=1: a one-GPU box cannot host two RCCL ranks)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from svision_amd import cli, synth
from svision_amd.io import bam
from svision_amd.sample import Sample
from tests import helpers
from tests.test_e2e_golden import device_model, expected  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GRCH38 = (248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
          135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
          46709983, 50818468, 156040895, 57227415)
NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]


@pytest.fixture(scope="module")
def miniature(tmp_path_factory):
    d = tmp_path_factory.mktemp("wg_mini")
    contigs = [(n, l // 200) for n, l in zip(NAMES, GRCH38)]
    cfg = synth.SimConfig(contigs=contigs, coverage=20, read_len_mean=9000, read_len_sd=1500, err_rate=0.003, sv_spacing=18_000,
                          sv_min_gap=9_000, sv_max=4000, inline_max=1500, seed=41)
    table, genome, _svs = synth.simulate(cfg)
    bam_path, fa = str(d / "wg_mini.bam"), str(d / "wg_mini.fa")
    bam.write_bam(bam_path, table, level=1, index=True)
    bam.write_fasta(fa, genome)
    return bam_path, fa, contigs


def _cli(args, env=None, timeout=900):
    return subprocess.run([sys.executable, os.path.join(ROOT, "SVision")] + args, capture_output=True, text=True, timeout=timeout,
                          env=dict(os.environ, PYTHONPATH=ROOT, **(env or {})))


ARGS = ["-n", "HGwg", "-s", "4", "--window_size", "500000", "--batch_size", "64"]


def test_whole_genome_miniature_pooled_equals_file_based_driver(miniature, device_model, tmp_path):
    from svision_amd.network.predict import load_classifier
    bam_path, fa, contigs = miniature
    out = str(tmp_path / "pooled")
    r = _cli(["-o", out, "-b", bam_path, "-m", device_model, "-g", fa, "-t", "8", "--debug"] + ARGS)
    assert r.returncode == 0, r.stdout + r.stderr
    pooled = open(os.path.join(out, "HGwg.svision.s4.vcf")).read()
    body = [l for l in pooled.splitlines() if not l.startswith("#")]
    assert len({l.split("\t")[0] for l in body}) >= 20 and len(body) > 200            # calls on (nearly) every chromosome
    n_images = sum(open(os.path.join(out, "segments", n + ".segments.all.bed")).read().count("\n") for n, _l in contigs)
    assert n_images > 5000
    windows = sum(-(-l // 500000) for _n, l in contigs)
    assert n_images / windows > 128                                                     # launches of several batches did happen

    # the file-based driver on the C oracle's scan, the CNN one batch of 64 per launch
    table = bam.read_bam(bam_path)
    sample = Sample.with_scan(table, bam.Fasta(fa), 50, helpers.oracle_scan(table, 50))
    ref_out = str(tmp_path / "filebased")
    opts = cli.parse_arguments(["-o", ref_out, "-b", bam_path, "-m", device_model, "-g", fa, "--debug"] + ARGS)
    merged = cli.run(opts, sample=sample, classifier=load_classifier(device_model, device="cuda:0"))
    assert open(merged).read() == pooled
    for n, _l in contigs:
        for rel in ("segments/%s.segments.all.bed" % n, "predict_results/%s.predict.s4.vcf" % n, "predict_results/%s.predict.s4.score.txt" % n):
            assert open(os.path.join(out, rel)).read() == open(os.path.join(ref_out, rel)).read(), rel


def test_exchange_over_rccl_equals_the_plain_run(expected, device_model, tmp_path):  # noqa: F811
    case = expected["cases"]["collect"]
    fasta = helpers.load_golden_fasta(case["data"] + ".fa.gz")
    fa = str(tmp_path / "genome.fa")
    bam.write_fasta(fa, {n: fasta._seq[n] for n in fasta.references})
    args = ["-b", os.path.join(helpers.GOLDEN, case["data"] + ".bam"), "-m", device_model, "-g", fa, "-n", "HGr", "-t", "3"] + case["args"]
    plain = _cli(["-o", str(tmp_path / "plain")] + args)
    assert plain.returncode == 0, plain.stdout + plain.stderr
    env = {"SVX_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29711",
           "SVX_DIST_BACKEND": "nccl", "SVX_TIMING": "1", "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
    rccl = _cli(["-o", str(tmp_path / "rccl")] + args, env=env)
    assert rccl.returncode == 0, rccl.stdout + rccl.stderr
    assert "exchange backend nccl" in rccl.stdout                                        # the group really was RCCL
    a = open(os.path.join(str(tmp_path / "plain"), "HGr.svision.s3.vcf")).read()
    b = open(os.path.join(str(tmp_path / "rccl"), "HGr.svision.s3.vcf")).read()
    assert a == b and a.count("\n") > 30


def test_eight_ranks_on_the_miniature_equal_one_rank(miniature, device_model, tmp_path):
    """the first contact with an 8-GPU node must not also be the first 8-rank run.  Eight torchrun ranks
    on the one GPU of the box (gloo: RCCL refuses duplicate devices; the driver's multi-GPU runs use nccl), the 24
    chromosomes LPT-sharded three to a rank, every rank streaming only its own byte ranges of the BAM through the device
    ingest engine with the host budget of an eighth of the box (quota-aware sizing: helpers, pread threads): rank 0's merged VCF
    is the one-rank run's, byte for byte, and every rank logs its shard."""
    bam_path, fa, contigs = miniature
    args = ["-b", bam_path, "-m", device_model, "-g", fa, "-t", "2"] + ARGS
    one = _cli(["-o", str(tmp_path / "one")] + args)
    assert one.returncode == 0, one.stdout + one.stderr
    env = dict(os.environ, PYTHONPATH=ROOT, SVX_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT="29653",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", "29653", os.path.join(ROOT, "SVision"), "-o", str(tmp_path / "eight")] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    a = open(os.path.join(str(tmp_path / "one"), "HGwg.svision.s4.vcf")).read()
    b = open(os.path.join(str(tmp_path / "eight"), "HGwg.svision.s4.vcf")).read()
    assert a == b and a.count("\n") > 200
    logs = sorted(f for f in os.listdir(str(tmp_path / "eight")) if f.endswith(".log"))
    assert len(logs) == 8                                          # one log per rank
    from svision_amd import dist as sdist
    shards = sdist.shard_chromosomes([n for n, _l in contigs], [l for _n, l in contigs], 8)
    assert sorted(len(s) for s in shards) == [3] * 8               # LPT on 24 chromosomes: three each
