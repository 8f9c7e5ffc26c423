"""BASELINE configs[0] (cfg1) under -m gpu.  The reference's demo BAM is absent (/root/reference/.MISSING_LARGE_BLOBS:1; the
command is /root/reference/README.md:94-96: ``SVision -o ... -b supports/HG00733.svision.demo.bam -m ... -g ... -n HG00733 -s 5``);
BASELINE.md section 2 defines the stand-in: ONE 75 Mb contig, HiFi N(15 kb, 2 kb) reads at 30x, ``-s 5``.

The whole command line on the device (BAM + FASTA + checkpoint -> merged VCF, device ingest, helper processes) against the
CPU port of the same run: the file-based driver on the C oracle's scan with oracle-rasterised images classified by plain
PyTorch fp32 on the host.  Segment TSV: byte-identical.  VCF: the end-to-end rule of DESIGN.md section 3a -- every column but
QUAL byte-identical, QUAL identical where no image's 2-decimal softmax flipped, bounded where one did."""
import os
import subprocess
import sys

import numpy as np
import pytest

from svision_amd import cli, synth
from svision_amd.io import bam
from svision_amd.sample import Sample
from tests import helpers
from tests.test_e2e_golden import _split_qual, device_model, expected  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG1_LEN = 75_000_000


class CpuPort:
    """classifier(records) -> (None, argmax, softmax): C oracle rasteriser + PyTorch-CPU fp32 AlexNet (oracle/alexnet_torch.py)."""
    from_records = True

    def __init__(self, params):
        import torch
        from oracle.alexnet_torch import TorchAlexNet
        self.torch = torch
        self.net = TorchAlexNet(params, device="cpu")
        self.images = 0
        self.probs = []

    def __call__(self, records):
        from oracle import cbind
        rec = np.ascontiguousarray(records.cpu().numpy() if hasattr(records, "cpu") else records, np.int32)
        img = cbind.rasterize(rec, "NCHW")
        _l, cls, prob = self.net.predict(self.torch.from_numpy(img))
        self.images += len(rec)
        self.probs.append(prob.numpy())
        return None, cls.numpy(), prob.numpy()


def test_cfg1_stand_in_command_line_equals_the_cpu_port(expected, device_model, tmp_path):  # noqa: F811
    from tests import e2e_weights
    table, genome, _svs = synth.simulate(synth.SimConfig(contigs=[("contig75", CFG1_LEN)], coverage=30.0, seed=17))
    bam_path, fa = str(tmp_path / "cfg1.bam"), str(tmp_path / "cfg1.fa")
    seg = bam.encode_reference_segment(table, seq="random", seed=17)
    bam.write_bam_segments(bam_path, ["contig75"], [CFG1_LEN], [seg], index=True)
    bam.write_fasta(fa, genome)
    args = ["-n", "HG00733", "-s", "5", "--batch_size", "64", "--debug"]

    # 1. the command line on the device
    out = str(tmp_path / "device")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "SVision"), "-o", out, "-b", bam_path, "-m", device_model, "-g", fa, "-t", "8"] + args,
                       capture_output=True, text=True, timeout=1500, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout + r.stderr
    got_vcf = open(os.path.join(out, "HG00733.svision.s5.vcf")).read()
    got_tsv = open(os.path.join(out, "segments", "contig75.segments.all.bed")).read()

    # 2. the CPU port of the same run
    disk = bam.read_bam(bam_path)
    sample = Sample.with_scan(disk, bam.Fasta(fa), 50, helpers.oracle_scan(disk, 50))
    port = CpuPort(e2e_weights.fixture_params(expected))
    ref_out = str(tmp_path / "cpu_port")
    merged = cli.run(cli.parse_arguments(["-o", ref_out, "-b", bam_path, "-m", device_model, "-g", fa] + args), sample=sample, classifier=port)
    want_vcf = open(merged).read()
    want_tsv = open(os.path.join(ref_out, "segments", "contig75.segments.all.bed")).read()
    n_images = want_tsv.count("\n")
    n_records = sum(1 for l in want_vcf.splitlines() if not l.startswith("#"))
    assert n_images > 8000 and n_records > 300 and port.images >= n_images      # 8 windows of ~1,700 candidate images

    # 3. Step 1 does not depend on the CNN: identical, always
    assert got_tsv == want_tsv
    # 4. the VCF per the end-to-end rule
    got_rest, got_q = _split_qual(got_vcf)
    want_rest, want_q = _split_qual(want_vcf)
    assert got_rest == want_rest
    differ = sum(a != b for a, b in zip(got_q, want_q))
    worst = max([abs(int(a) - int(b)) for a, b in zip(got_q, want_q)] or [0])
    body = [float(l.split("\t")[5]) for l in open(os.path.join(ref_out, "predict_results", "contig75.predict.s5.vcf")) if l.strip()]
    scores = [s for s in body if s != 0]
    span = max(scores) - min(scores) if scores else 0.0
    bound = int(np.ceil(400.0 / span)) + 1 if span > 0 else 100
    print("\n[cfg1] images %d  records %d  QUAL differs on %d records, max |dQUAL| %d (bound %d)" % (n_images, n_records, differ, worst, bound))
    # a flip needs an image whose softmax lies within ~1e-5 of an x.xx5 boundary: a handful per 10^4 images at most
    assert differ <= 3 + n_images // 2000 and worst <= bound
