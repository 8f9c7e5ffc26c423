"""GPU twins of the pipeline tests (-m gpu): the device scan feeds the collection step, the HIP
rasteriser feeds the PyTorch-ROCm AlexNet restored from a TF checkpoint prefix."""
import os

import numpy as np
import pytest
import torch

from svision_amd import cli
from svision_amd.network import tf_checkpoint as ck
from svision_amd.network.alexnet import AlexNet
from svision_amd.network.create_batch import BatchGenerator
from svision_amd.network.predict import load_classifier
from tests import helpers
from tests.test_cli_e2e import ChromInjected, _case, make_options

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_cli_fresh(args, timeout=600):
    """The command line in a fresh process (``-t N`` forks its helpers before the first HIP call there; forking from this
    long-lived test process, which owns a GPU context and gigabytes of device allocations, can stall for a minute)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "SVision")] + list(args), capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    return r


def test_cli_with_device_scan_reproduces_reference_vcf(tmp_path):
    case = _case()
    sample = helpers.golden_sample(50, device="cuda:0")
    merged = cli.run(make_options(str(tmp_path), case), sample=sample, classifier=ChromInjected(case, case["chrom_order"]))
    assert open(merged).read() == case["merged_vcf"]


@pytest.fixture(scope="module")
def checkpoint(tmp_path_factory):
    from oracle import alexnet_ref
    params = alexnet_ref.random_params(seed=7)
    prefix = str(tmp_path_factory.mktemp("ckpt") / "svision-cnn-model.ckpt")
    ck.write_checkpoint(prefix, params)
    return prefix, params


def test_classifier_from_checkpoint_matches_cpu_checker(checkpoint, tmp_path):
    """north_star tolerance: softmax within 1e-3 (fp32) of the CPU fp32 restatement, same classes."""
    from oracle import encode_ref
    prefix, params = checkpoint
    case = _case()
    bed = tmp_path / "chrB.bed"
    bed.write_text(case["chroms"]["chrB"]["tsv"])
    gen = BatchGenerator(str(bed), nb_classes=5, batch_size=64, layout="NCHW")
    classify = load_classifier(prefix, device="cuda:0")           # records -> (logits, class, softmax), sparse first layer
    from oracle.alexnet_torch import TorchAlexNet
    checker = TorchAlexNet(params, device="cpu")                  # plain PyTorch fp32 on the host
    for _ in range(gen.data_size // 64):
        lo = gen.pointer
        records, _labels = gen.next_records(64)
        _logits, cls, prob = classify(records)
        gen.pointer = lo
        images, _labels = gen.next_batch(64)                      # the dense image path of the BatchGenerator API
        want_img = encode_ref.encode_records(gen.records[lo:lo + 64])            # oracle rasteriser
        assert np.array_equal(images.cpu().numpy(), want_img.transpose(0, 3, 1, 2))
        _l, wcls, wprob = checker.predict(torch.from_numpy(np.ascontiguousarray(want_img.transpose(0, 3, 1, 2))))
        assert np.abs(prob - wprob.numpy()).max() < 1e-3
        sure = np.sort(wprob.numpy(), axis=1)[:, -1] - np.sort(wprob.numpy(), axis=1)[:, -2] > 2e-3
        assert np.array_equal(cls[sure], wcls.numpy()[sure])


def test_full_cli_on_device(checkpoint, tmp_path):
    """BAM file + FASTA + checkpoint prefix on disk -> VCF, everything on the device path."""
    from svision_amd.io import bam
    prefix, _params = checkpoint
    fasta = helpers.load_golden_fasta()
    fa = str(tmp_path / "genome.fa")
    bam.write_fasta(fa, {n: fasta._seq[n] for n in fasta.references})
    out = str(tmp_path / "out")
    opts = cli.parse_arguments(["-o", out, "-b", os.path.join(helpers.GOLDEN, "collect_small.bam"), "-m", prefix, "-g", fa,
                                "-n", "HGtest", "-s", "3", "--window_size", "150000", "--batch_size", "64", "--debug"])
    merged = cli.run(opts)
    body = [l for l in open(merged).read().splitlines() if not l.startswith("#")]
    assert len(body) > 0
    case = _case()
    for chrom in case["chrom_order"]:                            # the encode side is independent of the CNN
        got = open(os.path.join(out, "segments", chrom + ".segments.all.bed")).read()
        assert got == case["chroms"][chrom]["tsv"]
    # every record sits on a candidate site of the TSV
    sites = {tuple(l.split("\t")[0].split("+")[:3]) for c in case["chrom_order"] for l in case["chroms"][c]["tsv"].splitlines()}
    for l in body:
        f = l.split("\t")
        assert (f[0], f[1], f[7].split(";")[0][4:]) in sites


def test_streaming_pipelines_agree(checkpoint):
    """Single-process eager, graph-replay multi-stream, and the pooled (helper processes) pipelines
    produce the same per-window VCF bodies and scores."""
    from svision_amd.pipeline import HotPath, PooledHotPath
    from svision_amd.network.tf_checkpoint import read_checkpoint
    prefix, _ = checkpoint
    net = AlexNet(read_checkpoint(prefix), device="cuda:0")
    windows = [("chrA", 0, 150_000), ("chrA", 150_000, 300_000), ("chrA", 300_000, 420_000), ("chrB", 0, 150_000), ("chrB", 150_000, 200_000)]
    outs = []
    for kind in ("eager", "graph", "pool"):
        sample = helpers.golden_sample(50, device="cuda:0")
        opts = helpers.default_options(min_support=3, batch_size=64, bam_path="<resident>")
        if kind == "pool":
            hp = PooledHotPath(sample, opts, net, device="cuda:0", n_workers=3, n_streams=2)
        else:
            hp = HotPath(sample, opts, net, device="cuda:0", n_streams=1 if kind == "eager" else 3, use_graph=kind != "eager")
        got = {(r.chrom, r.start): (r.vcf, r.scores, r.n_sites, r.n_images) for r in hp.run_windows(windows)}
        if kind == "pool":
            hp.close()
        outs.append(got)
    assert outs[0] == outs[1] == outs[2]
    assert sum(v[3] for v in outs[0].values()) == 610 and sum(v[2] for v in outs[0].values()) > 20


def test_a_window_that_fails_after_parts_have_left_contributes_nothing(checkpoint, monkeypatch):
    """run_detect turns any exception of a window into a message and the window into nothing (run_collection.py:44-47).
    In the streaming pipeline parts of such a window may already be on the device when its collection fails: they are
    dropped, the other windows are untouched."""
    from svision_amd import pipeline
    from svision_amd.pipeline import PooledHotPath
    from svision_amd.network.tf_checkpoint import read_checkpoint
    prefix, _ = checkpoint
    net = AlexNet(read_checkpoint(prefix), device="cuda:0")
    windows = [("chrA", 0, 150_000), ("chrA", 150_000, 300_000), ("chrA", 300_000, 420_000), ("chrB", 0, 150_000), ("chrB", 150_000, 200_000)]
    opts = helpers.default_options(min_support=3, batch_size=64, bam_path="<resident>")

    def run():
        hp = PooledHotPath(helpers.golden_sample(50, device="cuda:0"), opts, net, device="cuda:0", n_workers=3, n_streams=2)
        try:
            return {(r.chrom, r.start): (r.vcf, r.scores, r.n_sites, r.n_images, r.head, r.tail) for r in hp.run_windows(windows)}
        finally:
            hp.close()
    want = run()
    victim = max(want, key=lambda k: want[k][3])                 # the window with most images: several parts
    assert want[victim][3] > 128
    real, real_detect, current = pipeline.iter_pair_lines, pipeline.detect_window, []

    def detect(options, sample, chrom, start, end, part_num=0):   # the helpers are forked after these patches: they inherit them
        current[:] = [(chrom, start)]
        return real_detect(options, sample, chrom, start, end, part_num)

    def failing(clusters, options):
        n = 0
        for lines in real(clusters, options):
            yield lines
            n += len(lines)
            if current[0] == victim and n > 96:
                raise ValueError("start out of range (-1)")
    monkeypatch.setattr(pipeline, "detect_window", detect)
    monkeypatch.setattr(pipeline, "iter_pair_lines", failing)
    monkeypatch.setattr(pipeline._collect_parts, "__defaults__", (32,))     # parts of >= 32 lines: some leave before the failure
    got = run()
    assert got[victim][:4] == ("", "", 0, 0) and got[victim][4] is None and got[victim][5] is None
    for key in want:
        if key != victim:
            assert got[key] == want[key]


def test_cli_hash_mode_on_device(checkpoint, tmp_path):
    """--hash end to end on the device path: BAM with read bases -> TSV identical to the reference's."""
    import json
    from svision_amd.io import bam
    prefix, _params = checkpoint
    fasta = helpers.load_golden_fasta("hash_collect.fa.gz")
    fa = str(tmp_path / "genome.fa")
    bam.write_fasta(fa, {n: fasta._seq[n] for n in fasta.references})
    out = str(tmp_path / "out")
    opts = cli.parse_arguments(["-o", out, "-b", os.path.join(helpers.GOLDEN, "hash_collect.bam"), "-m", prefix, "-g", fa,
                                "-n", "HGhash", "-s", "3", "--hash", "--batch_size", "64", "--debug"])
    try:
        cli.run(opts)
    except SystemExit:              # random weights may leave no supported call: upstream prints "Empty output" and exits too
        pass
    with open(os.path.join(helpers.GOLDEN, "hash_collect.expected.json")) as f:
        want = [w for w in json.load(f)["windows"] if w["hash"]][0]
    assert open(os.path.join(out, "segments", "chrH.segments.all.bed")).read() == want["tsv"]


@pytest.mark.parametrize("extra", [[], ["--contig"], ["-c", "chrA"], ["-c", "chrB:1000-120000"], ["--qname", "--min_mapq", "30"],
                                   ["--max_sv_size", "3000", "--min_sv_size", "80", "-s", "2"]],
                         ids=["default", "contig", "one-chromosome", "region", "qname-mapq", "size-limits"])
def test_streaming_cli_equals_file_based_cli(checkpoint, tmp_path, extra):
    """The device CLI (windows streamed, no TSV round trip) writes the same segment files, per-chromosome VCF bodies,
    scores and merged VCF as the file-based flow (run_detect -> cat -> Predict.run) with the same network, for several
    option sets of the command line."""
    from svision_amd.io import bam
    from svision_amd.network.predict import load_classifier
    prefix, _params = checkpoint
    fasta = helpers.load_golden_fasta()
    fa = str(tmp_path / "genome.fa")
    bam.write_fasta(fa, {n: fasta._seq[n] for n in fasta.references})
    outs = {}
    for kind in ("stream", "files"):
        out = str(tmp_path / kind)
        opts = cli.parse_arguments(["-o", out, "-b", os.path.join(helpers.GOLDEN, "collect_small.bam"), "-m", prefix, "-g", fa,
                                    "-n", "HGtest", "-s", "3", "--window_size", "150000", "--batch_size", "64", "--debug"] + extra)
        merged = cli.run(opts, classifier=None if kind == "stream" else load_classifier(prefix, device="cuda:0"))
        files = {"merged": open(merged).read()}
        for sub in ("segments", "predict_results"):
            for fn in sorted(os.listdir(os.path.join(out, sub))):
                files[sub + "/" + fn] = open(os.path.join(out, sub, fn)).read()
        outs[kind] = files
    assert outs["stream"].keys() == outs["files"].keys()
    for k in outs["stream"]:
        assert outs["stream"][k] == outs["files"][k], k


@pytest.mark.parametrize("threads,engine", [("1", "cpu"), ("3", "cpu"), ("3", "gpu")])
def test_two_ranks_from_an_indexed_bam_equal_one_rank(checkpoint, tmp_path, threads, engine):
    """The whole multi-rank command line from files: two torchrun ranks (gloo here: one GPU is shared, RCCL refuses
    duplicate devices; the driver's multi-GPU runs use nccl) each decode only their chromosomes through the .bai, stream
    them through the device path, and the single exchange gives rank 0 the merged VCF of the one-rank run, byte for byte."""
    import subprocess, sys
    from svision_amd.io import bam
    prefix, _params = checkpoint
    fasta = helpers.load_golden_fasta()
    fa = str(tmp_path / "genome.fa")
    bam.write_fasta(fa, {n: fasta._seq[n] for n in fasta.references})
    src = bam.read_bam(os.path.join(helpers.GOLDEN, "collect_small.bam"))
    path = str(tmp_path / "indexed.bam")
    bam.write_bam(path, src, index=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["-b", path, "-m", prefix, "-g", fa, "-n", "HGtest", "-s", "3", "--window_size", "150000", "--batch_size", "64"]
    one = cli.run(cli.parse_arguments(["-o", str(tmp_path / "one")] + args))
    env = dict(os.environ, PYTHONPATH=root, SVX_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", SVX_INGEST=engine)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29641", os.path.join(root, "SVision"), "-o", str(tmp_path / "two"), "-t", threads] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    two = os.path.join(str(tmp_path / "two"), os.path.basename(one))
    assert open(two).read() == open(one).read()
    logs = [f for f in os.listdir(str(tmp_path / "two")) if f.endswith(".log")]
    assert any("streamed from" in open(os.path.join(str(tmp_path / "two"), f)).read() for f in logs)


def test_cli_with_helper_processes_equals_one_process(checkpoint, tmp_path):
    """``-t 4`` (four forked host helpers, windows finishing in any order) writes the same files as the one-process run."""
    from svision_amd.io import bam
    prefix, _params = checkpoint
    fasta = helpers.load_golden_fasta()
    fa = str(tmp_path / "genome.fa")
    bam.write_fasta(fa, {n: fasta._seq[n] for n in fasta.references})
    args = ["-b", os.path.join(helpers.GOLDEN, "collect_small.bam"), "-m", prefix, "-g", fa, "-n", "HGtest", "-s", "3",
            "--window_size", "60000", "--batch_size", "64", "--debug"]
    one = cli.run(cli.parse_arguments(["-o", str(tmp_path / "one")] + args))
    r = run_cli_fresh(["-o", str(tmp_path / "four"), "-t", "4"] + args)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    four = os.path.join(str(tmp_path / "four"), os.path.basename(one))
    assert open(four).read() == open(one).read() and open(one).read().count("\n") > 20
    for sub in ("segments", "predict_results"):
        names = sorted(os.listdir(str(tmp_path / "one" / sub)))
        assert names == sorted(os.listdir(str(tmp_path / "four" / sub))) and len(names) >= 4
        for n in names:
            assert open(str(tmp_path / "four" / sub / n)).read() == open(str(tmp_path / "one" / sub / n)).read(), n


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_device_scan_equals_oracle_scan_on_simulated_samples(seed):
    """The resident scan of whole simulated samples (HiFi- and ONT-like CIGARs with clips, split alignments, in-CIGAR SVs)
    equals the oracle's, array for array -- and so does every window's TSV built on top of it."""
    from svision_amd import synth
    from svision_amd.collection.output_clusters import collect_pair_lines
    from svision_amd.collection.run_collection import detect_window
    from svision_amd.io import bam
    from svision_amd.sample import Sample
    rng = np.random.default_rng(seed)
    cfg = synth.SimConfig(contigs=[("c0", 300_000), ("c1", 180_000)], coverage=float(rng.choice([8, 16])),
                          read_len_mean=float(rng.choice([4000, 9000])), read_len_sd=1500.0, lognormal=bool(seed % 2),
                          err_rate=float(rng.choice([0.002, 0.03])), sv_spacing=5000.0, sv_min_gap=4000, sv_max=3000, inline_max=1500, seed=seed)
    table, genome, _ = synth.simulate(cfg)
    fasta = bam.Fasta(sequences=genome)
    dev_sample = Sample.from_table(table, fasta, 50, device="cuda:0")
    ref_sample = Sample.with_scan(table, fasta, 50, helpers.oracle_scan(table, 50))
    assert np.array_equal(dev_sample.gap_off, ref_sample.gap_off)
    assert dev_sample.gaps.tobytes() == ref_sample.gaps.tobytes()
    assert np.array_equal(dev_sample.stats, ref_sample.stats)
    opts = helpers.default_options(min_support=2)
    for chrom, clen in cfg.contigs:
        a = detect_window(opts, Sample.from_table(table, fasta, 50, device="cuda:0"), chrom, 0, clen)[1]
        b = detect_window(opts, Sample.with_scan(table, fasta, 50, helpers.oracle_scan(table, 50)), chrom, 0, clen)[1]
        ta = "".join(p.text() for p in collect_pair_lines(a, opts))
        tb = "".join(p.text() for p in collect_pair_lines(b, opts))
        assert ta == tb and ta.count("\n") > 10


def test_hash_mode_with_helper_processes(checkpoint, tmp_path):
    """--hash (read bases resident, k-mer re-aligner in the collection step) with -t 3: the helpers' segment TSV is the
    reference's, as in the one-process run."""
    import json
    from svision_amd.io import bam
    prefix, _params = checkpoint
    fasta = helpers.load_golden_fasta("hash_collect.fa.gz")
    fa = str(tmp_path / "genome.fa")
    bam.write_fasta(fa, {n: fasta._seq[n] for n in fasta.references})
    out = str(tmp_path / "out")
    r = run_cli_fresh(["-o", out, "-b", os.path.join(helpers.GOLDEN, "hash_collect.bam"), "-m", prefix, "-g", fa,
                       "-n", "HGhash", "-s", "3", "--hash", "--batch_size", "64", "--debug", "-t", "3"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]     # ("Empty output" exits with 0, as upstream)
    with open(os.path.join(helpers.GOLDEN, "hash_collect.expected.json")) as f:
        want = [w for w in json.load(f)["windows"] if w["hash"]][0]
    assert open(os.path.join(out, "segments", "chrH.segments.all.bed")).read() == want["tsv"]


@pytest.mark.gpu
def test_cli_graph_mode_on_device(checkpoint, tmp_path):
    """--graph --qname end to end on the device path, one process and ``-t 3``: the per-read graphs come out of the
    collection as the reference writes them (fixture), step 3 leaves the graph VCF (the plain VCF's records, each with its
    GraphID / GFA_* fields), the per-record graphs and the two summaries, and removes the per-site folders and the plain VCF."""
    import gzip
    import json
    from svision_amd.io import bam
    prefix, _params = checkpoint
    fasta = helpers.load_golden_fasta("graph_small.fa.gz")
    fa = str(tmp_path / "genome.fa")
    bam.write_fasta(fa, {n: fasta._seq[n] for n in fasta.references})
    with gzip.open(os.path.join(helpers.GOLDEN, "graph_small.expected.json.gz"), "rt") as f:
        want = json.load(f)
    args = ["-b", os.path.join(helpers.GOLDEN, "graph_small.bam"), "-m", prefix, "-g", fa, "-n", "HGg", "-s", "3",
            "--window_size", str(want["window"]), "--batch_size", "64", "--qname", "--debug"]
    plain = cli.run(cli.parse_arguments(["-o", str(tmp_path / "plain")] + args))
    body = [l for l in open(plain).read().splitlines() if not l.startswith("#")]
    outs = {}
    for tag, extra in (("one", []), ("three", ["-t", "3"])):
        out = str(tmp_path / tag)
        if extra:
            r = run_cli_fresh(["-o", out, "--graph"] + extra + args)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        else:
            assert cli.run(cli.parse_arguments(["-o", out, "--graph"] + args)).endswith("HGg.svision.s3.graph.vcf")
        assert not os.path.exists(os.path.join(out, "HGg.svision.s3.vcf"))
        got = [l for l in open(os.path.join(out, "HGg.svision.s3.graph.vcf")).read().splitlines() if not l.startswith("#")]
        assert len(got) == len(body) >= 1                  # (random weights: few calls; the complex ones are covered on the CPU, test_graph_golden)
        for a, b in zip(got, body):
            ca, cb = a.split("\t"), b.split("\t")
            assert ca[:7] == cb[:7] and ca[8:] == cb[8:] and ca[7].startswith(cb[7] + ";GraphID=")
            assert ("GraphID=-1;GFA_ID=.;GFA_S=.;GFA_L=." in ca[7]) == ("CSV" not in b)
        left = sorted(os.listdir(os.path.join(out, "graphs")))
        assert all(n.endswith(".gfa") for n in left) and len(left) == sum("CSV" in b for b in body)
        assert os.path.exists(os.path.join(out, "HGg.graph_exactly_match.txt")) and os.path.exists(os.path.join(out, "HGg.graph_symmetry_match.txt"))
        assert open(os.path.join(out, "segments", "chrG.segments.all.bed")).read() == "".join(w["tsv"] for w in want["windows"])
        outs[tag] = {n: open(os.path.join(out, n)).read() for n in ("HGg.svision.s3.graph.vcf", "HGg.graph_exactly_match.txt")}
    assert outs["one"] == outs["three"]


@pytest.mark.gpu
def test_cli_batch_size_does_not_change_results(checkpoint, tmp_path):
    """--batch_size is the padding granule and the launch size (two batches per graph replay), never a result: 16, 32
    (fc tile 32x32), 64, 128 (the reference default) and an odd 200 give the same VCF."""
    from svision_amd.io import bam
    prefix, _params = checkpoint
    fasta = helpers.load_golden_fasta()
    fa = str(tmp_path / "genome.fa")
    bam.write_fasta(fa, {n: fasta._seq[n] for n in fasta.references})
    texts = {}
    for bs in (16, 32, 64, 128, 200):
        opts = cli.parse_arguments(["-o", str(tmp_path / ("o%d" % bs)), "-b", os.path.join(helpers.GOLDEN, "collect_small.bam"), "-m", prefix,
                                    "-g", fa, "-n", "S", "-s", "3", "--window_size", "60000", "--batch_size", str(bs)])
        texts[bs] = open(cli.run(opts)).read()
    assert len(set(texts.values())) == 1 and texts[64].count("\n") > 20


def test_network_adopted_from_the_weight_cache_equals_the_network_built_from_the_checkpoint(tmp_path, monkeypatch):
    """weight_cache.py: the second process on a checkpoint adopts the first one's device-layout weights and background
    activations from one blob -- same buffers, same predictions, bit for bit -- and lazily captured launch graphs give what
    eagerly captured ones give."""
    from oracle import alexnet_ref
    from svision_amd.network import predict, weight_cache
    from svision_amd.pipeline import DeviceStage
    monkeypatch.setenv("SVX_CACHE_DIR", str(tmp_path / "cache"))
    prefix = str(tmp_path / "m.ckpt")
    ck.write_checkpoint(prefix, alexnet_ref.random_params(seed=11))
    predict._MODEL_CACHE.clear()
    first = predict.load_network(prefix, device="cuda:0")            # reads the bundle, packs, leaves the blob behind
    blobs = os.listdir(tmp_path / "cache")
    assert len(blobs) == 1 and blobs[0].startswith("m.ckpt.svx-packed-")
    predict._MODEL_CACHE.clear()
    second = predict.load_network(prefix, device="cuda:0")           # adopts the blob
    assert second is not first and second._background is not None     # the backgrounds came with it (no kernel ran for them)
    a, b = dict(first.named_buffers()), dict(second.named_buffers())
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    for k, v in first.background().items():
        assert torch.equal(v, second._background[k]), k
    case = _case()
    bed = tmp_path / "chrB.bed"
    bed.write_text(case["chroms"]["chrB"]["tsv"])
    gen = BatchGenerator(str(bed), nb_classes=5, batch_size=64, layout="NCHW")
    records, _labels = gen.next_records(64)
    rec = records.contiguous()                                           # int32 device tensor [64, 12]
    p1, p2 = first.predict_records_packed(rec), second.predict_records_packed(rec)
    assert torch.equal(p1, p2)
    # lazily built launch slots (the command line) == eagerly built ones (bench, services)
    eager, lazy = DeviceStage(first, 64, "cuda:0", n_streams=2, launch_batches=2), DeviceStage(second, 64, "cuda:0", n_streams=2, launch_batches=2, lazy=True)
    assert all(len(s) == 2 for s in eager.slots) and all(len(s) == 0 for s in lazy.slots)
    big = rec.repeat(3, 1)                                               # 192 images: one launch of 128 and one of 64
    o1, o2 = torch.empty((192, 6), device="cuda:0"), torch.empty((192, 6), device="cuda:0")
    eager.run(big, o1)
    lazy.run(big, o2)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and torch.equal(o1[:64], p1[:, :6])
    assert sum(len(s) for s in lazy.slots) == 2                          # only the shapes it met
    predict._MODEL_CACHE.clear()
    monkeypatch.setenv("SVX_WEIGHT_CACHE", "0")
    third = predict.load_network(prefix, device="cuda:0")
    assert third._background is None and torch.equal(third.fc6_w, first.fc6_w)
    predict._MODEL_CACHE.clear()
