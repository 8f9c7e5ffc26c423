#!/usr/bin/env python3
"""bench.py -- candidate SV sites/sec (encode+CNN) on N MI355X of one node.

Workloads (synthetic: the reference ships no BAM; svision_amd/synth.py, fixed seeds):

  wg (default)    BASELINE.json configs[2] stand-in, the config the metric is quoted on: 24 contigs with the GRCh38
                  primary lengths (3.1 Gb, 322 collection windows), HiFi reads N(15 kb, 2 kb) at 30x with planted SVs,
                  chromosomes LPT-sharded over the ranks exactly as the command line does it
                  (svision_amd.dist.shard_chromosomes, cli.load_rank_table).  `--steps K` (default 20) = K windows PER RANK:
                  the job is exactly K x N windows -- every chromosome keeps a prefix, the windows apportioned to the
                  chromosomes by length (fewer than 24: the longest chromosomes, one window each) -- sharded by chromosome:
                  WEAK scaling (the path partitions, no data-path collective); `steps` in the line is K, `config.windows`
                  K x N.  Without --steps (`--resident`), or once K x N reaches 322, the job is the genome whatever N: strong.
  cfg2            BASELINE.json configs[1] stand-in: a chr21-sized contig (46,709,983 bp), same read model.  N > 1:
                  every rank owns its own such shard (weak scaling).

`value` is the FILE-INCLUSIVE rate (wall of Step 1 + Step 2 from a BAM on local disk, SVision:259-328): the job's
windows are written as a BAM + .bai during set-up (random bases, binned qualities: ~0.42 compressed bytes per base), and the
timed region starts with nothing but that file and ends after the cross-rank exchange:

  file    read + BGZF inflate + record packing, chromosome by chromosome   (device engine: svx_bgzf_inflate / svx_bam_walk_*;
                                                                            SVX_INGEST=cpu: libdeflate on host threads)
  device  CIGAR/segment scan of every chromosome                           (svx_cigar_scan)
  host    reads -> signatures -> clusters -> segment pairs                 (parity-tested mirror of src/collection)
  device  similarity-image encoding + AlexNet fp32, batches of 64 candidate images
  host    per-site vote -> VCF body lines + scores

A *step* is one collection window of the reference driver (10 Mb, SVision:88) -- on every rank at once under weak scaling.
`value` is the median of --repeats (3) timed repeats of the whole job, each between its own barriers.  value = candidate sites (distinct region
keys of the chromosomes' segment TSVs: a site spanning a window boundary counts once) per second, whole job; ms_per_step
follows it.  `config.resident_sites_per_s` is the same job with the alignments already decoded and resident in HBM (the
headline of rounds 1-3): what the device pipeline does once ingestion is out of the way.  No data-path collective; one
score-range all_reduce + one record gather at the end, inside the timed region.  Rank 0 prints ONE compact JSON line (< 4 KB:
the contract's keys, roofline, cpu_baseline, parity_check); the full record (every leg, decoder traces, per-kernel rooflines) goes
to --detail (bench_detail.json).  After the timed legs the windows the CPU baseline is about to run go through the device path
once more (untimed) with the helpers returning their TSV text and the owner keeping the predictions: `parity_check` compares them
with the CPU port's (TSV and site keys bit-exact, softmax within 1e-3) and the process exits 3 on a mismatch.
`--gpus N` without a launcher environment starts the N ranks itself (torch.distributed.run, one per GPU, RCCL).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from svision_amd import build_host, dist as sdist, kernels, synth  # noqa: E402
from svision_amd.io import bam  # noqa: E402
from svision_amd.network.alexnet import AlexNet, checkpoint_shapes  # noqa: E402
from svision_amd.pipeline import HelperPool, PooledHotPath  # noqa: E402
from svision_amd.sample import Sample  # noqa: E402

IMG_BYTES = 227 * 227 * 3 * 4 + 48            # 618,348 B written + 48 B read per image
CNN_FLOP = 1_440_662_592                      # FLOP per image of the dense network
HBM_PEAK = 8.0e12                             # MI355X_MICROARCH.md: HBM3E 8 TB/s
F32_MFMA_PEAK = 157.3e12                      # MI355X_MICROARCH.md: FP32 matrix peak (f32-in MFMA)
CHR21 = 46_709_983
CFG1_LEN = 75_000_000                        # cfg1 = one 75 Mb contig, HiFi 30x, -s 5
# GRCh38 primary assembly, chr1..chr22, chrX, chrY (header order of a GRCh38 BAM)
GRCH38 = (("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555), ("chr5", 181538259),
          ("chr6", 170805979), ("chr7", 159345973), ("chr8", 145138636), ("chr9", 138394717), ("chr10", 133797422),
          ("chr11", 135086622), ("chr12", 133275309), ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189),
          ("chr16", 90338345), ("chr17", 83257441), ("chr18", 80373285), ("chr19", 58617616), ("chr20", 64444167),
          ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415))
# dense FLOP per image of the layers behind the first one (2 x MAC) and their output pixels per image
LAYER_FLOP = {"conv2": 447_897_600, "conv3": 299_040_768, "conv4": 224_280_576, "conv5": 149_520_384, "fc": 109_092_864}
LAYER_PIX = {"conv2": 729, "conv3": 169, "conv4": 169, "conv5": 169}
WINDOW = 10_000_000
def _pmc_traffic():
    """The newest profiles/rNN_pmc_traffic.json: the round's rocprofv3 PMC passes over the device stage (tools/r06_profile.sh)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                return json.load(f)
        except (OSError, ValueError):
            continue
    return {}


TRAFFIC = _pmc_traffic()


def random_weights(seed=0):
    rng = np.random.default_rng(seed)
    p = {}
    for k, shp in checkpoint_shapes().items():
        if k.endswith("biases"):
            p[k] = (rng.standard_normal(shp) * 0.1).astype(np.float32)
        else:
            p[k] = (rng.standard_normal(shp) * np.sqrt(2.0 / np.prod(shp[:-1]))).astype(np.float32)
    p["conv1/weights"] *= np.float32(0.02)
    return p


def options_ns(batch):
    import types
    return types.SimpleNamespace(
        out_path=None, bam_path="<resident>", model_path=None, genome=None, sample="bench", thread_num=1, min_support=5,
        chrom=None, hash=False, qname=False, graph=False, contig=False, debug=False, min_mapq=10, min_sv_size=50,
        max_sv_size=1000000, window_size=WINDOW, patition_max_distance=5000, cluster_max_distance=0.3,
        batch_size=batch, min_gt_depth=4, homo_thresh=0.8, hete_thresh=0.2, k_size=10, min_accept=50, max_hash_len=1000)


def windows_of(name, length):
    return [(name, pos, min(length, pos + WINDOW)) for pos in range(0, length, WINDOW)]


def job_contigs(steps):
    """The contigs of a `wg` job of exactly `steps` windows (None / >= 322: the whole genome): a prefix of every
    chromosome, windows apportioned by length (largest remainder, at least one each); fewer steps than chromosomes: the
    `steps` longest chromosomes, one window each.  Header order is kept."""
    full = [(n, l, len(windows_of(n, l))) for n, l in GRCH38]
    total = sum(w for _n, _l, w in full)
    if not steps or steps >= total:
        return list(GRCH38)
    if steps < len(full):
        keep = {n for n, _l, _w in sorted(full, key=lambda t: -t[1])[:steps]}
        return [(n, min(l, WINDOW)) for n, l, _w in full if n in keep]
    quota = [w * steps / total for _n, _l, w in full]
    k = [max(1, int(q)) for q in quota]
    order = sorted(range(len(full)), key=lambda i: -(quota[i] - int(quota[i])))
    i = 0
    while sum(k) < steps:                                     # largest remainders first
        j = order[i % len(order)]
        if k[j] < full[j][2]:
            k[j] += 1
        i += 1
    while sum(k) > steps:                                     # the one-window minimum overshot: take from the longest
        j = max(range(len(k)), key=lambda t: k[t])
        k[j] -= 1
    return [(n, min(l, kk * WINDOW)) for (n, l, _w), kk in zip(full, k)]


def wg_job(steps, world):
    """The `wg` job of an N-rank run: --steps K = K windows PER RANK -> exactly K x N windows of the genome (job_contigs), weak
    scaling; no --steps, or K x N >= the genome's 322 windows: the genome whatever N, strong scaling.  -> (contigs, strong?)."""
    full = sum(len(windows_of(n, l)) for n, l in GRCH38)
    want = steps * world if steps else None
    return job_contigs(want), (want is None or want >= full)


def _simulate_contig(job):
    """job: dict(name, length, coverage, seed, kind, e2e=(tid in the rank's file, prefix length) or None) ->
    (AlignmentTable, genome bytes, BGZF segment of the prefix or None) of one contig (runs in a forked worker)."""
    name, length, coverage, seed, kind = job["name"], job["length"], job["coverage"], job["seed"], job.get("kind")
    if kind == "contig":                     # assembly-vs-reference stand-in: two haplotypes of ~2 Mb contigs, 0.1 % small events
        cfg = synth.SimConfig(contigs=[(name, length)], coverage=coverage, seed=seed, read_len_mean=2_000_000, read_len_sd=600_000,
                              err_rate=0.001)
    elif kind == "ont":                      # ONT ultra-long stand-in: log-normal lengths (median 50 kb), 5 % small events
        cfg = synth.SimConfig(contigs=[(name, length)], coverage=coverage, seed=seed, read_len_mean=50_000, lognormal=True,
                              lognormal_sigma=0.7, err_rate=0.05)
    else:
        cfg = synth.SimConfig(contigs=[(name, length)], coverage=coverage, seed=seed)
    table, genome, _svs = synth.simulate(cfg)
    segment = None
    if job.get("e2e") is not None:           # the records of the e2e leg's prefix as one compressed BGZF segment (realistic SEQ / QUAL)
        tid, prefix = job["e2e"]
        part = table if prefix >= length else table.subset(np.flatnonzero(table.pos < prefix))
        part.tid[:] = tid
        segment = bam.encode_reference_segment(part, seq="random", seed=seed)
        if part is table:
            table.tid[:] = 0
        if job.get("spill_dir"):                 # whole-genome files: the compressed bytes wait on disk, not in the parent's memory (62 GB)
            path = os.path.join(job["spill_dir"], "segment_%s.bin" % name)
            with open(path, "wb") as f:
                f.write(segment["data"])
            segment["data_path"], segment["data"] = path, None
    return table, genome[name], segment


def build_workload(args, rank, world, cores):
    """-> (parts: (name, length, table, genome bytes) per contig of this rank, windows of this rank, strong?, windows of the
    whole job, e2e = dict(path of this rank's BAM, windows, bytes) or None)."""
    e2e_prefix = {}
    if args.workload in ("cfg1", "cfg2", "ont"):
        # one contig per rank (weak scaling): chr21-sized (cfg2, ont) or the 75 Mb contig BASELINE.md section 2 defines as the
        # stand-in of the reference's missing demo BAM (cfg1; `-s 5` is options_ns' min_support)
        name = "contig75" if args.workload == "cfg1" else "chr21"
        jobs = [dict(name=name, length=args.contig_len, coverage=args.coverage, seed=1 + rank, kind="ont" if args.workload == "ont" else None)]
        strong = False
        total_windows = None
        if args.e2e_windows:
            e2e_prefix = {name: min(args.contig_len, args.e2e_windows * WINDOW)}
    else:
        # wg with --steps K: K windows PER RANK -- the job is K x N windows of the genome (a prefix of every chromosome), sharded by
        # chromosome: weak scaling, as a path that partitions with no data-path collective is to be reported.  Without --steps (or
        # once K x N reaches the genome's 322 windows) the job is the genome whatever N: strong scaling.
        wg_contigs, wg_strong = wg_job(args.steps, world)
        contigs = list(GRCH38) if args.workload == "contig" else wg_contigs
        shards = sdist.shard_chromosomes([n for n, _l in contigs], [l for _n, l in contigs], world)
        shard = shards[rank]
        length_of = dict(contigs)
        args.rank_mb = [sum(length_of[c] for c in sh) / 1e6 for sh in shards]               # the LPT loads, for the report
        jobs = [dict(name=n, length=l, coverage=2.0 if args.workload == "contig" else args.coverage, seed=100 + i,
                     kind="contig" if args.workload == "contig" else None) for i, (n, l) in enumerate(contigs) if n in shard]
        strong = args.workload == "contig" or wg_strong
        total_windows = len(contigs) if args.workload == "contig" else sum(len(windows_of(n, l)) for n, l in contigs)
        if args.e2e_windows and args.workload == "contig":
            e2e_prefix = {n: length_of[n] for n in shard}               # --contig: one task per chromosome, the file holds all of them
        if args.e2e_windows and args.workload == "wg":
            # the file-inclusive leg runs on a bounded job: the same chromosomes, every one cut to its share of --e2e-windows
            small = dict(job_contigs(args.e2e_windows * world)) if total_windows > args.e2e_windows * world else length_of
            e2e_prefix = {n: min(small[n], length_of[n]) for n in shard if n in small}
    k = 0
    bam_dir = None
    if e2e_prefix:
        import tempfile
        bam_dir = tempfile.mkdtemp(prefix="svx_bench_bam_", dir=args.bam_dir)
    for j in jobs:
        if j["name"] in e2e_prefix:
            j["e2e"] = (k, e2e_prefix[j["name"]])
            if sum(e2e_prefix.values()) > 1_200_000_000:                  # > ~24 GB of file: see _simulate_contig
                j["spill_dir"] = bam_dir
            k += 1
    if len(jobs) > 1:
        import multiprocessing as mp
        pool = mp.get_context("fork").Pool(min(len(jobs), args.sim_procs or max(1, cores // world)))     # before the first HIP call
        try:
            made = pool.map(_simulate_contig, jobs, chunksize=1)
        finally:
            # close + join, not the context manager's terminate(): under rocprofv3 a forked worker inherits the tool's
            # SIGTERM handler, which does not let it die, and the parent then waits for it for ever
            pool.close()
            pool.join()
    else:
        made = [_simulate_contig(j) for j in jobs]
    parts = [(j["name"], j["length"], t, g) for j, (t, g, _seg) in zip(jobs, made)]
    if args.workload == "contig":             # --contig: one task per chromosome (SVision:161-180)
        windows = [(name, 0, length) for name, length, _t, _g in parts]
    else:
        windows = [w for name, length, _t, _g in parts for w in windows_of(name, length)]
    e2e = None
    if e2e_prefix:
        names = [j["name"] for j in jobs if "e2e" in j]
        d = bam_dir
        path = os.path.join(d, "rank%d.bam" % rank)
        segs = [seg for _t, _g, seg in made if seg is not None]
        bam.write_bam_segments(path, names, [dict((j["name"], j["length"]) for j in jobs)[n] for n in names], segs, index=True)
        e2e = {"path": path, "dir": d, "references": names,
               "windows": [(n, 0, e2e_prefix[n]) for n in names] if args.workload == "contig" else [w for n in names for w in windows_of(n, e2e_prefix[n])],
               "bytes": os.path.getsize(path), "inflated": sum(s["inflated"] for s in segs)}
    return parts, windows, strong, total_windows, e2e


MAX_FILE_WINDOWS = 100                        # 19 GB of BAM (46 GB inflated) written during set-up: what a default-sized run may cost (an explicit --e2e-windows may ask for more)


def resolve_defaults(args):
    """--steps / --e2e-windows / --contig-len defaults per workload (see the module docstring)."""
    if args.workload == "cfg1" and args.contig_len == CHR21:
        args.contig_len = CFG1_LEN
    if args.workload == "wg" and args.steps is None and not args.resident:
        args.steps = 20
    if args.e2e_windows is None:
        if args.workload == "wg" and not args.resident:
            args.e2e_windows = min(args.steps, MAX_FILE_WINDOWS)
        elif args.workload in ("cfg1", "cfg2", "ont"):
            args.e2e_windows = min(-(-args.contig_len // WINDOW), 20)
        else:
            args.e2e_windows = 20
    return args


def median_leg(legs):
    """The repeat with the median wall time (upper median of an even count).  `seconds` is already the max over the ranks, so
    every rank picks the same repeat."""
    order = sorted(range(len(legs)), key=lambda i: legs[i]["seconds"])
    return legs[order[len(order) // 2]]


def repeats_summary(legs, rate):
    """min / median / max of the repeats' rates (`rate`: leg -> sites/s) and their wall seconds, in run order."""
    vals = sorted(rate(l) for l in legs)
    return {"n": len(legs), "min": vals[0], "median": rate(median_leg(legs)), "max": vals[-1],
            "seconds": [round(l["seconds"], 4) for l in legs]}


def _clip(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


LINE_LIMIT = 4096                             # bytes of the stdout line (a 29 KB line was not parsed by the driver)


def compact_line(full, detail_path=None):
    """The ONE stdout line: the contract's keys, the roofline and cpu_baseline objects, the parity verdict -- numbers only, every
    string clipped -- built from the full record (which goes to --detail).  Always < LINE_LIMIT bytes and strict JSON
    (no NaN / Infinity: such a value becomes null)."""
    def num(v):
        if isinstance(v, (bool, type(None), str)):
            return v
        if isinstance(v, (int, np.integer)):
            return int(v)
        v = float(v)
        return v if np.isfinite(v) else None

    cfg, rf = full.get("config", {}), full.get("roofline", {})
    line = {k: num(full.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                          "scaling", "vs_baseline", "dtype", "data")}
    if full.get("value_repeats"):
        line["value_repeats"] = {k: ([num(x) for x in v] if isinstance(v, list) else num(v)) for k, v in full["value_repeats"].items()}
    e2e = full.get("e2e") or {}
    line["config"] = {
        "workload": _clip(cfg.get("workload", ""), 240),
        "timed_region": _clip(cfg.get("timed_region_kind") or cfg.get("timed_region", ""), 160),
        "batch": num(cfg.get("batch")), "windows": num(cfg.get("windows", full.get("steps"))),
        "sites_per_step": num(cfg.get("sites_per_step")), "images_per_site": num(cfg.get("images_per_site")),
        "images_per_s": num(cfg.get("images_per_s")),
        "resident_sites_per_s": num(cfg.get("resident_sites_per_s")),
        "file_inclusive_over_resident": num(cfg.get("file_inclusive_over_resident")),
        "bam_bytes": num(e2e.get("bam_bytes")), "ingest_engine": e2e.get("ingest_engine"),
        "host_cores": num(cfg.get("host_cores")), "host_cpus_pinned": num(cfg.get("host_cpus_pinned")), "host_workers_per_rank": num(cfg.get("host_workers_per_rank")),
        "rccl_world": num(cfg.get("rccl_world")), "imbalance": num(cfg.get("imbalance")),
        "parallelism": _clip(cfg.get("parallelism", ""), 140),
    }
    cold = full.get("e2e_cold_cache") or {}
    if cold:
        line["config"]["cold_page_cache_sites_per_s"] = num(cold.get("value"))
    line["roofline"] = {"kernel": _clip(rf.get("kernel_short") or rf.get("kernel", ""), 200)}
    for k in ("bound", "achieved", "peak", "unit", "frac", "frac_stage_alone", "frac_algorithmic", "dense_stage_frac", "traffic",
              "ms_per_batch", "device_busy_frac"):
        line["roofline"][k] = num(rf.get(k))
    dom = rf.get("dominant_kernel")
    if dom:
        line["roofline"]["dominant_kernel"] = {k: (num(v) if not isinstance(v, str) else _clip(v, 60)) for k, v in dom.items()}
    hbm = full.get("roofline_hbm")
    if hbm:
        line["roofline_hbm"] = {k: num(v) if not isinstance(v, str) else _clip(v, 120) for k, v in hbm.items()}
    cpu = full.get("cpu_baseline")
    if cpu:
        line["cpu_baseline"] = {"value": num(cpu.get("value")), "unit": cpu.get("unit"), "cores": num(cpu.get("cores")), "kind": cpu.get("kind"),
                                "sample": _clip(cpu.get("sample_short") or cpu.get("sample", ""), 260)}
    par = full.get("parity_check")
    if par:
        line["parity_check"] = {k: (num(v) if not isinstance(v, (list, dict)) else v) for k, v in par.items() if k != "per_window"}
    if detail_path:
        line["detail"] = detail_path
    text = json.dumps(line, allow_nan=False)
    if len(text) >= LINE_LIMIT:                   # cannot happen with the clips above; if it ever does, the strings go first
        for obj, key in ((line["config"], "workload"), (line["roofline"], "kernel"), (line.get("cpu_baseline", {}), "sample"), (line["config"], "parallelism")):
            if key in obj:
                obj[key] = _clip(obj[key], 60)
        text = json.dumps(line, allow_nan=False)
    assert len(text) < LINE_LIMIT, len(text)
    return text


def rank_command(n_ranks, port, argv):
    """The launcher line of `python bench.py --gpus N` without a launcher: N ranks of this script on one node, rendezvous on
    127.0.0.1 (the container's host name may not resolve), the caller's own arguments passed through."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def rank_environment(environ):
    """dmabuf IPC between the ranks' HIP contexts (the host driver supports nothing else: RCCL fails with hipIpcGetMemHandle otherwise)."""
    return dict(environ, HSA_ENABLE_IPC_MODE_LEGACY=environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: run the N ranks under torch.distributed.run and pass rank 0's
    JSON line through."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    return subprocess.call(rank_command(args.gpus, port, sys.argv[1:]), env=rank_environment(os.environ))


def first_contact(world, grouped, backend, rank_mb, rank_file_bytes, identities):
    """What an N-GPU run says about its own layout before any rate is believed (the line's `config.first_contact`; rank 0 also
    prints it to stderr): the size of the RCCL world, the device every rank resolved to, the LPT loads and their imbalance --
    chromosomes are not divisible, so max / mean of the loads bounds the strong-scaling speed-up at N / imbalance before
    any other loss (the 24 GRCh38 chromosomes on 8 ranks: 1.036 -> at most 7.72 x of 8) -- and the bytes of BAM every rank reads."""
    imbalance = (max(rank_mb) / (sum(rank_mb) / len(rank_mb))) if rank_mb else None
    dup = sdist.duplicate_devices(identities)
    return {"rccl_world": world if grouped and backend == "nccl" else 0, "world": world, "backend": backend if grouped else None,
            "devices": ["%s %s" % tuple(i) for i in identities], "ranks_sharing_a_device": sorted(r for v in dup.values() for r in v),
            "rank_mb": [round(v, 1) for v in rank_mb] if rank_mb else None, "imbalance": imbalance,
            "strong_scaling_cap": (world / imbalance) if imbalance else None,
            "rank_bam_bytes": [int(v) for v in rank_file_bytes] if rank_file_bytes else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="windows per rank (wg: default 20 -- the job is steps x ranks windows of the genome, sharded by chromosome; with --resident and no --steps all 322 windows whatever the ranks; cfg2 / ont: default 200)")
    ap.add_argument("--resident", action="store_true", help="`value` = the resident leg (alignments decoded and in HBM before the timed region; the headline of rounds 1-3): for jobs whose BAM would be too large to write during set-up (the whole genome: 62 GB)")
    ap.add_argument("--no-other-engine", action="store_true", help="skip the file-inclusive leg with the other ingest engine")
    ap.add_argument("--e2e-sweep", default=None, help="experiments: further file-inclusive legs in the same process, one per ';'-separated set of "
                                                      "comma-separated NAME=VALUE environment settings (e.g. 'SVX_STAGE_SLOTS=8;SVX_STAGE_SLOTS=8,SVX_PIPE_GROUP_MB=256'); "
                                                      "a set may be repeated; their seconds go to stderr and to the line's `e2e_sweep`")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=("cfg1", "cfg2", "wg", "ont", "contig"), default="wg",
                    help="wg: 24 GRCh38-length contigs sharded over the ranks (the default: the config the metric is quoted on; --steps K = K windows per rank, weak scaling; without --steps the whole genome, strong scaling); "
                         "cfg2: chr21-sized HiFi sample per rank (weak scaling); ont: chr21-sized ONT ultra-long stand-in per rank (BASELINE configs[3] stress: ~5,000 CIGAR ops per read); "
                         "contig: --contig mode, two haplotypes of ~2 Mb assembly contigs on the 24 GRCh38-length chromosomes, one task per chromosome")
    ap.add_argument("--batch", type=int, default=64, help="candidate images per CNN batch (BASELINE configs[1])")
    ap.add_argument("--contig-len", type=int, default=CHR21)
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--workers", type=int, default=0, help="helper processes per rank for the Python host glue (default: 8, capped at twice the usable CPUs per rank: a helper waits for its window's predictions between collection and vote)")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams the per-batch graphs are replayed on")
    ap.add_argument("--inflight", type=int, default=6, help="windows enqueued on the device at once")
    ap.add_argument("--launch-batches", type=int, default=4, help="batches of --batch images per device launch (graph replay)")
    ap.add_argument("--e2e-windows", type=int, default=None,
                    help="windows of the file-inclusive leg: the job cut to that many windows is written as a BAM (+ .bai, random bases, binned "
                         "qualities) to local disk during set-up and run from the file.  Default: the job's own windows (wg: --steps, at most 100 "
                         "= a 19 GB file); 0 = no file leg (`value` is then the resident leg)")
    ap.add_argument("--decode-threads", type=int, default=0, help="inflate threads of the e2e leg per rank (default: cores / ranks - helpers - 2, at most 128)")
    ap.add_argument("--bam-dir", default=None, help="where the synthetic BAM of the e2e leg is written (default: the temp directory)")
    ap.add_argument("--sim-procs", type=int, default=0, help="processes that simulate + compress the workload during set-up (default: the usable CPUs per rank; "
                                                             "fewer for the whole-genome file: a process holds a chromosome's records, inflated and compressed)")
    ap.add_argument("--no-cold-leg", action="store_true", help="skip the file-inclusive leg with the BAM's pages dropped from the page cache first (`e2e_cold_cache`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-calibration", action="store_true", help="skip the per-kernel timings outside the timed region")
    ap.add_argument("--repeats", type=int, default=3, help="timed repeats of the job (each one times exactly --steps windows between its own barriers); "
                                                           "`value` / `ms_per_step` are the repeat with the MEDIAN wall time, value_repeats holds min / median / max")
    ap.add_argument("--detail", default="bench_detail.json", help="where the full record goes (decoder traces, per-kernel rooflines, the other legs, notes); "
                                                                  "stdout carries ONE compact JSON line (< 4 KB) whose numbers are a subset of it")
    ap.add_argument("--host-cpus", type=int, default=0, help="pin this rank and everything it forks or starts afterwards (host helpers, pread / decode threads, the runtime's own threads) "
                                                             "to K CPUs (sched_setaffinity after the workload is built, before the first fork): the host budget an N-rank run on "
                                                             "this box leaves a rank (VERDICT r5 item 5).  0: no pinning")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity leg (the device path's TSV / site keys / softmax against the CPU port's on the windows the CPU baseline runs)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    resolve_defaults(args)

    rank, world = sdist.env_rank()
    from svision_amd.ingest import decode_threads, effective_cpus
    cores, visible_cpus = effective_cpus()        # the cgroup's CPU-time quota, not the CPUs the container merely sees
    workers = args.workers if args.workers > 0 else max(1, min(8, 2 * cores // world))
    # ---- untimed set-up.  Order matters: everything that forks (simulation pool, CPU-baseline pool, host helpers)
    # happens before the first HIP call (forking with a live GPU context makes the driver evict / restore the queues)
    parts, windows, strong, total_windows, e2e = build_workload(args, rank, world, cores)
    if args.host_cpus > 0:
        # the K CPUs of this rank: disjoint sets per rank, taken from the CPUs the process may run on
        allowed = sorted(os.sched_getaffinity(0))
        k = min(args.host_cpus, len(allowed))
        os.sched_setaffinity(0, set(allowed[(rank * k) % len(allowed):][:k]) or set(allowed[:k]))
        cores, visible_cpus = effective_cpus()
        workers = args.workers if args.workers > 0 else max(1, min(8, 2 * cores // world))
    opts = options_ns(args.batch)
    if args.workload == "contig":             # SVision:161-180, collect_signatures.py:125
        opts.contig, opts.min_support = True, 1
    table = bam.concat_tables([t for _n, _l, t, _g in parts]) if len(parts) > 1 else parts[0][2]
    fasta = bam.Fasta(sequences={n: g for n, _l, _t, g in parts})
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    cpu_pool = CpuBaselinePool(cores, table, fasta, opts, visible_cpus) if want_cpu else None
    pool = HelperPool(workers, opts, table=table, fasta=fasta)
    torch.cuda.set_device(sdist.local_device_index())
    dev = torch.device("cuda", torch.cuda.current_device())
    rank, world = sdist.init_from_env()
    sample = Sample.from_table(table, fasta, opts.min_sv_size, device=dev)
    pool.attach_scan(sample)
    if e2e is not None:
        # The file-inclusive legs run with a warm device allocator, as the resident leg runs after --warmup steps: the first large
        # hipMalloc of a process on a box whose previous process has just exited costs 0.15-0.35 s (nothing else in the process
        # gets a HIP call through meanwhile); one block of the size the legs' buffers add up to is allocated here and stays in
        # the caching allocator, which cuts the legs' buffers out of it.
        # (compressed bytes x 2.4 inflated x 1.5 for the two-kernel inflate's sequence streams + packed arrays, several groups in flight)
        # (peak of the 20-window job: compressed 3.9 GB + inflated 9.3 GB + the two-kernel inflate's sequence streams of the groups
        # in flight 13 GB + packed arrays: ~27 GB.  The block serves allocations made on the default stream: the decoder takes its
        # large buffers there, ingest_gpu.launch)
        torch.empty(min(48 << 30, 8 * int(e2e["bytes"])), dtype=torch.uint8, device=dev)
    net = AlexNet(random_weights(0), device=dev)
    net.executed = torch.zeros(5, dtype=torch.int64, device=dev)     # executed conv pixels per layer + images, summed on the device
    hot = PooledHotPath(sample, opts, net, device=dev, n_streams=args.streams, max_inflight=args.inflight, launch_batches=args.launch_batches, pool=pool)
    hot.record_timing = True                  # HIP events around every launch, on the launch's stream (roofline.achieved)

    import torch.distributed as tdist
    grouped = tdist.is_available() and tdist.is_initialized()

    def sync_all():
        if grouped:
            tdist.barrier()
        torch.cuda.synchronize()

    from svision_amd.pipeline import distinct_sites, stitch_windows

    def run(seq, rescan=True, sink=None):
        """seq: windows in task order (a chromosome's windows contiguous).  Windows complete in any order; a run of
        consecutive windows of one chromosome is one per-chromosome vote (sites spanning a window boundary are written once)."""
        sites = images = records = 0
        scores = []
        hot.reset_timing()
        done = {}
        sample = lambda chrom, start: hot.feed.get(chrom, block=True, start=start)[1]     # noqa: E731  (the resident Sample, or the one that served the window)
        # maximal runs of ascending windows of one chromosome; a run is voted and stitched as soon as its last window is through
        # (as the command line writes a chromosome when its last window is done, cli._run_pooled), while the device works on the others
        runs, run_of, lo = [], {}, 0
        while lo < len(seq):
            hi = lo + 1
            while hi < len(seq) and seq[hi][0] == seq[hi - 1][0] and seq[hi][1] == seq[hi - 1][2]:
                hi += 1
            for w in range(lo, hi):
                run_of[w] = len(runs)
            runs.append([lo, hi, hi - lo])
            lo = hi
        for res in hot.run_windows(seq, rescan=rescan):
            images += res.n_images
            done[res.wid] = res
            if sink is not None:                                                  # the parity leg: the window's TSV text and predictions
                sink[(res.chrom, res.start, res.end)] = (res.tsv, res.classes, res.probs)
            run = runs[run_of[res.wid]]
            run[2] -= 1
            if run[2] == 0:
                part = [done.pop(w) for w in range(run[0], run[1])]
                sites += distinct_sites(part)                                  # a site spanning a window boundary counts once
                for vcf_text, score_text in stitch_windows(part, opts, sample).values():
                    records += vcf_text.count("\n")
                    scores += [float(s_) for s_ in score_text.split()]
        return sites, images, records, scores

    sharded = args.workload in ("wg", "contig")                       # the rank runs its LPT shard of one job (the others: a job of its own per rank)
    if sharded:
        timed = list(windows)                                         # this rank's share of the job
        steps_job = total_windows
    else:
        k = args.steps or (len(windows) if args.workload == "cfg1" else 200)
        timed = [windows[i % len(windows)] for i in range(k)]
        steps_job = k
    if windows:
        run([windows[i % len(windows)] for i in range(args.warmup)])
    if e2e is not None and not args.resident and args.warmup > 0:
        # ... and the warm-up steps of the file-driven path: the first --warmup windows' chromosomes FROM THE FILE, untimed --
        # the ingest kernels' code objects, the pinned staging ring, the caching allocators' first blocks (what a process that
        # has read a file before finds in place)
        warm = dict(e2e, header_references=e2e["references"])
        warm["references"] = []
        for w in e2e["windows"]:
            if w[0] not in warm["references"] and len(warm["references"]) < args.warmup:
                warm["references"].append(w[0])
        warm["windows"] = [w for w in e2e["windows"] if w[0] in warm["references"]][:max(args.warmup, len(warm["references"]))]
        run_from_file(args, warm, hot, run, fasta, opts, dev, sync_all, cores, world, workers, grouped, engine=os.environ.get("SVX_INGEST", "auto"), keep=True)
    sync_all()

    def reduce_sum_max(values):
        t = torch.tensor(values, dtype=torch.float64, device=dev)
        if not grouped:
            return t.cpu().numpy(), t.cpu().numpy()
        tmax = t.clone()
        tdist.all_reduce(t, op=tdist.ReduceOp.SUM)
        tdist.all_reduce(tmax, op=tdist.ReduceOp.MAX)
        return t.cpu().numpy(), tmax.cpu().numpy()

    def stage_stats():
        """What the device stage did since net.executed was zeroed / the timing was reset, summed over the ranks:
        executed conv pixels per layer, images, device milliseconds (union of the launches' HIP-event intervals)."""
        executed = net.executed.cpu().numpy().astype(np.float64)     # [conv2, conv3, conv4, conv5 pixels, images]
        tot, _ = reduce_sum_max([hot.device_busy_ms(), hot.device_images] + executed.tolist())
        return {"dev_ms": float(tot[0]), "dev_images": float(tot[1]), "pix": tot[2:6], "images": max(float(tot[6]), 1.0)}

    # ---- the timed region of `value`: the job from its BAM (file-inclusive).  Its own barrier + synchronize on both sides,
    # max over ranks (run_from_file).
    e2e_block = e2e_other = None
    e2e_legs = []
    if e2e is not None and not args.resident:
        for _r in range(max(1, args.repeats)):
            net.executed.zero_()
            leg = run_from_file(args, e2e, hot, run, fasta, opts, dev, sync_all, cores, world, workers, grouped, engine=os.environ.get("SVX_INGEST", "auto"), keep=True)
            leg["stage"] = stage_stats()
            e2e_legs.append(leg)
        e2e_block = median_leg(e2e_legs)
    # ---- the same job with the alignments decoded and resident in HBM (the headline of rounds 1-3; --resident: `value`)
    res_legs = []
    for _r in range(max(1, args.repeats)):
        sync_all()
        net.executed.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sites, images, records, scores = run(timed)
        # the single cross-shard exchange of the job: score range + record gather (dist.py)
        sdist.exchange_score_range(scores)
        sdist.gather_texts({"rank%d" % rank: "%d records" % records})
        sync_all()
        dt = time.perf_counter() - t0
        if os.environ.get("SVX_TIMING") and rank == 0:
            print("owner thread seconds over %.3f s: %s" % (dt, {k: round(v, 3) for k, v in getattr(hot, "owner_profile", {}).items()}), file=sys.stderr)
        st = stage_stats()
        tot, tmax = reduce_sum_max([sites, images, dt])
        res_legs.append({"seconds": float(tmax[2]), "sites": float(tot[0]), "images": float(tot[1]), "stage": st})
    res_leg = median_leg(res_legs)
    res_stage, res_sites, res_images, dt = res_leg["stage"], res_leg["sites"], res_leg["images"], res_leg["seconds"]
    sweep = []
    if e2e is not None and args.e2e_sweep:
        for spec in args.e2e_sweep.split(";"):
            env = dict(kv.split("=", 1) for kv in spec.split(",") if "=" in kv)
            saved = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                leg = run_from_file(args, e2e, hot, run, fasta, opts, dev, sync_all, cores, world, workers, grouped, engine=os.environ.get("SVX_INGEST", "auto"), keep=True)
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            dd = leg["rank0_feed"].get("device_decoder", {})
            sweep.append({"env": env, "seconds": leg["seconds"], "read_s": dd.get("read_s"), "pread_s": dd.get("pread_s"), "slot_wait_s": dd.get("slot_wait_s"),
                          "last_ready_s": leg["rank0_feed"].get("last_ready_s"), "device_busy_frac": leg["device_busy_frac"], "trace": dd.get("trace"),
                          "cnn_span_ms": leg.get("cnn_span_ms_rank0"), "cnn_gaps_ms": leg.get("cnn_gaps_ms_rank0")})
            if rank == 0:
                print("e2e sweep %s: %.3f s (read %s, pread %s, slot wait %s, last chromosome ready %s)" % (
                    spec or "(default)", leg["seconds"], dd.get("read_s"), dd.get("pread_s"), dd.get("slot_wait_s"), leg["rank0_feed"].get("last_ready_s")), file=sys.stderr)
    e2e_cold = None
    if e2e is not None and not args.resident and not args.no_cold_leg:
        # the same leg on a cold file: the BAM's pages are written back and dropped from the page cache first ("BAM on
        # local NVMe"; `value` reads it from the page cache it was written through).  Reported beside `value`, never as `value`.
        e2e_cold = run_from_file(args, e2e, hot, run, fasta, opts, dev, sync_all, cores, world, workers, grouped, engine=os.environ.get("SVX_INGEST", "auto"), keep=True, cold=True)
    if e2e is not None and not args.no_other_engine:
        # the file-inclusive leg with the other ingest engine, reported beside it (default engine: BGZF inflate + record packing
        # on the device; the other: libdeflate on the host's threads)
        first = e2e_block["ingest_engine"] if e2e_block is not None else None
        other = os.environ.get("SVX_INGEST", "auto") if first is None else ("cpu" if first == "gpu" else "gpu")
        e2e_other = run_from_file(args, e2e, hot, run, fasta, opts, dev, sync_all, cores, world, workers, grouped, engine=other, keep=True)

    # the N-GPU first-contact checklist (every rank takes part in the two small gathers)
    file_bytes = [int(e2e["bytes"]) if e2e is not None else 0]
    if grouped:
        gathered = [None] * world
        tdist.all_gather_object(gathered, file_bytes[0])
        file_bytes = gathered
    contact = first_contact(world, grouped, tdist.get_backend() if grouped else None, getattr(args, "rank_mb", None), file_bytes, sdist.gather_identities())
    if rank == 0 and world > 1:
        print("first contact: %s" % json.dumps(contact), file=sys.stderr)
    B = args.batch
    headline_e2e = e2e_block is not None
    stage = e2e_block["stage"] if headline_e2e else res_stage       # the device stage over the timed region of `value`
    job_s = e2e_block["seconds"] if headline_e2e else dt
    job_sites = e2e_block["sites"] if headline_e2e else res_sites
    job_images = e2e_block["images"] if headline_e2e else res_images
    # windows of the whole job (all ranks), and the line's `steps`: strong scaling -- the job's windows; weak -- the windows of ONE rank
    # (a step = one window on every rank at once; value = the sites of all ranks / the slowest rank's time)
    job_windows = e2e_block["windows"] if headline_e2e else (steps_job if sharded else steps_job * world)
    job_steps = job_windows if strong else max(1, int(round(job_windows / world)))

    def stage_report(st, wall):
        frac = {k: float(st["pix"][i] / (st["images"] * LAYER_PIX[k])) for i, k in enumerate(("conv2", "conv3", "conv4", "conv5"))}
        executed_flop = sum(LAYER_FLOP[k] * frac[k] for k in frac) + LAYER_FLOP["fc"]      # per image, every timed batch counted
        dev_s = st["dev_ms"] * 1e-3 / world                           # mean device time per rank
        return {"frac": frac, "executed_flop": executed_flop, "dev_s": dev_s,
                "executed_tflops": executed_flop * st["dev_images"] / world / max(dev_s, 1e-9) / 1e12,
                "algorithmic_tflops": CNN_FLOP * st["dev_images"] / world / max(dev_s, 1e-9) / 1e12,
                "ms_batch": st["dev_ms"] / max(st["dev_images"] / B, 1), "batches": st["dev_images"] / B, "busy": dev_s / max(wall, 1e-9)}
    rep = stage_report(stage, job_s)
    rep_res = stage_report(res_stage, dt)
    calib = kernel_calibration(hot, sample, net, dev, B * max(1, args.launch_batches), windows[0]) if rank == 0 and not args.no_calibration and windows else {}
    dense = calib.get("device_stage_eager_1_stream_dense_convolutions", {})
    workload = (("cfg5 stand-in: --contig mode, two haplotypes of N(2 Mb, 0.6 Mb) assembly contigs (0.1 % small events) on 24 "
                 "chromosomes of GRCh38 length, one task per chromosome, chromosomes LPT-sharded over the ranks")
                if args.workload == "contig" else
                ("cfg1 stand-in (BASELINE.md section 2: the demo BAM is absent): one 75 Mb contig, synthetic HiFi N(15kb,2kb) reads, %gx, -s 5"
                 % args.coverage) if args.workload == "cfg1" else
                ("cfg3 stand-in: synthetic whole-genome HiFi, 24 contigs of GRCh38 length (N(15kb,2kb) reads, %gx), "
                 "chromosomes LPT-sharded over the ranks%s" % (args.coverage, "" if strong else "; --steps windows per rank (a prefix of every chromosome)"))
                if args.workload == "wg" else
                ("cfg4 stand-in on one contig per rank: synthetic ONT ultra-long chr21 (%d bp, log-normal reads, median 50 kb, "
                 "5 %% small events, %gx)" % (args.contig_len, args.coverage)) if args.workload == "ont" else
                ("cfg2 stand-in: synthetic HiFi chr21 (%d bp, N(15kb,2kb) reads, %gx)" % (args.contig_len, args.coverage)))
    line = {
        "metric": "candidate SV sites/sec (encode+CNN)",
        "value": job_sites / job_s,
        "unit": "sites/s",
        "n_gpus": world,
        "steps": job_steps,
        "warmup": args.warmup,
        "ms_per_step": job_s / max(job_steps, 1) * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload,
                   "timed_region": ("file-inclusive (SURVEY 8(d), SVision:259-328): from opening the job's BAM on local disk (%s bytes, BGZF, random bases + "
                                    "binned qualities) -- read, BGZF inflate + record packing (ingest engine: %s), device scan, collection, encode + "
                                    "CNN, vote -- to the end of the cross-rank exchange" % (e2e_block["bam_bytes"], e2e_block["ingest_engine"]))
                                   if headline_e2e else "resident: alignments decoded and in HBM before the timed region (--resident, or a workload without a file leg)",
                   "warm_state": ("`value` is measured in a process that has run --warmup windows through both paths before: the caching allocator holds one block "
                                  "of min(48 GB, 8 x the file) allocated during set-up, the pinned staging ring and the kernels' code objects exist, the guess of the slices' margin "
                                  "(~7 ms of host zlib per file) is cached, and the BAM "
                                  "sits in the page cache it was written through.  e2e_cold_cache is the same leg with the file's pages dropped first; a first "
                                  "command-line run of a fresh process additionally pays the first hipMalloc / hipHostMalloc calls (0.15-0.35 s, measured: "
                                  "tools/e2e_cli_timing.py)") if headline_e2e else None,
                   "step": ("one chromosome" if args.workload == "contig" else "one 10 Mb collection window") +
                           " through [file -> inflate -> records ->] scan -> collection -> encode + CNN (batches of %d candidate images, fp32) -> vote" % B,
                   "batch": B, "alignments_rank0": len(table), "cigar_ops_rank0": int(table.cigar.size),
                   "windows_rank0": len(windows), "windows": job_windows, "sites_per_step": job_sites / max(job_windows, 1),
                   "images_per_site": job_images / max(job_sites, 1), "images_per_s": job_images / job_s,
                   "resident_sites_per_s": res_sites / dt, "resident_seconds": dt, "resident_steps": steps_job,
                   "resident_ms_per_step": dt / max(steps_job, 1) * 1e3,
                   "file_inclusive_over_resident": (job_sites / job_s) / max(res_sites / dt, 1e-9) if headline_e2e else None,
                   "rccl_world": world if grouped and tdist.get_backend() == "nccl" else 0, "dist_backend": (tdist.get_backend() if grouped else None),
                   "first_contact": contact,
                   "rank_mb": [round(v, 1) for v in getattr(args, "rank_mb", [])] or None,
                   "imbalance": (max(args.rank_mb) / (sum(args.rank_mb) / len(args.rank_mb)) if getattr(args, "rank_mb", None) else None),
                   "host_cpus_pinned": args.host_cpus or None, "host_workers_per_rank": workers, "host_modules_compiled": not build_host.compiled_state()[1], "host_cores": cores, "host_cpus_visible": visible_cpus, "streams": args.streams, "batches_per_launch": args.launch_batches,
                   "parallelism": "one process per GPU, chromosomes per rank, no data-path collective "
                                  "(score-range all_reduce + record gather once)"},
        "roofline": {"kernel": "device stage per batch of %d images (a graph replay carries --launch-batches of them): encode_conv1_kernel (rasterise + sparse conv1) + "
                               "active_counts / active_lists + conv_wave_list_kernel x4 (fp32 MFMA, conv2-5 on the active pixels) + "
                               "bias_relu_pool_lrn x2 + fc_splitk / fc_reduce x2 (fc6, fc7) + fc8_softmax_kernel; HIP events on the launch's stream "
                               "around every launch of the timed region of `value`, device time = union of the intervals, %d streams" % (B, args.streams),
                     "bound": "mfma", "achieved": rep["executed_tflops"], "peak": F32_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                     "frac": rep["executed_tflops"] * 1e12 / F32_MFMA_PEAK,
                     # the same three ways:
                     "frac_executed": rep["executed_tflops"] * 1e12 / F32_MFMA_PEAK,
                     "frac_algorithmic": rep["algorithmic_tflops"] * 1e12 / F32_MFMA_PEAK,
                     "dense_stage_frac": dense.get("frac"),
                     # the same launches with the device to themselves (the resident leg of this run): the kernels' own figure
                     "frac_stage_alone": rep_res["executed_tflops"] * 1e12 / F32_MFMA_PEAK,
                     "note": ("In the file-inclusive leg -- the timed region of `value` -- the CNN's launches share the chip with the inflate kernels "
                              "(a tokens launch owns every CU while it runs, an LZ launch slows the convolutions next to it) and their "
                              "intervals include that; frac_stage_alone is the same stage with the device to itself, timed in this run "
                              "(resident_leg).  " if headline_e2e else "") +
                             "frac = frac_executed: FLOP the matrix pipe EXECUTED (2 x MAC of the conv2..conv5 outputs actually computed, "
                             "counted on the device over every timed batch, + fc6..fc8) / device time / peak -- the figure that measures kernel "
                             "quality.  frac_algorithmic is SURVEY 8(d)'s literal formula, 1,440,662,592 FLOP x images / device time / peak: it "
                             "exceeds frac_executed (and may exceed 1) by exactly the two structural savings -- the sparse first layer and the "
                             "active-set convolutions, constant folding on the weights, bit-identical to the dense result "
                             "(test_active_path_is_bit_identical_to_the_dense_path) -- not by matrix-pipe utilisation.  dense_stage_frac: the same "
                             "stage with conv2..conv5 computed at every pixel (AlexNet(active=False)), timed in this run (roofline_kernels).",
                     "executed_flop_per_image": rep["executed_flop"], "active_fraction": {k: round(v, 4) for k, v in rep["frac"].items()},
                     "algorithmic_tflops": rep["algorithmic_tflops"],
                     "algorithmic_speedup": CNN_FLOP / rep["executed_flop"],
                     "traffic": TRAFFIC.get("bytes"),
                     "traffic_note": ("HBM / fabric bytes per launch of %d images (the unit `achieved` is quoted per launch of as well: %.1f GFLOP executed), "
                                      "%.3g read + %.3g written, against %.3g algorithmic (%.2fx); PMC counters cannot be read inside this process: "
                                      % (TRAFFIC["images_per_launch"], rep["executed_flop"] * TRAFFIC["images_per_launch"] / 1e9, TRAFFIC["read_bytes"],
                                         TRAFFIC["written_bytes"], TRAFFIC["algorithmic_bytes"], TRAFFIC["bytes"] / TRAFFIC["algorithmic_bytes"])
                                      + TRAFFIC["source"]) if TRAFFIC else "no PMC summary under profiles/",
                     "ms_per_batch": rep["ms_batch"], "batches": rep["batches"], "device_busy_frac": rep["busy"],
                     "resident_leg": {"frac_executed": rep_res["executed_tflops"] * 1e12 / F32_MFMA_PEAK, "ms_per_batch": rep_res["ms_batch"],
                                      "batches": rep_res["batches"], "device_busy_frac": rep_res["busy"]}},
    }
    line["config"]["timed_region_kind"] = (("file-inclusive: BAM on local disk (%.2f GB, page cache warm) -> read, device BGZF inflate + record walk, scan, collection, "
                                            "encode + CNN, vote, cross-rank exchange" % (e2e_block["bam_bytes"] / 1e9)) if headline_e2e else
                                           "resident: alignments decoded and in HBM before the timed region")
    line["roofline"]["kernel_short"] = ("device stage per batch of %d images: encode_conv1 + conv_wave_list x4 (fp32 MFMA 32x32x2, active pixels) + pool/LRN x2 + "
                                        "fc_splitk x2 + fc8_softmax; HIP events around every launch of the timed region" % B)
    legs = e2e_legs if headline_e2e else res_legs
    line["value_repeats"] = repeats_summary(legs, lambda l: l["sites"] / l["seconds"])
    line["resident_repeats"] = repeats_summary(res_legs, lambda l: l["sites"] / l["seconds"])
    if e2e_block is not None:
        for l in e2e_legs:
            l.pop("stage", None)
        e2e_block["resident_sites_per_s"] = res_sites / dt
        e2e_block["ratio_to_resident"] = e2e_block["value"] / max(res_sites / dt, 1e-9)
        line["e2e"] = e2e_block
        line["e2e_repeats"] = [l for l in e2e_legs if l is not e2e_block]
    if e2e_cold is not None:
        e2e_cold["ratio_to_resident"] = e2e_cold["value"] / max(res_sites / dt, 1e-9)
        e2e_cold["note"] = ("the file-inclusive leg once more with the BAM (and its index) dropped from the page cache first (fsync + posix_fadvise DONTNEED; "
                            "page_cache.after_drop = the fraction of its pages still cached when the leg started, .filesystem = where it lies): the reads come "
                            "from the box's disk.  Allocator, staging ring and code objects are as warm as in `value`.")
        line["e2e_cold_cache"] = e2e_cold
    if e2e_other is not None:
        e2e_other["ratio_to_resident"] = e2e_other["value"] / max(res_sites / dt, 1e-9)
        line["e2e_host_ingest" if e2e_other["ingest_engine"] == "cpu" else "e2e_device_ingest"] = e2e_other
    if sweep:
        line["e2e_sweep"] = sweep
    if calib:
        line["roofline_kernels"] = calib
        k3 = [calib.get("conv_wave_list_kernel %s" % n) for n in ("conv3", "conv4", "conv5")]
        if all(k3):
            # the launch that dominates the stage (rocprofv3's conv_wave_list_kernel<3>: conv3, conv4, conv5 -- 59 % of the stage's kernel
            # time), timed live with HIP events on this run's own records: what profiles/rNN_b_device_stage_kernel_stats.csv must agree with
            t_us = sum(k["us"] for k in k3)
            executed = sum(k["achieved"] * k["us"] for k in k3) / t_us                      # TFLOP/s over the three launches
            dense = sum(LAYER_FLOP[n] for n in ("conv3", "conv4", "conv5")) * B * max(1, args.launch_batches) / (t_us * 1e-6) / 1e12
            line["roofline"]["dominant_kernel"] = {"name": "conv_wave_list_kernel<3>", "launches": "conv3, conv4, conv5 of %d images" % (B * max(1, args.launch_batches)),
                                                   "avg_us": t_us / 3, "achieved": executed, "frac": executed * 1e12 / F32_MFMA_PEAK,
                                                   "frac_algorithmic": dense * 1e12 / F32_MFMA_PEAK}
        scan_k, ras_k = calib.get("cigar_scan (4 kernels)"), next((v for k, v in calib.items() if k.startswith("raster_kernel")), None)
        if scan_k and ras_k:                  # the two HBM-bound kernels of the path, beside the MFMA-bound stage
            line["roofline_hbm"] = {"peak": HBM_PEAK / 1e9, "unit": "GB/s", "cigar_scan_achieved": scan_k["achieved"], "cigar_scan_frac": scan_k["frac"],
                                    "raster_achieved": ras_k["achieved"], "raster_frac": ras_k["frac"]}
    # ---- parity leg (untimed): the windows the CPU baseline is about to run, through the path `value` was measured on, with the
    # helpers returning their TSV text and the owner keeping the predictions
    gpu_side = None
    parity_windows = cpu_pool.windows_of_run(e2e["windows"] if headline_e2e else windows) if cpu_pool is not None and not args.no_parity else []
    if parity_windows:
        gpu_side = {}
        hot.pool.set_option("want_tsv", True)
        hot.keep_predictions = True
        if headline_e2e:
            sub = dict(e2e, header_references=e2e["references"], windows=parity_windows)
            sub["references"] = [n for n in e2e["references"] if any(w[0] == n for w in parity_windows)]
            run_from_file(args, sub, hot, run, fasta, opts, dev, sync_all, cores, world, workers, grouped,
                          engine=os.environ.get("SVX_INGEST", "auto"), keep=True, sink=gpu_side)
        else:
            run(parity_windows, sink=gpu_side)
        hot.keep_predictions = False
    hot.close()
    if e2e is not None:
        import shutil
        shutil.rmtree(e2e["dir"], ignore_errors=True)
    parity_ok = True
    if cpu_pool is not None:
        line["cpu_baseline"], cpu_side = cpu_pool.run(e2e["windows"] if headline_e2e and parity_windows else windows)
        if gpu_side is not None:
            line["parity_check"] = parity_check(gpu_side, cpu_side, "file-inclusive" if headline_e2e else "resident")
            parity_ok = line["parity_check"]["ok"]
    if rank == 0:
        detail = None
        if args.detail:
            try:
                with open(args.detail, "w") as f:
                    json.dump(line, f, indent=1, default=str)
                detail = args.detail
            except OSError as exc:
                print("bench: could not write %s: %s" % (args.detail, exc), file=sys.stderr)
        print(json.dumps({k: line[k] for k in ("value_repeats", "resident_repeats", "parity_check") if k in line}), file=sys.stderr)
        sys.stderr.flush()
        print(compact_line(line, detail))
        sys.stdout.flush()
    if grouped:
        tdist.destroy_process_group()
    if not parity_ok:
        print("bench: PARITY MISMATCH between the device path and the CPU port: %s" % json.dumps(line["parity_check"], default=str)[:2000], file=sys.stderr)
        sys.exit(3)


SOFTMAX_TOL = 1e-3                            # BASELINE.json north_star: CNN softmax within 1e-3 fp32


def parity_check(gpu_side, cpu_side, path):
    """The device path against the CPU port on the windows both ran.  gpu_side: {window: (TSV text, classes, probs)} from the
    pipeline `value` was measured on; cpu_side: per CPU-baseline process {window, tsv_sha, sites_sha, n_lines, index (the TSV
    lines it classified), classes, probs}.  Bit-exact: the window's segment TSV (every pair line: region key, both segments,
    lengths -- output_clusters.py:84-89) and the ordered list of distinct site keys; within SOFTMAX_TOL: the softmax of every
    image the port classified (predict.py:209)."""
    import hashlib
    per, seen = [], {}
    tsv_equal = sites_equal = True
    worst, images, class_diff = 0.0, 0, 0
    for c in cpu_side:
        w = tuple(c["window"])
        if w not in gpu_side:
            per.append({"window": list(w), "error": "not run by the device leg"})
            tsv_equal = sites_equal = False
            continue
        tsv, classes, probs = gpu_side[w]
        if w not in seen:
            regions = []
            for ln in (tsv or "").splitlines():
                r = ln.split("\t", 1)[0]
                if not regions or regions[-1] != r:
                    regions.append(r)
            seen[w] = (hashlib.sha256((tsv or "").encode()).hexdigest(), hashlib.sha256("\n".join(regions).encode()).hexdigest(),
                       (tsv or "").count("\n"), len(set(regions)))
        g_tsv, g_sites, g_lines, g_nsites = seen[w]
        t_ok, s_ok = g_tsv == c["tsv_sha"] and g_lines == c["n_lines"], g_sites == c["sites_sha"]
        tsv_equal, sites_equal = tsv_equal and t_ok, sites_equal and s_ok
        delta = 0.0
        if len(c["index"]) and t_ok:
            idx = np.asarray(c["index"], np.int64)
            delta = float(np.abs(np.asarray(probs)[idx].astype(np.float64) - c["probs"].astype(np.float64)).max())
            class_diff += int((np.asarray(classes)[idx] != c["classes"]).sum())
            images += len(idx)
        worst = max(worst, delta)
        per.append({"window": list(w), "lines": g_lines, "sites": g_nsites, "tsv_equal": t_ok, "sites_equal": s_ok,
                    "images_compared": len(c["index"]) if t_ok else 0, "max_softmax_delta": delta})
    ok = bool(tsv_equal and sites_equal and worst <= SOFTMAX_TOL and per)
    return {"ok": ok, "path": path, "windows": len(seen), "tsv_equal": bool(tsv_equal), "sites_equal": bool(sites_equal),
            "tsv_lines": int(sum(v[2] for v in seen.values())), "sites": int(sum(v[3] for v in seen.values())),
            "images_compared": images, "max_softmax_delta": worst, "softmax_tol": SOFTMAX_TOL, "argmax_differs": class_diff,
            "per_window": per}


def cached_fraction(path):
    """Fraction of a file's pages that sit in the page cache (mincore over a mapping of it); None where that cannot be asked."""
    import ctypes
    import mmap
    try:
        size = os.path.getsize(path)
        if size == 0:
            return 0.0
        with open(path, "rb") as f:
            m = mmap.mmap(f.fileno(), size, prot=mmap.PROT_READ)
        try:
            page = mmap.PAGESIZE
            n = (size + page - 1) // page
            vec = (ctypes.c_ubyte * n)()
            libc = ctypes.CDLL(None, use_errno=True)
            # mmap objects opened read-only do not export a writable buffer: take the address through a NumPy view
            arr = np.frombuffer(m, dtype=np.uint8)
            rc = libc.mincore(ctypes.c_void_p(arr.ctypes.data), ctypes.c_size_t(size), vec)
            del arr
            if rc != 0:
                return None
            return float(sum(v & 1 for v in vec)) / n
        finally:
            m.close()
    except Exception:                                          # noqa: BLE001 -- a report, never a failure
        return None


def drop_file_cache(paths):
    """Write the files back and ask the kernel to drop their pages (posix_fadvise DONTNEED) -> {path: cached fraction afterwards}:
    what a first run on a file that nobody has read yet finds.  (A file on tmpfs stays where it is: its pages ARE the file.)"""
    out = {}
    for path in paths:
        try:
            fd = os.open(path, os.O_RDONLY)
            try:
                os.fsync(fd)
                os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
            finally:
                os.close(fd)
        except OSError:
            pass
        out[os.path.basename(path)] = cached_fraction(path)
    return out


def filesystem_of(path):
    best = ("", "?")
    try:
        real = os.path.realpath(path)
        with open("/proc/mounts") as f:
            for line in f:
                _dev, mnt, fstype = line.split()[:3]
                if (real == mnt or real.startswith(mnt.rstrip("/") + "/")) and len(mnt) > len(best[0]):
                    best = (mnt, fstype)
    except OSError:
        pass
    return best[1]


def run_from_file(args, e2e, hot, run, fasta, opts, dev, sync_all, cores, world, workers, grouped, engine="cpu", keep=False, cold=False, sink=None):
    """The file-inclusive leg (wall of Step 1 + Step 2 with the BAM on local disk): a timed region of its
    own that starts with nothing but the file -- svx_bam_stream_* reads and inflates it on host threads chromosome by
    chromosome, every chromosome is uploaded and scanned on the device when it arrives and handed to the helpers through
    shared memory, the windows flow through the same pipeline -- and ends after the cross-rank exchange."""
    import shutil
    import torch.distributed as tdist
    from svision_amd.ingest import ChromosomeFeed, StaticFeed
    refs = e2e["references"]                                   # this leg's chromosomes; the file's dictionary may hold more (warm-up pass)
    header_refs = e2e.get("header_references", refs)
    lens = [fasta.get_reference_length(n) for n in header_refs]
    from svision_amd.ingest import decode_threads
    threads = args.decode_threads or decode_threads(world, workers)
    resident = hot.feed
    cache = None
    if cold:                                                   # the leg beside `value` that reads the file from the disk, not from the page cache
        cache = {"after_drop": drop_file_cache([e2e["path"], e2e["path"] + ".bai"]), "filesystem": filesystem_of(e2e["path"])}
    sync_all()
    t0 = time.perf_counter()
    tasks = {}
    for c, a, b in e2e["windows"]:
        tasks.setdefault(c, []).append((a, b))
    feed = ChromosomeFeed(e2e["path"], fasta, opts, refs, header_refs, lens, device=dev, index=e2e["path"] + ".bai", threads=threads, engine=engine, tasks=tasks)
    hot.feed = feed
    try:
        sites, images, records, scores = run(e2e["windows"], rescan=False, sink=sink)
        sdist.exchange_score_range(scores)
        sdist.gather_texts({"rank%d" % sdist.world()[0]: "%d records" % records})
        sync_all()
        dt = time.perf_counter() - t0
    finally:
        for chrom in refs:
            hot.release(chrom)
        feed.close()
        hot.feed = resident
        if not keep:
            shutil.rmtree(e2e["dir"], ignore_errors=True)
    dev_ms = hot.device_busy_ms()
    cnn_gaps, cnn_span = [], None
    if getattr(hot, "batch_events", None):                      # where the CNN stood still: gaps of more than 3 ms between two launches
        ref = hot._t_ref
        iv = sorted((ref.elapsed_time(ev[0]), ref.elapsed_time(ev[1])) for ev in hot.batch_events)
        hi = iv[0][1]
        for lo_, hi_ in iv[1:]:
            if lo_ - hi > 3.0:
                cnn_gaps.append([round(hi, 1), round(lo_ - hi, 1)])
            hi = max(hi, hi_)
        cnn_span = round(hi - iv[0][0], 1)
    tot = torch.tensor([sites, images, len(e2e["windows"]), e2e["bytes"], e2e["inflated"], dev_ms], dtype=torch.float64, device=dev)
    if grouped:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        tdist.all_reduce(tot, op=tdist.ReduceOp.SUM)
        tdist.all_reduce(tmax, op=tdist.ReduceOp.MAX)
        dt = float(tmax.item())
    tot = tot.cpu().numpy()
    st = feed.stats
    if cache is not None:
        cache["after_leg"] = {os.path.basename(e2e["path"]): cached_fraction(e2e["path"])}
    return {"page_cache": cache if cache is not None else "warm (the file was written during set-up and read by the warm-up pass)",
            "ingest_engine": st.get("engine"), "value": float(tot[0]) / dt, "unit": "sites/s", "seconds": dt, "windows": int(tot[2]), "sites": int(tot[0]), "images": int(tot[1]),
            "bam_bytes": int(tot[3]), "inflated_bytes": int(tot[4]), "compressed_GB_per_s": float(tot[3]) / dt / 1e9,
            "inflated_GB_per_s": float(tot[4]) / dt / 1e9, "inflate_threads_per_rank": threads, "device_busy_frac": float(tot[5]) * 1e-3 / world / dt,
            "cnn_span_ms_rank0": cnn_span, "cnn_gaps_ms_rank0": cnn_gaps,        # from the first launch's upload: [start of the gap, length] of every pause > 3 ms
            "rank0_feed": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()},
            "owner_profile_rank0": {k: round(v, 4) for k, v in getattr(hot, "owner_profile", {}).items()},
            "note": "timed from opening the BAM (random bases, 7-bin qualities: ~0.4 compressed bytes per base, like a HiFi BAM) to the end of "
                    "the cross-rank exchange; the BAM sits in the page cache of the box (written during set-up), not on a cold disk"}


def kernel_calibration(hot, sample, net, dev, B, window, reps=20):
    """Live per-kernel timings (HIP events on the launch stream, eager, outside the timed region) on the workload's own
    records: the first full launch (B = batch x batches per launch images) of one window, each kernel fed with the tensors
    the stage really hands it."""
    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    out = {}
    res = hot.collect(*window, rescan=False)
    if res.records.shape[0] < B:
        return out
    rec = torch.from_numpy(res.records[:B]).to(dev)
    saved, net.executed = net.executed, None
    bg = net.background()
    x1, touched = kernels.encode_conv1(rec, net.conv1_hwio, net.conv1_base, touched=True)
    out["encode_conv1_kernel"] = {"bound": "latency", "launch": "%d images" % B,
                                  "us": timed(lambda: kernels.encode_conv1(rec, net.conv1_hwio, net.conv1_base, touched=True)) * 1e6,
                                  "note": "rasterise + sparse conv1 + relu + pool + LRN; replaces %.1f MB of image traffic and %.1f GFLOP "
                                          "of dense conv1 per launch" % (IMG_BYTES * B / 1e6, 210_830_400 * B / 1e9)}
    l2, l3, l4, l5, counts, rows2 = kernels.alexnet_active_sets(touched, rows=True)
    out["active_counts + active_lists"] = {"bound": "latency", "launch": "%d images" % B, "us": timed(lambda: kernels.alexnet_active_sets(touched)) * 1e6}
    cnt = counts.cpu().numpy().astype(np.float64)
    x = x1
    for li, (name, lst, bias, relu, groups) in enumerate((("conv2", l2, None, False, 2), ("conv3", l3, net.conv3_b, True, 1),
                                                          ("conv4", l4, net.conv4_b, True, 2), ("conv5", l5, None, False, 2))):
        w = getattr(net, name + "_w")
        if name == "conv2":                  # as in the stage: active pixels only, the pool reads the background for the others
            out2 = torch.empty((B, 32, 27, 27, 8), dtype=torch.float32, device=dev)
            fn = lambda x=x, w=w, lst=lst: kernels.conv2d_same(x, w, None, groups=2, pixels=lst, pixel_count=counts[0:1], out=out2)   # noqa: E731
        else:
            fn = lambda x=x, w=w, lst=lst, li=li, bias=bias, relu=relu, groups=groups, name=name: kernels.conv2d_same(       # noqa: E731
                x, w, bias, groups=groups, relu=relu, pixels=lst, pixel_count=counts[li:li + 1], background=bg[name])
        t = timed(fn)
        npix = B * LAYER_PIX[name]
        computed = npix if cnt[li] * 100 >= npix * 97 else cnt[li]
        flop = LAYER_FLOP[name] * computed / LAYER_PIX[name]
        out["conv_wave_list_kernel %s" % name] = {"bound": "mfma", "launch": "%d of %d output pixels active (%d images), %s" % (cnt[li], npix, B, tuple(w.shape)),
                                                   "us": t * 1e6, "achieved": flop / t / 1e12, "peak": F32_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                                                   "frac": flop / t / F32_MFMA_PEAK, "note": "executed FLOP of the launch / its duration"}
        y = fn()
        tdense = timed(lambda x=x, w=w, bias=bias, relu=relu, groups=groups: kernels.conv2d_same(x, w, bias, groups=groups, relu=relu))
        out["conv_wave_kernel %s (dense mode, same input)" % name] = {"bound": "mfma", "us": tdense * 1e6, "achieved": LAYER_FLOP[name] * B / tdense / 1e12,
                                                                      "peak": F32_MFMA_PEAK / 1e12, "unit": "TFLOP/s", "frac": LAYER_FLOP[name] * B / tdense / F32_MFMA_PEAK}
        if name in ("conv2", "conv5"):
            b2 = getattr(net, name + "_b")
            extra = dict(active_rows=rows2, background=bg["conv2"]) if name == "conv2" else {}
            t = timed(lambda y=y, b2=b2, name=name, extra=extra: kernels.bias_relu_pool_lrn(y, b2, lrn=name == "conv2", **extra))
            out["bias_relu_pool_lrn_kernel %s" % name] = {"bound": "hbm", "us": t * 1e6, "achieved": (y.numel() * 4 + y.numel()) / t / 1e9,
                                                          "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": (y.numel() * 5) / t / HBM_PEAK,
                                                          "note": "launch-latency regime (%.1f MB)" % (y.numel() * 5 / 1e6)}
            x = kernels.bias_relu_pool_lrn(y, b2, lrn=name == "conv2", **extra)
        else:
            x = y
    h = x.reshape(B, 9216)
    t6 = timed(lambda: kernels.fc_bias_act(h, net.fc6_w, net.fc6_b, relu=True))
    h6 = kernels.fc_bias_act(h, net.fc6_w, net.fc6_b, relu=True)
    t7 = timed(lambda: kernels.fc_bias_act(h6, net.fc7_w, net.fc7_b, relu=True))
    wbytes = (net.fc6_w.numel() + net.fc7_w.numel()) * 4
    fc_flop = 2.0 * B * (9216 * 4096 + 4096 * 4096)
    out["fc_splitk_kernel + fc_reduce_kernel (fc6, fc7)"] = {
        "bound": "mfma", "us": (t6 + t7) * 1e6, "achieved": fc_flop / (t6 + t7) / 1e12, "peak": F32_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
        "frac": fc_flop / (t6 + t7) / F32_MFMA_PEAK,
        "note": "fp32 MFMA over a 218 MB weight stream: %.0f GB/s of weights (%.2f of the HBM peak) at this launch size (%d images); "
                "at one batch of 64 the two limits are balanced, from two batches on the matrix pipe is the bound"
                % (wbytes / (t6 + t7) / 1e9, wbytes / (t6 + t7) / HBM_PEAK, B)}
    t = timed(lambda: net.predict_records_packed(rec))
    out["device_stage_eager_1_stream"] = {"bound": "mfma", "launch": "%d images of the workload" % B, "us": t * 1e6}
    dense_net = AlexNet(random_weights(0), device=dev, active=False)
    td = timed(lambda: dense_net.predict_records_packed(rec))
    out["device_stage_eager_1_stream_dense_convolutions"] = {
        "bound": "mfma", "launch": "%d images" % B, "us": td * 1e6, "achieved": (CNN_FLOP - 210_830_400) * B / td / 1e12, "peak": F32_MFMA_PEAK / 1e12,
        "unit": "TFLOP/s", "frac": (CNN_FLOP - 210_830_400) * B / td / F32_MFMA_PEAK,
        "note": "the same stage with conv2..conv5 computed at every pixel (AlexNet(active=False)); outputs bit-identical; FLOP of "
                "conv2..fc8 (conv1 is sparse in both)"}
    del dense_net
    net.executed = saved
    # the two memory-bound kernels of the path at streaming sizes
    table = sample.table
    d_cigar, d_off, d_pos, _ = sample.device_buffers
    n = len(table)
    cap = max(1024, int(sample.gap_off[-1]) + 16)
    t = timed(lambda: kernels.cigar_scan(d_cigar, d_off, d_pos, sample.min_sv, gaps_cap=cap))
    alg = 4 * int(table.cigar.size) + 32 * n + 24 * int(sample.gap_off[-1])
    out["cigar_scan (4 kernels)"] = {"bound": "hbm", "launch": "%d alignments, %d ops" % (n, table.cigar.size), "us": t * 1e6,
                                     "achieved": alg / t / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": alg / t / HBM_PEAK}
    nbig = 2048
    big = torch.from_numpy(np.resize(res.records, (nbig, 12)).astype(np.int32)).to(dev)
    big_img = torch.empty((nbig, 3, 227, 227), dtype=torch.float32, device=dev)
    t = timed(lambda: kernels.rasterize(big, layout="NCHW", out=big_img))
    out["raster_kernel %d images" % nbig] = {"bound": "hbm", "launch": "%d images (%.1f GB written)" % (nbig, IMG_BYTES * nbig / 1e9), "us": t * 1e6,
                                             "achieved": IMG_BYTES * nbig / t / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                             "frac": IMG_BYTES * nbig / t / HBM_PEAK,
                                             "note": "dense image path (BatchGenerator API); the pipeline uses encode_conv1 instead"}
    return out


# --------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle port of the same step on this box's host cores, as a pool of P single-thread processes, P = the
# CPUs the process may really use (affinity mask capped by the cgroup's CPU-time quota: svision_amd.ingest.effective_cpus) --
# the reference's `-t P` (SVision:261,311) -- on a bounded sample of the same workload.
def _cpu_worker(conn, table, fasta, opts):
    import io
    from oracle import cbind
    from oracle.alexnet_torch import TorchAlexNet
    from svision_amd.collection.output_clusters import collect_pair_lines
    from svision_amd.collection.run_collection import detect_window
    from svision_amd.network.predict import Predict, SiteVoter
    torch.set_num_threads(1)
    net = None
    sample = None
    while True:
        msg = conn.recv()
        if msg is None:
            return
        chrom, start, end, part, parts, max_images = msg
        t0 = time.perf_counter()
        if sample is None:                                    # C oracle scan of the window's chromosome (once per process)
            sub = table.subset(np.flatnonzero(table.tid == table.get_tid(chrom)))
            scan = cbind.cigar_scan(sub.cigar, sub.cig_off.astype(np.uint64), sub.pos, opts.min_sv_size)
            sample = Sample.with_scan(sub, fasta, opts.min_sv_size, scan)
            net = TorchAlexNet(random_weights(0), device="cpu")
        _sigs, clusters = detect_window(opts, sample, chrom, start, end)
        all_lines = collect_pair_lines(clusters, opts)
        regions = []
        for ln in all_lines:                                  # this process's share of the window's sites
            if not regions or regions[-1] != ln.region:
                regions.append(ln.region)
        mine = set(regions[part::parts])
        index = [i for i, ln in enumerate(all_lines) if ln.region in mine]
        lines = [all_lines[i] for i in index]
        B = 128                                               # reference default batch (SVision:88)
        recs = np.asarray([ln.record() for ln in lines], np.int32).reshape(-1, 12)
        done, outs = 0, []
        while done < len(lines) and done < max_images:
            x = cbind.rasterize(recs[done:done + B], "NCHW")
            _l, cls, prob = net.predict(torch.from_numpy(x))
            outs.append((cls.numpy(), prob.numpy()))
            done = min(len(lines), done + B)
        sites = 0
        if done:
            voter = SiteVoter(Predict(chrom, None), io.StringIO(), io.StringIO(), opts, sample)
            voter.feed_batch([ln.label() for ln in lines[:done]], np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs]))
            voter.finish()
            sites = len({ln.region for ln in lines[:done]})
        seconds = time.perf_counter() - t0
        # what bench.py's parity_check compares with the device path (outside this process's own seconds): digests of the window's
        # whole TSV and of its ordered site keys, and the softmax of the images classified here
        import hashlib
        text = "".join(ln.text() for ln in all_lines)
        side = {"window": (chrom, start, end), "tsv_sha": hashlib.sha256(text.encode()).hexdigest(),
                "sites_sha": hashlib.sha256("\n".join(regions).encode()).hexdigest(), "n_lines": len(all_lines),
                "index": np.asarray(index[:done], np.int64),
                "classes": np.concatenate([o[0] for o in outs])[:done] if outs else np.empty(0, np.int64),
                "probs": np.concatenate([o[1] for o in outs])[:done] if outs else np.empty((0, 5), np.float32)}
        conn.send((sites, done, seconds, side))


class CpuBaselinePool:
    """P forked single-thread processes (forked before the first HIP call; idle until run())."""

    def __init__(self, procs, table, fasta, opts, visible=None):
        import multiprocessing as mp
        self.visible = visible or procs
        ctx = mp.get_context("fork")
        self.conns, self.procs = [], []
        for _ in range(procs):
            a, b = ctx.Pipe(duplex=True)
            p = ctx.Process(target=_cpu_worker, args=(b, table, fasta, opts), daemon=True)
            p.start()
            b.close()
            self.conns.append(a)
            self.procs.append(p)

    def windows_of_run(self, windows):
        """The distinct windows run(windows) hands to the processes, in order."""
        P = len(self.conns)
        per_win = max(1, P // max(len(windows), 1))
        out = []
        for i in range(P):
            w = tuple(windows[(i // per_win) % len(windows)])
            if w not in out:
                out.append(w)
        return out

    def run(self, windows, images_per_proc=None):
        """Every process takes 1/P of the sites of one window (round-robin over the windows) and at most `images_per_proc`
        of their images: about 10-30 s of wall time; value = sites classified by the pool / wall time.
        -> (the cpu_baseline object, per process what parity_check compares)."""
        P = len(self.conns)
        if images_per_proc is None:                           # ~20 s of wall: a 1-thread process classifies 15-80 images per second
            images_per_proc = 192 if P >= 64 else 1536
        per_win = max(1, P // max(len(windows), 1))
        t0 = time.perf_counter()
        for i, c in enumerate(self.conns):
            chrom, start, end = windows[(i // per_win) % len(windows)]
            c.send((chrom, start, end, i % per_win, per_win, images_per_proc))
        got = [c.recv() for c in self.conns]
        wall = time.perf_counter() - t0
        for c in self.conns:
            c.send(None)
        for p in self.procs:
            p.join(timeout=5)
        sites, images = sum(g[0] for g in got), sum(g[1] for g in got)
        short = ("%d single-thread processes (the reference's -t P): oracle C scan + host collection of one 10 Mb window each, oracle C rasteriser + "
                 "PyTorch-CPU fp32 AlexNet (batch 128) + vote on <= %d images: %d sites, %d images in %.1f s wall" % (P, images_per_proc, sites, images, wall))
        return {"value": sites / wall, "unit": "sites/s", "cores": P, "kind": "port", "cpus_visible": self.visible, "sample_short": short,
                "sample": "pool of %d single-thread processes (the reference's -t P, SVision:261,311), each: C oracle scan of its window's "
                          "chromosome, host collection of one 10 Mb window, then C oracle rasteriser + PyTorch-CPU fp32 AlexNet (batch 128, "
                          "1 thread) + vote on its 1/%d share of that window's sites, capped at %d images: %d sites, %d images in %.1f s "
                          "wall (slowest process %.1f s)" % (P, per_win, images_per_proc, sites, images, wall, max(g[2] for g in got))}, [g[3] for g in got]


if __name__ == "__main__":
    main()
