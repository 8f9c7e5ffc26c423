#!/usr/bin/env python3
"""bench.py -- candidate SV sites/sec (encode+CNN) on N MI355X of one node.

Workload (BASELINE.json configs[1] stand-in, synthetic: the reference ships no BAM):
a chr21-sized contig (46,709,983 bp), HiFi reads N(15 kb, 2 kb) at 30x with planted
SVs; its alignments are decoded to packed arrays and resident in HBM before the timed
region.  A *step* is one collection window of the reference driver (10 Mb, SVision:88)
through the whole hot path:

  device  CIGAR/segment scan of the window's alignments        (svx_cigar_scan)
  host    reads -> signatures -> clusters -> segment pairs      (parity-tested mirror of src/collection)
  device  similarity-image rasterisation + AlexNet fp32, batches of 64 candidate images
  host    per-site vote -> VCF body lines + scores

value = candidate sites (distinct regions of the segment TSV) per second, whole job.
With N>1 every rank owns its own chromosome-sized shard (weak scaling, no data-path
collective; one score-range all_reduce + one record gather at the end, inside the timed
region).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from svision_amd import dist as sdist, kernels, synth  # noqa: E402
from svision_amd.io import bam  # noqa: E402
from svision_amd.network.alexnet import AlexNet, checkpoint_shapes  # noqa: E402
from svision_amd.pipeline import HelperPool, PooledHotPath  # noqa: E402
from svision_amd.sample import Sample  # noqa: E402

IMG_BYTES = 227 * 227 * 3 * 4 + 48            # 618,348 B written + 48 B read per image
CNN_FLOP = 1_440_662_592                      # FLOP per image
HBM_PEAK = 8.0e12                             # MI355X_MICROARCH.md: HBM3E 8 TB/s
F32_MFMA_PEAK = 157.3e12                      # MI355X_MICROARCH.md: FP32 matrix peak (f32-in MFMA)
# HBM/fabric bytes per 64-image batch of the device stage, from PMC counters (rocprofv3 --pmc FETCH_SIZE and
# --pmc WRITE_SIZE in separate passes, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md), summed over its
# kernels: profiles/r01_pmc_traffic.md.  Offline measurement (counters cannot be read inside this process).
PMC_TRAFFIC_PER_BATCH64 = 7.04e8
CHR21 = 46_709_983


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
        max_sv_size=1000000, window_size=10000000, patition_max_distance=5000, cluster_max_distance=0.3,
        batch_size=batch, min_gt_depth=4, homo_thresh=0.8, hete_thresh=0.2, k_size=10, min_accept=50, max_hash_len=1000)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="candidate images per CNN batch (BASELINE configs[1])")
    ap.add_argument("--contig-len", type=int, default=CHR21)
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--workers", type=int, default=16, help="helper processes for the Python host glue (collection, vote)")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams the per-batch graphs are replayed on")
    ap.add_argument("--inflight", type=int, default=6, help="windows enqueued on the device at once")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # order matters: the host helpers are forked before the first HIP call (forking with a live GPU context makes the
    # driver evict / restore the queues: seconds of stall), then the device scan, then RCCL
    rank = int(os.environ.get("RANK", "0"))

    # ---- untimed set-up: synthetic sample -> packed arrays -> HBM ----
    cfg = synth.SimConfig(contigs=[("chr21", args.contig_len)], coverage=args.coverage, seed=1 + rank)
    table, genome, _svs = synth.simulate(cfg)
    opts = options_ns(args.batch)
    fasta = bam.Fasta(sequences=genome)
    pool = HelperPool(args.workers, opts, table=table, fasta=fasta)
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sample = Sample.from_table(table, fasta, opts.min_sv_size, device=dev)
    pool.attach_scan(sample)
    net = AlexNet(random_weights(0), device=dev)
    hot = PooledHotPath(sample, opts, net, device=dev, n_streams=args.streams, max_inflight=args.inflight, pool=pool)
    rank, world = sdist.init_from_env()
    windows = []
    pos = 0
    while pos < args.contig_len:
        windows.append(("chr21", pos, min(args.contig_len, pos + opts.window_size)))
        pos += opts.window_size

    import torch.distributed as tdist
    grouped = tdist.is_available() and tdist.is_initialized()

    def sync_all():
        if grouped:
            tdist.barrier()
        torch.cuda.synchronize()

    def run(n_steps):
        seq = [windows[i % len(windows)] for i in range(n_steps)]
        sites = images = records = 0
        scores = []
        hot.device_events.clear()
        for res in hot.run_windows(seq):
            sites += res.n_sites; images += res.n_images; records += res.n_records
            scores += [float(s) for s in res.scores.split()]
        return sites, images, records, scores

    run(args.warmup)
    sync_all()
    t0 = time.perf_counter()
    sites, images, records, scores = run(args.steps)
    # the single cross-shard exchange of the job: score range + record gather (dist.py)
    sdist.exchange_score_range(scores)
    sdist.gather_texts({"rank%d" % rank: "%d records" % records})
    sync_all()
    dt = time.perf_counter() - t0

    totals = torch.tensor([sites, images, dt], dtype=torch.float64, device=dev)
    if grouped:
        tmax = totals.clone()
        tdist.all_reduce(totals, op=tdist.ReduceOp.SUM)
        tdist.all_reduce(tmax, op=tdist.ReduceOp.MAX)
        dt = float(tmax[2].item())
    tot_sites, tot_images = float(totals[0].item()), float(totals[1].item())

    B = args.batch
    active = active_fractions(hot, net, windows[0], dev)
    dev_ms = sum(e0.elapsed_time(e1) for e0, e1, _n in hot.device_events)
    dev_images = sum(n for _e0, _e1, n in hot.device_events)
    hot.close()
    cnn_tflops = CNN_FLOP * dev_images / (dev_ms * 1e-3) / 1e12
    ms_batch = dev_ms / max(dev_images / B, 1)
    line = {
        "metric": "candidate SV sites/sec (encode+CNN)",
        "value": tot_sites / dt,
        "unit": "sites/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "cfg2 stand-in: synthetic HiFi chr21 (%d bp, N(15kb,2kb) reads, %gx), step = one 10 Mb window, "
                               "CNN batch = %d candidate images, fp32" % (args.contig_len, args.coverage, B),
                   "batch": B, "alignments": len(table), "cigar_ops": int(table.cigar.size), "windows": len(windows),
                   "sites_per_step": sites / args.steps, "images_per_site": images / max(sites, 1),
                   "images_per_s": tot_images / dt, "host_workers": args.workers, "streams": args.streams, "parallelism": "one process per GPU, chromosome-sized shard per rank, "
                   "no data-path collective (score-range all_reduce + record gather once)"},
        "roofline": {"kernel": "device stage per batch of %d images: encode_conv1_kernel (rasterise + sparse conv1) + active-set lists + "
                               "conv_igemm_kernel x4 (fp32 MFMA, conv2-5 on the active pixels) + pool/LRN epilogues + fc6/fc7 (hipBLASLt) + "
                               "fc8_softmax_kernel, graph replays on %d streams" % (B, args.streams), "bound": "mfma",
                     "achieved": cnn_tflops, "peak": F32_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                     "frac": cnn_tflops * 1e12 / F32_MFMA_PEAK,
                     "note": "achieved = ALGORITHMIC FLOP of the dense network (SURVEY 8(d): 1.44 GFLOP per image) / device time.  "
                             "Two exact structural savings execute fewer of them, so frac can exceed 1: the sparse first layer "
                             "(conv1's 211 MFLOP per image are never issued) and the active-set convolutions (conv2..conv5 compute "
                             "only the outputs with a line of the image in their receptive field; the rest is the precomputed "
                             "response to an empty image, bit-identical to the dense result).  executed_* = what the matrix "
                             "pipe actually ran, from the active fractions of one window.",
                     "active_fraction": active["fractions"], "executed_flop_per_image": active["executed_flop"],
                     "executed_tflops": active["executed_flop"] * dev_images / (dev_ms * 1e-3) / 1e12,
                     "executed_frac": active["executed_flop"] * dev_images / (dev_ms * 1e-3) / F32_MFMA_PEAK,
                     "traffic": PMC_TRAFFIC_PER_BATCH64 * B / 64 if B == 64 else None,
                     "traffic_unit": "bytes per batch (rocprofv3 PMC, profiles/r01_pmc_traffic.md; algorithmic minimum ~4.8e8: "
                                     "227.6 MB weights + activations)", "ms_per_batch": ms_batch,
                     "device_busy_frac": dev_ms * 1e-3 / dt},
        "roofline_kernels": kernel_calibration(sample, net, dev, B),
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_windows(sample, opts, windows, net)
    if rank == 0:
        print(json.dumps(line))
    if grouped:
        tdist.destroy_process_group()


# dense FLOP per image of the layers behind the first one (2 x MAC)
LAYER_FLOP = {"conv2": 447_897_600, "conv3": 299_040_768, "conv4": 224_280_576, "conv5": 149_520_384, "fc": 109_092_864}


def active_fractions(hot, net, window, dev):
    """Share of conv2..conv5 outputs the active-set path computes, measured on one window's records (untimed)."""
    res = hot.collect(*window, rescan=False)
    rec = torch.from_numpy(res.records).to(dev)
    if not getattr(net, "active", False) or rec.shape[0] == 0:
        return {"fractions": None, "executed_flop": float(sum(LAYER_FLOP.values()))}
    _x, touched = kernels.encode_conv1(rec, net.conv1_hwio, net.conv1_base, touched=True)
    counts = kernels.alexnet_active_sets(touched)[4].cpu().numpy().astype(np.float64)
    n = rec.shape[0]
    frac = {"conv2": counts[0] / (n * 729), "conv3": counts[1] / (n * 169), "conv4": counts[2] / (n * 169), "conv5": counts[3] / (n * 169)}
    executed = sum(LAYER_FLOP[k] * frac[k] for k in frac) + LAYER_FLOP["fc"]
    return {"fractions": {k: round(float(v), 4) for k, v in frac.items()}, "executed_flop": float(executed)}


def kernel_calibration(sample, net, dev, B, reps=20):
    """Live per-kernel timings of the hand-written kernels (HIP events on the launch stream, eager,
    outside the timed region) at the sizes the pipeline uses them."""
    from tests import datagen

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
    rec = torch.from_numpy(datagen.random_records(B, seed=7, hostile=False)).to(dev)
    img = torch.empty((B, 3, 227, 227), dtype=torch.float32, device=dev)
    t = timed(lambda: kernels.rasterize(rec, layout="NCHW", out=img))
    out["raster_kernel"] = {"bound": "hbm", "launch": "%d images" % B, "us": t * 1e6, "achieved": IMG_BYTES * B / t / 1e9,
                            "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": IMG_BYTES * B / t / HBM_PEAK,
                            "note": "dense image path (BatchGenerator API); the pipeline uses encode_conv1 instead"}
    big = torch.from_numpy(datagen.random_records(4096, seed=8, hostile=False)).to(dev)
    big_img = torch.empty((4096, 3, 227, 227), dtype=torch.float32, device=dev)
    t = timed(lambda: kernels.rasterize(big, layout="NCHW", out=big_img))
    out["raster_kernel 4096 images"] = {"bound": "hbm", "launch": "4096 images (2.5 GB written)", "us": t * 1e6,
                                        "achieved": IMG_BYTES * 4096 / t / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                        "frac": IMG_BYTES * 4096 / t / HBM_PEAK,
                                        "note": "streaming regime; a bare float4 store stream reaches 5.75 TB/s on this chip"}
    del big_img
    t = timed(lambda: kernels.encode_conv1(rec, net.conv1_hwio, net.conv1_base))
    out["encode_conv1_kernel"] = {"bound": "latency", "launch": "%d images" % B, "us": t * 1e6,
                                  "note": "rasterise + sparse conv1 + relu + pool + LRN; replaces %.1f MB of image traffic and "
                                          "13.5 GFLOP of dense conv1 per batch" % (IMG_BYTES * B / 1e6)}
    table = sample.table
    d_cigar, d_off, d_pos, _ = sample.device_buffers
    n = len(table)
    cap = max(1024, int(sample.gap_off[-1]) + 16)
    t = timed(lambda: kernels.cigar_scan(d_cigar, d_off, d_pos, sample.min_sv, gaps_cap=cap))
    alg = 4 * int(table.cigar.size) + 32 * n + 24 * int(sample.gap_off[-1])
    out["cigar_scan (4 kernels)"] = {"bound": "hbm", "launch": "%d alignments, %d ops" % (n, table.cigar.size), "us": t * 1e6,
                                     "achieved": alg / t / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": alg / t / HBM_PEAK}
    # the dominant kernel: fp32 MFMA implicit-GEMM convolution on the four layer shapes (2 x MAC algorithmic FLOP per launch)
    for name, cin, cout, hw, k, groups in (("conv2", 96, 256, 27, 5, 2), ("conv3", 256, 384, 13, 3, 1), ("conv4", 384, 384, 13, 3, 2),
                                           ("conv5", 384, 256, 13, 3, 2)):
        x = kernels.to_c8(torch.randn(B, cin, hw, hw, device=dev))
        w = kernels.pack_conv_weights(torch.randn(k, k, cin // groups, cout, device=dev) * 0.05)
        bias = torch.randn(cout, device=dev)
        t = timed(lambda: kernels.conv2d_same(x, w, bias, groups=groups, relu=True))
        flop = 2.0 * B * hw * hw * cout * (cin // groups) * k * k
        out["conv_igemm_kernel %s" % name] = {"bound": "mfma", "launch": "%d x %d x %d x %d -> %d, %dx%d, %d group(s)" % (B, cin, hw, hw, cout, k, k, groups),
                                               "us": t * 1e6, "achieved": flop / t / 1e12, "peak": F32_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                                               "frac": flop / t / F32_MFMA_PEAK}
    dense_net = AlexNet(random_weights(0), device=dev, active=False)
    t = timed(lambda: dense_net.predict_records(rec))
    out["device_stage_eager_1_stream_dense_convolutions"] = {
        "bound": "mfma", "launch": "%d images" % B, "us": t * 1e6, "achieved": CNN_FLOP * B / t / 1e12, "peak": F32_MFMA_PEAK / 1e12,
        "unit": "TFLOP/s", "frac": CNN_FLOP * B / t / F32_MFMA_PEAK,
        "note": "the same stage with conv2..conv5 computed at every pixel (AlexNet(active=False)); outputs bit-identical"}
    del dense_net
    t = timed(lambda: net.predict_records(rec))
    out["device_stage_eager_1_stream"] = {"bound": "mfma", "launch": "%d images" % B, "us": t * 1e6,
                                          "achieved": CNN_FLOP * B / t / 1e12, "peak": F32_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                                          "frac": CNN_FLOP * B / t / F32_MFMA_PEAK}
    return out


def cpu_baseline_windows(sample, opts, windows, gpu_net, min_seconds=12.0, max_windows=3):
    """cpu_baseline over as many windows as it takes to reach ~10-30 s of CPU work (bounded sample of the same workload)."""
    parts, sites, seconds = [], 0.0, 0.0
    for window in windows[:max_windows]:
        r = cpu_baseline(sample, opts, window, gpu_net)
        parts.append(r)
        sites += r["_sites"]
        seconds += r["_seconds"]
        if seconds >= min_seconds:
            break
    out = {k: v for k, v in parts[0].items() if not k.startswith("_")}
    out["value"] = sites / seconds
    out["sample"] = "%d window(s), %.1f sites, %.1f s CPU in total; per window: %s" % (len(parts), sites, seconds, parts[0]["sample"])
    return out


def cpu_baseline(sample, opts, window, gpu_net):
    """The oracle port of the same step on this box's host cores: C restatement of the scan and of
    the rasteriser (single thread), plain PyTorch CPU fp32 AlexNet (the reference runs CPU
    TensorFlow), the same host collection / vote code; timed on a bounded sample (one window's
    collection, the first sites' images)."""
    import io
    from oracle import cbind
    from svision_amd.collection.output_clusters import collect_pair_lines
    from svision_amd.collection.run_collection import detect_window
    from svision_amd.network.predict import Predict, SiteVoter
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    params = {}
    from oracle.alexnet_torch import TorchAlexNet
    net = TorchAlexNet(random_weights(0), device="cpu")
    table = sample.table
    chrom, start, end = window
    t0 = time.perf_counter()
    cbind.cigar_scan(table.cigar, table.cig_off.astype(np.uint64), table.pos, opts.min_sv_size)
    t_scan = (time.perf_counter() - t0) * (end - start) / max(table.lengths[0], 1)
    t0 = time.perf_counter()
    _sigs, clusters = detect_window(opts, sample, chrom, start, end)
    lines = collect_pair_lines(clusters, opts)
    t_collect = time.perf_counter() - t0
    n_sites_window = len({ln.region for ln in lines})
    # bounded sample: the first sites' images, up to ~20 s of CNN time
    B = 128                                           # reference default batch (SVision:88)
    recs = np.asarray([ln.record() for ln in lines], np.int32)
    done, t_enc_cnn, outs = 0, 0.0, []
    while done < len(lines) and t_enc_cnn < 20.0:
        t0 = time.perf_counter()
        x = cbind.rasterize(recs[done:done + B], "NCHW")
        _l, cls, prob = net.predict(torch.from_numpy(x))
        t_enc_cnn += time.perf_counter() - t0
        outs.append((cls.numpy(), prob.numpy()))
        done = min(len(lines), done + B)
    t0 = time.perf_counter()
    voter = SiteVoter(Predict(chrom, None), io.StringIO(), io.StringIO(), opts, sample)
    voter.feed_batch([ln.label() for ln in lines[:done]], np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs]))
    voter.finish()
    t_vote = time.perf_counter() - t0
    frac = done / max(len(lines), 1)
    sites = n_sites_window * frac
    total = (t_scan + t_collect) * frac + t_enc_cnn + t_vote
    return {"value": sites / total, "unit": "sites/s", "cores": threads, "kind": "port", "_sites": sites, "_seconds": total,
            "sample": "first %d of %d images (%.1f sites) of window %s:%d-%d: C oracle scan+rasteriser (1 thread), torch CPU fp32 "
                      "AlexNet batch %d on %d threads, same host collection/vote code; %.1f s CPU" %
                      (done, len(lines), sites, chrom, start, end, B, threads, total)}


if __name__ == "__main__":
    main()
