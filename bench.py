#!/usr/bin/env python3
"""bench.py -- candidate SV sites/sec (encode+CNN) on N MI355X of one node.

A *step* is one pass of the hot path over one batch of synthetic input that is
already resident in HBM: CIGAR/segment scan of the alignments behind the batch,
rasterisation of B candidate similarity images, AlexNet forward + argmax/softmax.
Workload: BASELINE.json configs[1] stand-in ("HiFi chr21, 1xMI355X, batch=64
candidate images"), synthetic (no real BAM ships with the reference).

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline`
(dominant kernel) and `cpu_baseline` (oracle port timed on this box, N=1 only).
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

from svision_amd import kernels  # noqa: E402
from svision_amd.network.alexnet import AlexNet, checkpoint_shapes  # noqa: E402

IMG_BYTES = 227 * 227 * 3 * 4 + 48            # 618,348 B written + 48 B read per image
CNN_FLOP = 1_440_662_592                      # FLOP per image
HBM_PEAK = 8.0e12                             # MI355X_MICROARCH.md: HBM3E 8 TB/s
F32_MFMA_PEAK = 157.3e12                      # MI355X_MICROARCH.md: FP32 matrix peak


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


def make_workload(batch, aln_per_image, seed):
    """Synthetic HiFi-like batch: `batch` segment-pair records + the CIGARs of the
    alignments behind them (15 kb reads, ~0.1-0.3 % error -> a few hundred ops each)."""
    from tests import datagen
    rec = datagen.random_records(batch, seed=seed, hostile=False)
    cigar, off, ref_start = datagen.random_cigars(batch * aln_per_image, seed=seed + 1, mean_ops=300, long_gap_rate=0.003)
    return rec, cigar, off, ref_start


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64, help="candidate images per CNN batch (BASELINE configs[1])")
    ap.add_argument("--aln-per-image", type=int, default=9, help="alignments scanned per candidate image (chr21 30x: ~9e4 reads / ~1e4 images)")
    ap.add_argument("--images-per-site", type=float, default=20.0, help="images per candidate site (SURVEY 8a cfg2 estimate: 1e4 images / 500 sites)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    B = args.batch
    # each rank owns its own shard of candidate sites (weak scaling: fixed work per GPU, no data-path collective)
    rec, cigar, off, ref_start = make_workload(B, args.aln_per_image, seed=1000 + rank)
    d_rec = torch.from_numpy(rec).to(dev)
    d_cigar = torch.from_numpy(cigar.view(np.int32)).to(dev)
    d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_ref = torch.from_numpy(ref_start).to(dev)
    net = AlexNet(random_weights(0), device=dev)
    img = torch.empty((B, 3, 227, 227), dtype=torch.float32, device=dev)

    gaps_cap = max(1024, ref_start.size)
    ev = {k: [] for k in ("scan", "raster", "cnn")}

    def step(record):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if record else None
        if record: e[0].record()
        scan = kernels.cigar_scan(d_cigar, d_off, d_ref, 50, gaps_cap=gaps_cap)
        if record: e[1].record()
        kernels.rasterize(d_rec, layout="NCHW", out=img)
        if record: e[2].record()
        logits, cls, prob = net.predict(img)
        if record:
            e[3].record()
            ev["scan"].append((e[0], e[1])); ev["raster"].append((e[1], e[2])); ev["cnn"].append((e[2], e[3]))
        return scan, cls, prob

    def sync_all():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()}
    images = B * args.steps * world
    sites = images / args.images_per_site
    n_ops = int(cigar.size)
    raster_bytes = IMG_BYTES * B
    cnn_flop = CNN_FLOP * B
    line = {
        "metric": "candidate SV sites/sec (encode+CNN)",
        "value": sites / dt,
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
        "config": {"workload": "cfg2 stand-in: HiFi chr21-like, batch=%d candidate images/step, %d alignments scanned/step, CNN fp32" % (B, B * args.aln_per_image),
                   "batch": B, "images_per_site": args.images_per_site, "images_per_s": images / dt,
                   "cigar_ops_per_step": n_ops, "parallelism": "sites sharded per GPU, no data-path collective"},
        "roofline": {"kernel": "AlexNet forward (MIOpen/hipBLASLt fp32)", "bound": "mfma", "achieved": cnn_flop / (ms["cnn"] * 1e-3) / 1e12,
                     "peak": F32_MFMA_PEAK / 1e12, "unit": "TFLOP/s", "frac": cnn_flop / (ms["cnn"] * 1e-3) / F32_MFMA_PEAK, "traffic": None},
        "roofline_kernels": {
            "raster_kernel": {"bound": "hbm", "achieved": raster_bytes / (ms["raster"] * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                              "frac": raster_bytes / (ms["raster"] * 1e-3) / HBM_PEAK, "ms": ms["raster"], "traffic": None},
            "cigar_scan": {"bound": "hbm", "achieved": (4 * n_ops + 32 * ref_start.size) / (ms["scan"] * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9,
                           "unit": "GB/s", "frac": (4 * n_ops + 32 * ref_start.size) / (ms["scan"] * 1e-3) / HBM_PEAK, "ms": ms["scan"], "traffic": None},
        },
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(rec, cigar, off, ref_start, args)
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def cpu_baseline(rec, cigar, off, ref_start, args):
    """The oracle port (C restatement of scan + rasteriser, torch-CPU fp32 AlexNet with
    the reference's batch of 128 -> here the same batch as the GPU leg) timed on this
    box's host cores on a bounded sample of the same workload."""
    from oracle import cbind
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    net = AlexNet(random_weights(0), device="cpu")
    B = rec.shape[0]
    reps, t_total = 0, 0.0
    t_budget = 15.0
    while t_total < t_budget and reps < 50:
        t0 = time.perf_counter()
        cbind.cigar_scan(cigar, off, ref_start, 50)
        x = cbind.rasterize(rec, "NCHW")
        net.predict(torch.from_numpy(x))
        t_total += time.perf_counter() - t0
        reps += 1
    sites = reps * B / args.images_per_site
    return {"value": sites / t_total, "unit": "sites/s", "cores": cores, "kind": "port",
            "sample": "%d steps of the same %d-image batch (C oracle scan+raster single thread, torch CPU fp32 AlexNet on %d threads)" % (reps, B, cores)}


if __name__ == "__main__":
    main()
