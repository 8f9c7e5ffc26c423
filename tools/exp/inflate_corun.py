"""Does svx_bgzf_inflate ride along with the CNN stage?  Device stage (graph replays, 3 streams) alone, the inflate kernel
alone, and both at once on different streams; plus the pinned host-to-device rate of the box."""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from bench import random_weights, options_ns
from svision_amd import kernels, synth
from svision_amd.io import bam
from svision_amd.sample import Sample
from svision_amd.collection.output_clusters import collect_pair_lines
from svision_amd.collection.run_collection import detect_window
from svision_amd.network.alexnet import AlexNet
from svision_amd.pipeline import DeviceStage
dev = torch.device("cuda:0")
# ---- H2D
for mb in (64, 512, 2048):
    pin = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    d.copy_(pin, non_blocking=True); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        d.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    print("H2D pinned %4d MB: %.1f GB/s" % (mb, 3 * (mb << 20) / (time.perf_counter() - t) / 1e9), flush=True)
    del pin, d
# ---- workloads
path = "/tmp/scal.bam"
raw = np.fromfile(path, np.uint8)
src_off, src_len, isize, _b = kernels.bgzf_block_table(raw)
padded = np.zeros((raw.size + 31) // 16 * 16, np.uint8); padded[:raw.size] = raw
d_comp = torch.from_numpy(padded).to(dev)
k = 4
src4 = np.concatenate([src_off] * k); len4 = np.concatenate([src_len] * k); isz4 = np.concatenate([isize] * k)
B, G, L = 64, 4, 6_000_000
table, genome, _ = synth.simulate(synth.SimConfig(contigs=[("chr21", L)], coverage=30, seed=1))
sample = Sample.from_table(table, bam.Fasta(sequences=genome), 50, device=dev)
_s, clusters = detect_window(options_ns(B), sample, "chr21", 0, L)
lines = collect_pair_lines(clusters, options_ns(B))
n = (len(lines) // (B * G)) * B * G
rec = torch.from_numpy(np.asarray([ln.record() for ln in lines[:n]], np.int32)).to(dev)
net = AlexNet(random_weights(0), device=dev)
st = DeviceStage(net, B, dev, n_streams=3, launch_batches=G)
out = torch.empty((n, 6), device=dev)
st.run(rec, out); torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev)

def stage(reps):
    t = time.perf_counter()
    for _ in range(reps):
        st.run(rec, out)
    torch.cuda.synchronize()
    return time.perf_counter() - t

def inflate(reps):
    t = time.perf_counter()
    with torch.cuda.stream(side):
        for _ in range(reps):
            o, s = kernels.bgzf_inflate(d_comp, src4, len4, isz4)
    side.synchronize()
    return time.perf_counter() - t, o.numel()

reps = max(4, int(0.6 / (n / B * 0.42e-3)))
t_stage = stage(reps)
t_inf, nbytes = inflate(2)
print("stage alone: %d batches in %.3f s = %.4f ms/batch" % (reps * n // B, t_stage, t_stage / (reps * n // B) * 1e3))
print("inflate alone: %.1f GB in %.3f s = %.1f GB/s" % (2 * nbytes / 1e9, t_inf, 2 * nbytes / t_inf / 1e9), flush=True)
res = {}
th = threading.Thread(target=lambda: res.update(inf=inflate(2)))
torch.cuda.synchronize()
t0 = time.perf_counter()
th.start()
t_stage2 = stage(reps)
th.join()
t_all = time.perf_counter() - t0
print("together: stage %.3f s (%.4f ms/batch), inflate %.3f s (%.1f GB/s), wall %.3f s vs sum alone %.3f s" % (
    t_stage2, t_stage2 / (reps * n // B) * 1e3, res["inf"][0], 2 * nbytes / res["inf"][0] / 1e9, t_all, t_stage + t_inf))
