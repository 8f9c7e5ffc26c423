"""Does the BGZF inflate kernel run in the SHADOW of the matrix-bound CNN stage once the register file has room for both?

usage: corun_shadow.py LIB [blocks=65536] [variant=private] [prio=low|high|normal]

LIB: a libsvx build (svision_amd/libsvx.so, or a variant whose convolutions / fc kernels allocate 176 VGPRs -- at most two of
their waves per SIMD, 160 registers left for one inflate wave: svx_shadow.hpp).  Measured: the device stage alone (graph replays
on 3 streams), one inflate launch of `blocks` blocks alone, and the stage running while inflate launches follow each other on a
side stream."""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from svision_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
NBLOCKS = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
VARIANT = sys.argv[3] if len(sys.argv) > 3 else "private"
PRIO = sys.argv[4] if len(sys.argv) > 4 else "low"
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
path = "/tmp/scal.bam"
if not os.path.exists(path):
    table, _g, _ = synth.simulate(synth.SimConfig(contigs=[("c%d" % i, 10_000_000) for i in range(4)], coverage=30.0, seed=2), with_genome=False)
    segs = [bam.encode_reference_segment(table.subset(np.flatnonzero(table.tid == t)), seed=t) for t in range(4)]
    bam.write_bam_segments(path, table.references, table.lengths, segs)
raw = np.fromfile(path, np.uint8)
src_off, src_len, isize, _b = kernels.bgzf_block_table(raw)
padded = np.zeros((raw.size + 31) // 16 * 16, np.uint8); padded[:raw.size] = raw
d_comp = torch.from_numpy(padded).to(dev)
k = (NBLOCKS + len(isize) - 1) // len(isize)
srcN = np.concatenate([src_off] * k)[:NBLOCKS]; lenN = np.concatenate([src_len] * k)[:NBLOCKS]; iszN = np.concatenate([isize] * k)[:NBLOCKS]
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
lo_p, hi_p = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
side = torch.cuda.Stream(device=dev, priority={"low": 0, "normal": 0, "high": -1}[PRIO])
nb = n // B


def stage_chunk():
    st.run(rec, out)


def stage_for(seconds=None, until=None):
    """run stage chunks until `until()` or for `seconds`; -> (batches, seconds)"""
    torch.cuda.synchronize()
    t0 = time.perf_counter(); done = 0
    while True:
        stage_chunk(); done += nb
        torch.cuda.current_stream().synchronize()
        for s in st.streams: s.synchronize()
        now = time.perf_counter() - t0
        if (until is not None and until()) or (seconds is not None and now >= seconds):
            return done, now


def inflate(reps):
    evs = []
    with torch.cuda.stream(side):
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o, s = kernels.bgzf_inflate(d_comp, srcN, lenN, iszN, wave=VARIANT)
            e1.record()
            evs.append((e0, e1))
    side.synchronize()
    return [a.elapsed_time(b) for a, b in evs], o.numel(), int(s.ne(0).sum())


b, t = stage_for(seconds=0.4)
ms_alone = t / b * 1e3
inflate(1)
tt, nbytes, bad = inflate(2)
inf_alone = min(tt)
print("[%s] blocks %d (%s, %s prio): stage alone %.4f ms/batch | inflate alone %.1f ms = %.1f GB/s (bad %d)" % (
    os.path.basename(sys.argv[1]), NBLOCKS, VARIANT, PRIO, ms_alone, inf_alone, nbytes / inf_alone / 1e6, bad), flush=True)
res = {}
REPS = 3
th = threading.Thread(target=lambda: res.update(inf=inflate(REPS)))
torch.cuda.synchronize()
t0 = time.perf_counter()
th.start()
b2, t2 = stage_for(until=lambda: not th.is_alive())
th.join()
wall = time.perf_counter() - t0
tt2 = res["inf"][0]
print("    together: stage %.4f ms/batch over %.3f s (x%.2f) | inflate %s ms each (x%.2f) | wall %.3f s for %d batches + %d launches; serial would be %.3f s -> overlap gain %.2fx" % (
    t2 / b2 * 1e3, t2, (t2 / b2 * 1e3) / ms_alone, ["%.1f" % x for x in tt2], (sum(tt2) / REPS) / inf_alone, wall, b2, REPS,
    b2 * ms_alone * 1e-3 + REPS * inf_alone * 1e-3, (b2 * ms_alone * 1e-3 + REPS * inf_alone * 1e-3) / wall), flush=True)
