"""Latency of a tiny kernel + pinned read-back on a high-priority stream of its own (svision_amd.streams "scan") while
N inflate launches (tokens + LZ, 28 k blocks each, back to back) run on the "ingest" streams and, optionally, the CNN stage
replays its graphs on three normal-priority streams.  argv: N (0-3), cnn (0/1)."""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from svision_amd import kernels, _lib, streams
dev = torch.device("cuda:0")
N, CNN = int(sys.argv[1]), int(sys.argv[2])
raw = np.fromfile("/tmp/scal.bam", np.uint8)
src_off, src_len, isize, _b = kernels.bgzf_block_table(raw)
padded = np.zeros((raw.size + 31) // 16 * 16, np.uint8); padded[:raw.size] = raw
d_comp = torch.from_numpy(padded).to(dev)
stop = threading.Event()
threads = []

def inflate_loop(i):
    s = streams.get("ingest%d" % i, dev)
    ws = torch.empty(int(_lib.load().svx_bgzf_inflate_fast_ws_bytes(int(isize.astype(np.uint64).sum()), len(isize))), dtype=torch.uint8, device=dev)
    with torch.cuda.stream(s):
        d_out = None
        while not stop.is_set():
            n = len(src_off)
            dst = np.zeros(n + 1, np.uint64); dst[1:] = np.cumsum(isize.astype(np.uint64))
            if d_out is None:
                d_out = torch.empty(int(dst[-1]), dtype=torch.uint8, device=dev)
                d_status = torch.zeros(n, dtype=torch.int32, device=dev)
                d_src = torch.from_numpy(src_off.view(np.int64)).to(dev); d_len = torch.from_numpy(src_len.view(np.int32)).to(dev); d_dst = torch.from_numpy(dst.view(np.int64)).to(dev)
            kernels.launch_inflate(_lib.load(), "fast", d_comp.data_ptr(), d_src.data_ptr(), d_len.data_ptr(), d_dst.data_ptr(), n, d_out.data_ptr(), d_status.data_ptr(), int(dst[-1]), dev, ws=ws)
            s.synchronize()

if CNN:
    from bench import random_weights, options_ns
    from svision_amd import synth
    from svision_amd.io import bam
    from svision_amd.sample import Sample
    from svision_amd.collection.output_clusters import collect_pair_lines
    from svision_amd.collection.run_collection import detect_window
    from svision_amd.network.alexnet import AlexNet
    from svision_amd.pipeline import DeviceStage
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

    def cnn_loop():
        while not stop.is_set():
            for _ in range(4):
                st.run(rec, out)
            torch.cuda.current_stream().synchronize()
    threads.append(threading.Thread(target=cnn_loop))
for i in range(N):
    threads.append(threading.Thread(target=inflate_loop, args=(i,)))
for t in threads:
    t.start()
time.sleep(0.5)
T = streams.get("scan", dev)
x = torch.zeros(1024, device=dev)
pin = torch.zeros(1024).pin_memory()
lat = []
for _ in range(150):
    t = time.perf_counter()
    with torch.cuda.stream(T):
        x.add_(1.0)
        pin.copy_(x, non_blocking=True)
        ev = torch.cuda.Event(); ev.record()
    ev.synchronize()
    lat.append(time.perf_counter() - t)
    time.sleep(0.004)
stop.set()
for t in threads:
    t.join()
torch.cuda.synchronize()
lat = np.asarray(lat) * 1e3
print("inflate streams %d, cnn %d: probe latency median %.2f ms, p90 %.2f, p99 %.2f, max %.2f" % (N, CNN, np.median(lat), np.percentile(lat, 90), np.percentile(lat, 99), lat.max()), flush=True)
