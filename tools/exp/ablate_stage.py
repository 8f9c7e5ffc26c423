#!/usr/bin/env python3
"""Which kernels cost what once replays overlap on several streams: the device stage with groups of kernels left out
(their outputs frozen from a first full run).  ms per 64-image batch."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import random_weights, options_ns
from svision_amd import kernels, synth
from svision_amd.io import bam
from svision_amd.sample import Sample
from svision_amd.collection.output_clusters import collect_pair_lines
from svision_amd.collection.run_collection import detect_window
from svision_amd.network.alexnet import AlexNet
from svision_amd.pipeline import DeviceStage

dev = torch.device("cuda:0")
B = 64
L = 6000000
table, genome, _ = synth.simulate(synth.SimConfig(contigs=[("chr21", L)], coverage=30, seed=1))
sample = Sample.from_table(table, bam.Fasta(sequences=genome), 50, device=dev)
_s, clusters = detect_window(options_ns(B), sample, "chr21", 0, L)
lines = collect_pair_lines(clusters, options_ns(B))
n = (len(lines) // B) * B
rec = torch.from_numpy(np.asarray([ln.record() for ln in lines[:n]], np.int32)).to(dev)
net = AlexNet(random_weights(0), device=dev)

net.background()               # before the wrappers: its one-image launches must not be the frozen outputs
skip = set()
ONLY_LAYER = None
frozen = {}
orig = {k: getattr(kernels, k) for k in ("encode_conv1", "alexnet_active_sets", "conv2d_same", "bias_relu_pool_lrn", "fc_bias_act", "fc8_softmax")}


def wrap(name, keyfn):
    def f(*a, **kw):
        key = (name, keyfn(*a, **kw), int(a[0].shape[0]))
        if (name in skip or (name == "conv2d_same" and ONLY_LAYER is not None and key[1] != getattr(net, ONLY_LAYER + "_w").data_ptr())) and key in frozen:
            return frozen[key]
        r = orig[name](*a, **kw)
        if key not in frozen:
            frozen[key] = r
        return r
    return f


kernels.encode_conv1 = wrap("encode_conv1", lambda *a, **kw: 0)
kernels.alexnet_active_sets = wrap("alexnet_active_sets", lambda *a, **kw: 0)
kernels.conv2d_same = wrap("conv2d_same", lambda x, w, *a, **kw: w.data_ptr())
kernels.bias_relu_pool_lrn = wrap("bias_relu_pool_lrn", lambda x, b, **kw: b.data_ptr())
kernels.fc_bias_act = wrap("fc_bias_act", lambda x, w, *a, **kw: w.data_ptr())
kernels.fc8_softmax = wrap("fc8_softmax", lambda *a, **kw: 0)


def stage_ms(n_streams, reps=4):
    st = DeviceStage(net, B, dev, n_streams=n_streams, launch_batches=int(os.environ.get('GROUP', '1')))
    out = torch.empty((n, 6), device=dev)
    st.run(rec, out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        st.run(rec, out)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    return sorted(ts)[len(ts) // 2] / (n // B) * 1e3


# a typical batch (median active-pixel count of conv2), repeated: the frozen outputs are then exact
cnt = []
for b in range(n // B):
    x, touched = orig["encode_conv1"](rec[b * B:(b + 1) * B], net.conv1_hwio, net.conv1_base, touched=True)
    cnt.append(int(orig["alexnet_active_sets"](touched)[4][0]))
pick = int(np.argsort(cnt)[len(cnt) // 2])
print("active conv2 pixels per batch: min %d median %d max %d" % (min(cnt), cnt[pick], max(cnt)))
rec = rec[pick * B:(pick + 1) * B].repeat(n // B, 1).contiguous()
net.predict_records_packed(rec[:B])          # first full eager runs freeze every output (per launch size)
net.predict_records_packed(rec[:B * int(os.environ.get('GROUP', '1'))])
torch.cuda.synchronize()
ALL = set(orig)
cases = [("full", set()), ("-encode", {"encode_conv1"}), ("-active", {"alexnet_active_sets"}), ("-conv", {"conv2d_same"}),
         ("-pool", {"bias_relu_pool_lrn"}), ("-fc", {"fc_bias_act"}), ("-fc8", {"fc8_softmax"}),
         ("only conv", ALL - {"conv2d_same"}), ("only conv+active", ALL - {"conv2d_same", "alexnet_active_sets"}),
         ("only conv2", ALL - {"conv2d_same"}), ("only conv3", ALL - {"conv2d_same"}), ("only conv4", ALL - {"conv2d_same"}), ("only conv5", ALL - {"conv2d_same"}),
         ("only fc", ALL - {"fc_bias_act"}), ("only encode", ALL - {"encode_conv1"}), ("conv+fc", ALL - {"conv2d_same", "fc_bias_act"})]
print("records", n, "batches", n // B)
only = os.environ.get("CASE")
for name, sk in (cases if only is None else [cases[int(only)]]):
    skip.clear(); skip.update(sk)
    ONLY_LAYER = name.split()[1] if name.startswith("only conv") and len(name) == 10 else None
    print("%-18s " % name + "  ".join("s%d %.4f" % (ns, stage_ms(ns)) for ns in (1, 2, 4)), flush=True)
