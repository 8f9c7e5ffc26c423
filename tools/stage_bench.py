#!/usr/bin/env python3
"""Device stage (encode_conv1 .. fc8_softmax) on the records of real candidate sites: ms per 64-image batch as graph
replays on 1..4 streams, and -- with SVX_EXP_LIB=<libsvx built with -DSVX_CONV_EXPERIMENT> -- a coordinate sweep of the
wave-tile shape forced per convolution layer (list mode).  Used to fit the shape cost model of svx_conv.hip."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import random_weights, options_ns
from svision_amd import _lib, kernels, synth
EXP = os.environ.get("SVX_EXP_LIB")
if EXP:
    _lib.LIB_PATH = EXP
from svision_amd.io import bam
from svision_amd.sample import Sample
from svision_amd.collection.output_clusters import collect_pair_lines
from svision_amd.collection.run_collection import detect_window
from svision_amd.network.alexnet import AlexNet
from svision_amd.pipeline import DeviceStage

dev = torch.device("cuda:0")
B = 64
G = int(os.environ.get("GROUP", "1"))          # batches of 64 per launch
L = int(os.environ.get("WINDOW", "6000000"))
table, genome, _ = synth.simulate(synth.SimConfig(contigs=[("chr21", L)], coverage=30, seed=1))
sample = Sample.from_table(table, bam.Fasta(sequences=genome), 50, device=dev)
_s, clusters = detect_window(options_ns(B), sample, "chr21", 0, L)
lines = collect_pair_lines(clusters, options_ns(B))
n = (len(lines) // (B * G)) * B * G
rec = torch.from_numpy(np.asarray([ln.record() for ln in lines[:n]], np.int32)).to(dev)
print("records", n, "batches", n // B, flush=True)
net = AlexNet(random_weights(0), device=dev)

forced = {}                      # layer name -> shape id
_orig = kernels.conv2d_same
_names = {(5, 256): "conv2", (3, 384, 1): "conv3", (3, 384, 2): "conv4", (3, 256): "conv5"}


def hooked(x, w, bias=None, groups=1, **kw):
    k, cout = int(w.shape[0]), int(w.shape[3])
    name = _names.get((k, cout)) or _names.get((k, cout, groups))
    if kw.get("pixels") is not None and name in forced:
        os.environ["SVX_CONV_SHAPE"] = str(forced[name])
    else:
        os.environ.pop("SVX_CONV_SHAPE", None)
    return _orig(x, w, bias, groups=groups, **kw)


kernels.conv2d_same = hooked

if os.environ.get("FC") == "blaslt":             # A/B: the vendor GEMM instead of svx_fc_bias_act (same box, same run)
    _unpacked = {}

    def fc_lt(x, w_packed, bias, relu=True, out=None, ws=None):
        key = w_packed.data_ptr()
        if key not in _unpacked:
            nb, kq = w_packed.shape[0], w_packed.shape[1]
            _unpacked[key] = w_packed.permute(0, 2, 1, 3).reshape(nb * 32, kq * 8).contiguous()
        return torch._addmm_activation(bias, x, _unpacked[key].t(), use_gelu=False)
    kernels.fc_bias_act = fc_lt


def stage_ms(n_streams, reps=4):
    """median over `reps` passes of the whole record set (each pass = n / B batches)"""
    st = DeviceStage(net, B, dev, n_streams=n_streams, launch_batches=G)
    out = torch.empty((n, 6), device=dev)
    st.run(rec, out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        st.run(rec, out)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    return sorted(ts)[len(ts) // 2] / (n // B) * 1e3, out


base_ms = {}
ref_out = None
for ns in [int(v) for v in os.environ.get("STREAMS", "1,2,3,4").split(",")]:
    ms, out = stage_ms(ns, reps=int(os.environ.get("REPS", "4")))
    base_ms[ns] = ms
    ref_out = out.clone() if ref_out is None else ref_out
    print("streams %d: %.4f ms/batch  (default shapes)%s" % (ns, ms, "" if torch.equal(out, ref_out) else "  OUTPUT DIFFERS"), flush=True)
if EXP and os.environ.get("SWEEP", "1") == "1":
    nshapes = int(os.environ.get("SVX_N_SHAPES", "8"))
    for ns in [v for v in (1, 4) if v in base_ms]:
        for layer in ("conv2", "conv3", "conv4", "conv5"):
            row = []
            for sh in range(nshapes):
                forced.clear()
                forced[layer] = sh
                ms, out = stage_ms(ns, reps=3)
                row.append("%d:%.4f%s" % (sh, ms, "" if torch.equal(out, ref_out) else "!"))
            forced.clear()
            print("streams %d %s  base %.4f | %s" % (ns, layer, base_ms[ns], "  ".join(row)), flush=True)
