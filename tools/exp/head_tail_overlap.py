"""Can the non-MFMA head of a launch (encode_conv1 + active sets, ~105 us per 256 images) run under the MFMA tail of the launch
in front of it?  Eager launches: (A) everything on one stream, (B) heads on a high-priority stream, tails on a normal one,
(C) the same with both at normal priority, (D) two complete launches alternating on two normal streams."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from bench import random_weights, options_ns
from svision_amd import kernels, synth
from svision_amd.io import bam
from svision_amd.sample import Sample
from svision_amd.collection.output_clusters import collect_pair_lines
from svision_amd.collection.run_collection import detect_window
from svision_amd.network.alexnet import AlexNet
dev = torch.device("cuda:0")
B, L = 256, 6_000_000
table, genome, _ = synth.simulate(synth.SimConfig(contigs=[("chr21", L)], coverage=30, seed=1))
sample = Sample.from_table(table, bam.Fasta(sequences=genome), 50, device=dev)
_s, clusters = detect_window(options_ns(64), sample, "chr21", 0, L)
lines = collect_pair_lines(clusters, options_ns(64))
n = (len(lines) // B) * B
rec = torch.from_numpy(np.asarray([ln.record() for ln in lines[:n]], np.int32)).to(dev)
net = AlexNet(random_weights(0), device=dev)
net.executed = torch.zeros(5, dtype=torch.int64, device=dev)
bg = net.background()
launches = [rec[i:i + B] for i in range(0, n, B)]
print("%d launches of %d images" % (len(launches), B))


def head(r):
    x, touched = kernels.encode_conv1(r, net.conv1_hwio, net.conv1_base, touched=True)
    return (x,) + tuple(kernels.alexnet_active_sets(touched, totals=net.executed, rows=True))


def tail(h, r):
    x, l2, l3, l4, l5, counts, rows2 = h

    def conv(name, x, pixels, k, bias, relu, groups):
        return kernels.conv2d_same(x, getattr(net, name + "_w"), bias, groups=groups, relu=relu, pixels=pixels, pixel_count=counts[k:k + 1], background=bg[name])
    x = kernels.conv2d_same(x, net.conv2_w, None, groups=2, pixels=l2, pixel_count=counts[0:1],
                            out=torch.empty((r.shape[0], 32, 27, 27, 8), dtype=torch.float32, device=dev))
    x = kernels.bias_relu_pool_lrn(x, net.conv2_b, lrn=True, active_rows=rows2, background=bg["conv2"])
    x = conv("conv3", x, l3, 1, net.conv3_b, True, 1)
    x = conv("conv4", x, l4, 2, net.conv4_b, True, 2)
    x = conv("conv5", x, l5, 3, None, False, 2)
    x = kernels.bias_relu_pool_lrn(x, net.conv5_b, lrn=False)
    x = x.reshape(x.shape[0], 9216)
    x = kernels.fc_bias_act(x, net.fc6_w, net.fc6_b, relu=True)
    x = kernels.fc_bias_act(x, net.fc7_w, net.fc7_b, relu=True)
    return kernels.fc8_softmax(x, net.fc8_w, net.fc8_b)


def run_a(reps):
    outs = []
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        for r in launches:
            outs.append(tail(head(r), r))
        outs.clear()
    torch.cuda.synchronize()
    return time.perf_counter() - t


def run_b(reps, hi_prio):
    s_head = torch.cuda.Stream(device=dev, priority=-1 if hi_prio else 0)
    s_tail = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize(); t = time.perf_counter()
    keep = []
    for _ in range(reps):
        for r in launches:
            with torch.cuda.stream(s_head):
                h = head(r)
                ev = torch.cuda.Event(); ev.record()
            with torch.cuda.stream(s_tail):
                s_tail.wait_event(ev)
                for t_ in h:
                    t_.record_stream(s_tail)                   # allocated on the heads' stream, read here
                keep.append((h, tail(h, r)))
        keep.clear()
    torch.cuda.synchronize()
    return time.perf_counter() - t


def run_d(reps):
    ss = [torch.cuda.Stream(device=dev) for _ in range(2)]
    torch.cuda.synchronize(); t = time.perf_counter()
    keep = []
    for _ in range(reps):
        for i, r in enumerate(launches):
            with torch.cuda.stream(ss[i % 2]):
                keep.append(tail(head(r), r))
        keep.clear()
    torch.cuda.synchronize()
    return time.perf_counter() - t


ref = tail(head(launches[0]), launches[0]).clone()
reps = 6
for name, fn in (("A one stream", lambda: run_a(reps)), ("B heads on a high-priority stream", lambda: run_b(reps, True)),
                 ("C heads on a second normal stream", lambda: run_b(reps, False)), ("D whole launches on two streams", lambda: run_d(reps)),
                 ("A one stream (again)", lambda: run_a(reps))):
    fn()
    dt = min(fn() for _ in range(2))
    print("%-36s %.4f ms per batch of 64" % (name, dt / (reps * len(launches) * B / 64) * 1e3), flush=True)
assert torch.equal(ref, tail(head(launches[0]), launches[0]))
