#!/usr/bin/env python3
"""Do an MFMA-bound kernel and a memory-bound kernel overlap when issued on two streams?  Loops of each alone and of
both together (wall time)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svision_amd import kernels
dev = torch.device("cuda:0")
torch.manual_seed(0)
N = int(os.environ.get("N", "200"))
x = kernels.to_c8(torch.randn(64, 256, 13, 13, device=dev).clamp_min(0))
w = kernels.pack_conv_weights(torch.randn(3, 3, 256, 384, device=dev) * 0.02)
b = torch.randn(384, device=dev)
xf = torch.randn(64, 9216, device=dev)
w6 = kernels.pack_fc_weights(torch.randn(4096, 9216, device=dev) * 0.01)
b6 = torch.randn(4096, device=dev)
w7 = kernels.pack_fc_weights(torch.randn(4096, 4096, device=dev) * 0.01)
xp = kernels.to_c8(torch.randn(64, 256, 27, 27, device=dev))
bp = torch.randn(256, device=dev)
rec = None


def conv():
    kernels.conv2d_same(x, w, b, groups=1, relu=True)


def fc():
    h = kernels.fc_bias_act(xf, w6, b6, relu=True)
    kernels.fc_bias_act(h, w7, b6, relu=True)


def pool():
    kernels.bias_relu_pool_lrn(xp, bp, lrn=True)


def graph_of(fn, stream, reps):
    with torch.cuda.stream(stream):
        fn(); fn()
    stream.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        for _ in range(reps):
            fn()
    return g


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
R = 10


def wall(jobs):
    """jobs: list of (graph, stream)"""
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(N // R):
        for g, s in jobs:
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3


for name_b, fb in (("fc6+fc7", fc), ("pool1", pool), ("conv3", conv)):
    ga, gb = graph_of(conv, s1, R), graph_of(fb, s2, R)
    for _ in range(2):
        wall([(ga, s1), (gb, s2)])
    a, bb, both = wall([(ga, s1)]), wall([(gb, s2)]), wall([(ga, s1), (gb, s2)])
    print("conv3 dense x%d alone %.2f ms | %s x%d alone %.2f ms | together %.2f ms  (sum %.2f, max %.2f)" % (N, a, name_b, N, bb, both, a + bb, max(a, bb)), flush=True)
