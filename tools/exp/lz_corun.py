"""What do the two inflate kernels (svx_bgzf_inflate_fast) and the CNN stage cost each other when they share the device?  Device
stage (graph replays, 3 streams) alone, the inflate pair alone (85 k blocks per launch, high-priority stream), both at once.
argv[1]: the libsvx build to use (variants: -DSVX_LZ_NT=...); SVX_INFLATE2_ONLY_A=1: the tokens kernel alone."""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from svision_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
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
# ---- workloads
path = "/tmp/scal.bam"
raw = np.fromfile(path, np.uint8)
src_off, src_len, isize, _b = kernels.bgzf_block_table(raw)
padded = np.zeros((raw.size + 31) // 16 * 16, np.uint8); padded[:raw.size] = raw
d_comp = torch.from_numpy(padded).to(dev)
k = 3
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
def masked_stream(spec):
    """A stream restricted to a subset of the 256 CUs (hipExtStreamCreateWithCUMask).  spec: "low64" = bits 0..63, "mod4" = every
    fourth bit, "mod2" = every second, "low128"."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    bits = {"low64": [i < 64 for i in range(256)], "low128": [i < 128 for i in range(256)], "mod4": [i % 4 == 0 for i in range(256)],
            "mod2": [i % 2 == 0 for i in range(256)], "mod8x3": [i % 8 < 3 for i in range(256)]}[spec]
    words = (ctypes.c_uint32 * 8)(*[sum(1 << j for j in range(32) if bits[32 * w + j]) for w in range(8)])
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(8), words)
    assert rc == 0 and h.value, rc
    return torch.cuda.ExternalStream(h.value, device=dev)

MASK = os.environ.get("LZ_CU_MASK")
side = masked_stream(MASK) if MASK else torch.cuda.Stream(device=dev, priority=-1)
print("side stream:", MASK or "high priority, all CUs", flush=True)

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
            o, s = kernels.bgzf_inflate(d_comp, src4, len4, isz4, wave="fast", crc=False)
    side.synchronize()
    return time.perf_counter() - t, o.numel(), int(s.ne(0).sum())

reps = max(4, int(0.6 / (n / B * 0.42e-3)))
# ONE workspace for every call (kernel B alone needs the streams an earlier call's kernel A left there)
_ws = kernels.inflate_workspace(_lib.load(), "fast", int(isz4.astype(np.uint64).sum()), len(isz4), dev)
kernels.inflate_workspace = lambda *_a, **_k: _ws
torch.cuda.synchronize()
for only in ((None, "B") if not MASK else ("B",)):
  if only:
    if MASK:                                             # (the streams for kernel B: one full call first)
        kernels.bgzf_inflate(d_comp, src4, len4, isz4, wave="fast", crc=False); torch.cuda.synchronize()
    os.environ["SVX_INFLATE2_ONLY"] = only              # (kernel B alone: the streams of the calls before are still in the workspace -- same block of the caching allocator)
  print("---- kernels:", only or "A + B", flush=True)
  t_stage = stage(reps)
  inflate(1)
  t_inf, nbytes, _bad = inflate(6)
  print(os.path.basename(_lib.LIB_PATH), "stage alone: %d batches in %.3f s = %.4f ms/batch" % (reps * n // B, t_stage, t_stage / (reps * n // B) * 1e3))
  print("inflate alone: %.1f GB in %.3f s = %.1f GB/s" % (6 * nbytes / 1e9, t_inf, 6 * nbytes / t_inf / 1e9), flush=True)
  res = {}
  th = threading.Thread(target=lambda: res.update(inf=inflate(6)))
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  th.start()
  t_stage2 = stage(reps)
  th.join()
  t_all = time.perf_counter() - t0
  print("status != 0:", int(res["inf"][2]), flush=True)
  print("together: stage %.3f s (%.4f ms/batch), inflate %.3f s (%.1f GB/s), wall %.3f s vs sum alone %.3f s" % (
      t_stage2, t_stage2 / (reps * n // B) * 1e3, res["inf"][0], 6 * nbytes / res["inf"][0] / 1e9, t_all, t_stage + t_inf), flush=True)
