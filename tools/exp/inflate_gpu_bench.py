"""GPU BGZF inflate rate on a synthetic HiFi-like BAM: svx_bgzf_inflate (one lane per block), or with the argument
`wave` svx_bgzf_inflate_wave (one wave per block)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from svision_amd import kernels, synth
from svision_amd.io import bam
WAVE = "wave" in sys.argv[1:]
path = "/tmp/scal.bam"
if not os.path.exists(path):
    table, _g, _ = synth.simulate(synth.SimConfig(contigs=[("c%d" % i, 10_000_000) for i in range(4)], coverage=30.0, seed=2), with_genome=False)
    segs = [bam.encode_reference_segment(table.subset(np.flatnonzero(table.tid == t)), seed=t) for t in range(4)]
    bam.write_bam_segments(path, table.references, table.lengths, segs)
raw = np.fromfile(path, np.uint8)
t = time.time(); src_off, src_len, isize, _b = kernels.bgzf_block_table(raw); print("block table (python) %.2f s, %d blocks" % (time.time() - t, len(isize)))
padded = np.zeros((raw.size + 31) // 16 * 16, np.uint8); padded[:raw.size] = raw
pin = torch.from_numpy(padded).pin_memory()
torch.cuda.synchronize(); t = time.time(); d = pin.cuda(non_blocking=True); torch.cuda.synchronize(); print("H2D %.1f MB in %.1f ms" % (raw.size / 1e6, (time.time() - t) * 1e3))
for rep in range(3):
    torch.cuda.synchronize(); t = time.time()
    out, status = kernels.bgzf_inflate(d, src_off, src_len, isize, wave=WAVE)
    torch.cuda.synchronize(); dt = time.time() - t
    print("inflate %.1f MB -> %.1f MB in %.1f ms = %.1f GB/s inflated; bad blocks %d" % (raw.size / 1e6, out.numel() / 1e6, dt * 1e3, out.numel() / dt / 1e9, int(status.ne(0).sum())))
# the same blocks four times over: does the rate follow the number of lanes?
k = 4
src4 = np.concatenate([src_off] * k); len4 = np.concatenate([src_len] * k); isz4 = np.concatenate([isize] * k)
for rep in range(2):
    torch.cuda.synchronize(); t = time.time()
    out, status = kernels.bgzf_inflate(d, src4, len4, isz4, wave=WAVE)
    torch.cuda.synchronize(); dt = time.time() - t
    print("x%d: %d blocks -> %.1f MB in %.1f ms = %.1f GB/s inflated; bad blocks %d" % (k, len(isz4), out.numel() / 1e6, dt * 1e3, out.numel() / dt / 1e9, int(status.ne(0).sum())))
if "check" in sys.argv[1:]:
    # the launch's whole output against zlib, byte for byte (the tests do this on the golden BAMs; this is the same at 1.85 GB)
    import zlib
    out, status = kernels.bgzf_inflate(d, src_off, src_len, isize, wave=WAVE)
    got = out.cpu().numpy()
    dst = np.zeros(len(isize) + 1, np.int64); dst[1:] = np.cumsum(isize.astype(np.int64))
    t = time.time(); bad = 0
    mv = memoryview(raw)
    for i in range(len(isize)):
        want = zlib.decompress(mv[int(src_off[i]):int(src_off[i]) + int(src_len[i])], -15)
        if want != got[dst[i]:dst[i + 1]].tobytes():
            bad += 1
            if bad < 5: print("block", i, "differs")
    print("checked %d blocks (%.1f MB) against zlib in %.1f s: %d differ" % (len(isize), dst[-1] / 1e6, time.time() - t, bad))
