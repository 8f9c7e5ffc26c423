"""The two LZ kernels of svx_bgzf_inflate_fast behind their tokens kernels, by launch size: one lane per block (B, svx_lz_core.hpp)
against one wave per block (B', round 5).  HiFi-like BAM (random bases, binned qualities), its blocks tiled to the launch sizes.
python tools/exp/lz_wave_bench.py [contig Mb, default 4]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch

from svision_amd import _lib, kernels, synth
if os.environ.get("SVX_EXP_LIB"):
    _lib.LIB_PATH = os.environ["SVX_EXP_LIB"]
from svision_amd.io import bam

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 4
t, _g, _ = synth.simulate(synth.SimConfig(contigs=[("c", int(mb * 1e6))], coverage=30, seed=3), with_genome=False)
seg = bam.encode_reference_segment(t, seq="random", seed=1)
path = "/tmp/lzw.bam"
bam.write_bam_segments(path, t.references, t.lengths, [seg])
raw = np.fromfile(path, np.uint8)
src_off, src_len, isize, _b = kernels.bgzf_block_table(raw)
padded = np.zeros((raw.size + 31) // 16 * 16, np.uint8)
padded[:raw.size] = raw
d = torch.from_numpy(padded).cuda()
want = bam.bgzf_decompress(raw.tobytes())
n0 = len(isize)
print("%d blocks, %.1f MB inflated" % (n0, len(want) / 1e6), flush=True)
lib = _lib.load()


def run(variant, k, only=None):
    s, l, z = (np.concatenate([a] * k) for a in (src_off, src_len, isize))
    if only:
        os.environ["SVX_INFLATE2_ONLY"] = only
    best = 1e9
    out = None
    try:
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out, status = kernels.bgzf_inflate(d, s, l, z, wave=variant, crc=False)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
    finally:
        os.environ.pop("SVX_INFLATE2_ONLY", None)
    return best, out, status


for variant in ("fast-lane", "fast-wave"):
    _t, out, status = run(variant, 1)
    ok = not bool(status.any()) and out.cpu().numpy().tobytes() == want
    print(variant, "== zlib:", ok, flush=True)
for k in (1, 2, 4, 8, 16, 32):
    if n0 * k > 120_000:
        break
    row = ["%6d blocks" % (n0 * k)]
    for variant in ("fast-lane", "fast-wave"):
        ta, _o, _s = run(variant, k, only="A")
        tt, _o, _s = run(variant, k)
        row.append("%s: tokens %.1f ms, + LZ %.1f ms" % (variant, ta * 1e3, tt * 1e3))
    print("   ".join(row), flush=True)
