"""The tokens kernel alone (SVX_INFLATE2_ONLY=A) at ~85 k blocks; SVX_TOK_LDS adds LDS per wave (fewer waves per CU), SVX_EXP_LIB picks a build."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from svision_amd import _lib, kernels, synth
if os.environ.get("SVX_EXP_LIB"):
    _lib.LIB_PATH = os.environ["SVX_EXP_LIB"]
from svision_amd.io import bam
t, _g, _ = synth.simulate(synth.SimConfig(contigs=[("c", 4_000_000)], coverage=30, seed=3), with_genome=False)
seg = bam.encode_reference_segment(t, seq="random", seed=1)
bam.write_bam_segments("/tmp/tok.bam", t.references, t.lengths, [seg])
raw = np.fromfile("/tmp/tok.bam", np.uint8)
src_off, src_len, isize, _b = kernels.bgzf_block_table(raw)
padded = np.zeros((raw.size + 31) // 16 * 16, np.uint8); padded[:raw.size] = raw
d = torch.from_numpy(padded).cuda()
want = bam.bgzf_decompress(raw.tobytes())
out, status = kernels.bgzf_inflate(d, src_off, src_len, isize, wave="fast-lane", crc=False)
ok = not bool(status.any()) and out.cpu().numpy().tobytes() == want
k = max(1, 85_000 // len(isize))
s, l, z = (np.concatenate([a] * k) for a in (src_off, src_len, isize))
os.environ["SVX_INFLATE2_ONLY"] = "A"
best = 1e9
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    kernels.bgzf_inflate(d, s, l, z, wave="fast-lane", crc=False)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print("lds+%s lib %s: == zlib %s, %d blocks, tokens %.1f ms" % (os.environ.get("SVX_TOK_LDS", "0"), os.path.basename(_lib.LIB_PATH), ok, len(z), best * 1e3), flush=True)
