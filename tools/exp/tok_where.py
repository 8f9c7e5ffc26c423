"""Where does bgzf_tokens_kernel spend its time?  Kernel A alone (SVX_INFLATE2_ONLY=A) on 85 k blocks with libraries built with
-DSVX_TOK_DBG = 0 / 1 (tables built twice) / 2 (no transcode pass) / 4 (no speculative pass S1)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from svision_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
os.environ["SVX_INFLATE2_ONLY"] = "A"
import numpy as np, torch
from svision_amd import kernels
raw = np.fromfile("/tmp/scal.bam", np.uint8)
src_off, src_len, isize, _b = kernels.bgzf_block_table(raw)
padded = np.zeros((raw.size + 31) // 16 * 16, np.uint8); padded[:raw.size] = raw
d = torch.from_numpy(padded).cuda()
s, l, z = (np.concatenate([a] * 3) for a in (src_off, src_len, isize))
best = 1e9
for rep in range(4):
    torch.cuda.synchronize(); t = time.time()
    out, status = kernels.bgzf_inflate(d, s, l, z, wave="fast", crc=False)
    torch.cuda.synchronize(); best = min(best, time.time() - t)
print("%s: tokens kernel, %d blocks: %.1f ms" % (os.path.basename(sys.argv[1]), len(z), best * 1e3), flush=True)
