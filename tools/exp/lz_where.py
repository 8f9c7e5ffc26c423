"""Where does a turn of bgzf_lz_kernel wait?  The kernel pair on 28 k blocks with a library built with SVX_LZ_DBG = 0 / 1 (far
sources not loaded) / 2 (lines not stored) / 3 (both); the tokens kernel alone (SVX_INFLATE2_ONLY_A) is subtracted by the caller."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from svision_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch
from svision_amd import kernels
raw = np.fromfile("/tmp/scal.bam", np.uint8)
src_off, src_len, isize, _b = kernels.bgzf_block_table(raw)
padded = np.zeros((raw.size + 31) // 16 * 16, np.uint8); padded[:raw.size] = raw
d = torch.from_numpy(padded).cuda()
for k in (1, 3):
    s, l, z = (np.concatenate([a] * k) for a in (src_off, src_len, isize))
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t = time.time()
        out, status = kernels.bgzf_inflate(d, s, l, z, wave="fast", crc=False)
        torch.cuda.synchronize(); best = min(best, time.time() - t)
    print("%s x%d: %.1f ms" % (os.path.basename(sys.argv[1]), k, best * 1e3), flush=True)
