import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from svision_amd import kernels
raw = np.fromfile("/tmp/scal.bam", np.uint8)
src_off, src_len, isize, _b = kernels.bgzf_block_table(raw)
padded = np.zeros((raw.size + 31) // 16 * 16, np.uint8); padded[:raw.size] = raw
d = torch.from_numpy(padded).cuda()
for k in (1, 2, 3, 4, 6, 8):
    s = np.concatenate([src_off] * k); l = np.concatenate([src_len] * k); z = np.concatenate([isize] * k)
    best = 1e9
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time()
        out, status = kernels.bgzf_inflate(d, s, l, z, wave=(sys.argv[1] if len(sys.argv) > 1 else False))
        torch.cuda.synchronize(); best = min(best, time.time() - t)
    print("x%d: %6d blocks %5d waves: %.1f ms = %.1f GB/s" % (k, len(z), (len(z) + 63) // 64, best * 1e3, out.numel() / best / 1e9), flush=True)
    del out
