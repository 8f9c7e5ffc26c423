import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from svision_amd import kernels
raw = np.fromfile("/tmp/scal.bam", np.uint8)
src_off, src_len, isize, _b = kernels.bgzf_block_table(raw)
padded = np.zeros((raw.size + 31) // 16 * 16, np.uint8); padded[:raw.size] = raw
d = torch.from_numpy(padded).cuda()
out, status = kernels.bgzf_inflate(d, src_off, src_len, isize)
torch.cuda.synchronize()
print(out.numel(), int(status.ne(0).sum()))
