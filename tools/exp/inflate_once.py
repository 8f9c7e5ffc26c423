"""One launch of an inflate variant (argv[1]: lds | private | wave | fast; argv[2]: how many times the file's blocks), for
rocprofv3 --kernel-trace --stats / --pmc."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from svision_amd import kernels
variant = sys.argv[1] if len(sys.argv) > 1 else None
k = float(sys.argv[2]) if len(sys.argv) > 2 else 1     # (< 1: that fraction of the file's blocks -- a small launch: the wave-per-block LZ kernel's range)
raw = np.fromfile("/tmp/scal.bam", np.uint8)
src_off, src_len, isize, _b = kernels.bgzf_block_table(raw)
src_off, src_len, isize = ((np.concatenate([a] * int(k)) if k >= 1 else a[:max(1, int(len(a) * k))]) for a in (src_off, src_len, isize))
padded = np.zeros((raw.size + 31) // 16 * 16, np.uint8); padded[:raw.size] = raw
d = torch.from_numpy(padded).cuda()
for _ in range(2):
    out, status = kernels.bgzf_inflate(d, src_off, src_len, isize, wave=variant)
    torch.cuda.synchronize()
print(out.numel(), int(status.ne(0).sum()))
