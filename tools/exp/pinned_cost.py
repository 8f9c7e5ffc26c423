import time, torch, numpy as np, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
torch.cuda.init(); torch.zeros(1).cuda()
for mb in (16, 192, 768, 768, 192):
    t = time.perf_counter(); p = torch.empty(mb << 20, dtype=torch.uint8, pin_memory=True); dt = time.perf_counter() - t
    t = time.perf_counter(); p.zero_(); dz = time.perf_counter() - t
    print("pinned alloc %4d MB: %.1f ms (first touch %.1f ms)" % (mb, dt * 1e3, dz * 1e3), flush=True)
    del p
from svision_amd import _lib
lib = _lib.load()
path = "/tmp/scal.bam"
if os.path.exists(path):
    n = os.path.getsize(path)
    pin = torch.empty(n + 64, dtype=torch.uint8, pin_memory=True)
    for th in (1, 4, 8, 16):
        t = time.perf_counter(); lib.svx_read_range(path.encode(), 0, n, pin.data_ptr(), th); dt = time.perf_counter() - t
        print("svx_read_range %d threads: %.1f MB in %.1f ms = %.1f GB/s" % (th, n / 1e6, dt * 1e3, n / dt / 1e9), flush=True)
