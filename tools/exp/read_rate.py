"""How fast does svx_read_range (pread threads: page cache -> a host buffer) go on this box?  Pinned vs pageable destination,
thread counts, before / after an fsync of the freshly written file."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from svision_amd import _lib
lib = _lib.load()
path = "/tmp/read_rate.bin"
size = 3 << 30
if not os.path.exists(path) or os.path.getsize(path) != size:
    blk = np.random.default_rng(1).integers(0, 256, 64 << 20, dtype=np.uint8).tobytes()
    with open(path, "wb") as f:
        for _ in range(size // len(blk)):
            f.write(blk)
torch.cuda.init()
slot = 64 << 20
pin = torch.empty(4 * slot, dtype=torch.uint8, pin_memory=True)
page = np.empty(4 * slot, np.uint8); page[:] = 0


def run(dst_ptr, threads, label):
    t = time.perf_counter()
    off = 0
    k = 0
    while off < size:
        n = min(slot, size - off)
        rc = lib.svx_read_range(path.encode(), off, n, dst_ptr + (k % 4) * slot, threads)
        assert rc == 0
        off += n; k += 1
    dt = time.perf_counter() - t
    print("%-34s %2d threads: %.3f s = %.1f GB/s" % (label, threads, dt, size / dt / 1e9), flush=True)


for phase in ("fresh (dirty pages)", "after fsync"):
    if phase == "after fsync":
        fd = os.open(path, os.O_RDONLY); t = time.perf_counter(); os.fsync(fd); os.close(fd)
        print("fsync %.2f s" % (time.perf_counter() - t))
    for threads in (4, 8, 16, 32):
        run(pin.data_ptr(), threads, phase + ", pinned")
    run(page.ctypes.data, 8, phase + ", pageable")
    run(page.ctypes.data, 16, phase + ", pageable")
