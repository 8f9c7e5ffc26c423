"""Which streams of one priority class share a hardware queue?  N streams made in a row with hipStreamCreateWithPriority; a long
kernel (bgzf_lz_kernel, ~32 ms, back to back) on stream i, a tiny kernel + read-back on stream j: latencies of tens of ms = the two
share a queue.  argv[1]: priority (-1 / 0 / 1), argv[2]: N."""
import ctypes, os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from svision_amd import kernels, _lib
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
prio, N = int(sys.argv[1]), int(sys.argv[2])

def stream(p):
    h = ctypes.c_void_p()
    assert hip.hipStreamCreateWithPriority(ctypes.byref(h), ctypes.c_uint(1), ctypes.c_int(p)) == 0
    return torch.cuda.ExternalStream(h.value, device=dev)

raw = np.fromfile("/tmp/scal.bam", np.uint8)
src_off, src_len, isize, _b = kernels.bgzf_block_table(raw)
padded = np.zeros((raw.size + 31) // 16 * 16, np.uint8); padded[:raw.size] = raw
d_comp = torch.from_numpy(padded).to(dev)
_ws = kernels.inflate_workspace(_lib.load(), "fast", int(isize.astype(np.uint64).sum()), len(isize), dev)
kernels.inflate_workspace = lambda *_a, **_k: _ws
kernels.bgzf_inflate(d_comp, src_off, src_len, isize, wave="fast", crc=False); torch.cuda.synchronize()
os.environ["SVX_INFLATE2_ONLY"] = "B"
S = [stream(prio) for _ in range(N)]
x = torch.zeros(1024, device=dev)
pin = torch.zeros(1024).pin_memory()
print("priority %d, %d streams; rows: long kernel on stream i; columns: probe on stream j; max latency in ms" % (prio, N))
for i in range(N):
    stop = threading.Event()
    def long_work():
        with torch.cuda.stream(S[i]):
            while not stop.is_set():
                kernels.bgzf_inflate(d_comp, src_off, src_len, isize, wave="fast", crc=False)
                S[i].synchronize()
    th = threading.Thread(target=long_work); th.start()
    time.sleep(0.05)
    row = []
    for j in range(N):
        if j == i:
            row.append("   -- ")
            continue
        lat = []
        for _ in range(25):
            t = time.perf_counter()
            with torch.cuda.stream(S[j]):
                x.add_(1.0)
                pin.copy_(x, non_blocking=True)
            S[j].synchronize()
            lat.append(time.perf_counter() - t)
            time.sleep(0.002)
        row.append("%6.1f" % (max(lat) * 1e3))
    stop.set(); th.join(); torch.cuda.synchronize()
    print("i=%d: %s" % (i, " ".join(row)), flush=True)
