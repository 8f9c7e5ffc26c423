"""Latency of a tiny kernel + read-back on stream T while stream L runs the LZ kernel (32 ms launches, back to back) -- for the
priority classes of the two streams (HIP: -1 high, 0 normal, 1 low; streams made with hipStreamCreateWithPriority through ctypes:
torch hands out normal and high only).  argv: pairs "pl,pt" ..."""
import ctypes, os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from svision_amd import kernels, _lib
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
lo, hi = ctypes.c_int(), ctypes.c_int()
hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
print("priority range: least %d, greatest %d" % (lo.value, hi.value), flush=True)

def stream(prio):
    h = ctypes.c_void_p()
    rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), ctypes.c_uint(1), ctypes.c_int(prio))
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value, device=dev)

raw = np.fromfile("/tmp/scal.bam", np.uint8)
src_off, src_len, isize, _b = kernels.bgzf_block_table(raw)
padded = np.zeros((raw.size + 31) // 16 * 16, np.uint8); padded[:raw.size] = raw
d_comp = torch.from_numpy(padded).to(dev)
k = 3
s3, l3, z3 = (np.concatenate([a] * k) for a in (src_off, src_len, isize))
_ws = kernels.inflate_workspace(_lib.load(), "fast", int(z3.astype(np.uint64).sum()), len(z3), dev)
kernels.inflate_workspace = lambda *_a, **_k: _ws
kernels.bgzf_inflate(d_comp, s3, l3, z3, wave="fast", crc=False); torch.cuda.synchronize()
x = torch.zeros(1024, device=dev)
pin = torch.zeros(1024).pin_memory()
for spec in sys.argv[1:]:
    which, pl, pt = spec.split(",")
    os.environ["SVX_INFLATE2_ONLY"] = which if which in ("A", "B") else ""
    if which == "AB":
        os.environ.pop("SVX_INFLATE2_ONLY")
    L, T = stream(int(pl)), stream(int(pt))
    stop = threading.Event()
    def long_work():
        with torch.cuda.stream(L):
            while not stop.is_set():
                o, s = kernels.bgzf_inflate(d_comp, s3, l3, z3, wave="fast", crc=False)
                L.synchronize()
    th = threading.Thread(target=long_work); th.start()
    time.sleep(0.1)
    lat = []
    for _ in range(100):
        t = time.perf_counter()
        with torch.cuda.stream(T):
            x.add_(1.0)
            pin.copy_(x, non_blocking=True)
        T.synchronize()
        lat.append(time.perf_counter() - t)
        time.sleep(0.003)
    stop.set(); th.join(); torch.cuda.synchronize()
    lat = np.asarray(lat) * 1e3
    print("long = kernel %s on priority %s, probe on priority %s: latency median %.2f ms, p90 %.2f, max %.2f" % (which, pl, pt, np.median(lat), np.percentile(lat, 90), lat.max()), flush=True)
