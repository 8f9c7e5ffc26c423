"""The process's latency-critical HIP streams, made through the runtime itself so that each has a hardware queue of its own.

Why not ``torch.cuda.Stream(priority=-1)``: the runtime multiplexes its streams onto FOUR hardware queues per priority class
(``GPU_MAX_HW_QUEUES``), a new stream taking the least used queue of its class, and torch hands its streams out of a pool of 32 per
class -- the ingest streams, the copy stream, the scan stream, the spill stream and one rescan stream per chromosome ended up
sharing queues in no particular way.  Packets of one hardware queue are processed in order whatever stream they belong to, so a
chromosome's few small kernels (record extraction, ``svx_cigar_scan``) waited for a whole LZ launch of another group -- 35-60 ms
each time it happened, several times per job (the decoder's trace: "inflate + count done" -> "group finished" 42 ms, "finish()" ->
"scanned" 38-50 ms) -- and the pipeline behind waited with them.  Next to a long kernel on ANOTHER hardware queue the same small
kernel takes 0.04 ms at any priority (``tools/exp/prio_probe.py``).

So: the high class carries exactly four streams -- one for every group's tokens kernel (one after the other:
svx_bgzf_inflate_fast_on), two for the rest of the ingest groups in flight and one for the scans -- and nothing
else of the process asks for a high-priority stream; the copies (staging H2D, CIGAR spill D2H: DMA engines, no workgroups) go to
the LOW class, which nothing else uses either; the CNN's streams stay torch's normal-priority ones.
"""
import ctypes
import threading

import torch

_LOCK = threading.Lock()
_STREAMS = {}
_HIGH = ("tokens", "ingest0", "ingest1", "scan")
_LOW = ("copy", "spill")


def _hip_runtime():
    """The HIP runtime torch itself has loaded (its path from /proc/self/maps): a second copy of the library, found by name
    somewhere else on the search path, would hand out streams the first knows nothing about."""
    import torch  # noqa: F401 -- (loads it)
    path = None
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
    except OSError:
        pass
    return ctypes.CDLL(path or "libamdhip64.so")


def _create(hip, priority):
    handle = ctypes.c_void_p()
    rc = hip.hipStreamCreateWithPriority(ctypes.byref(handle), ctypes.c_uint(1), ctypes.c_int(priority))      # hipStreamNonBlocking
    if rc != 0 or not handle.value:
        raise RuntimeError("hipStreamCreateWithPriority(%d) failed: %d" % (priority, rc))
    return handle.value


def get(name, device):
    """-> the process-wide stream ``name`` ("tokens", "ingest0", "ingest1", "scan": high priority; "copy", "spill": low) of ``device``."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    with _LOCK:
        if key not in _STREAMS:
            hip = _hip_runtime()
            least, greatest = ctypes.c_int(), ctypes.c_int()
            with torch.cuda.device(key):
                hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest))
                made = {}
                for n in _HIGH:                                 # all of a class at once: each takes a hardware queue nobody holds yet
                    made[n] = torch.cuda.ExternalStream(_create(hip, greatest.value), device=torch.device("cuda", key))
                for n in _LOW:
                    made[n] = torch.cuda.ExternalStream(_create(hip, least.value), device=torch.device("cuda", key))
            _STREAMS[key] = made
        return _STREAMS[key][name]
