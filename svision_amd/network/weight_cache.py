"""Device-layout weights of a checkpoint, cached next to it (the cold process).

The reference rebuilds its TensorFlow graph and restores the 228 MB checkpoint for every chromosome
(/root/reference/src/network/predict.py:165-184).  Here a process loads the network once -- and a SECOND process on the same
checkpoint should not repeat what the first one computed: reading the bundle (index + 228 MB of tensor data through the
protobuf index), packing conv2..5 / fc6..7 into the kernels' layouts (include/svx.h: C8 conv weights, [n/32, k/8, 32, 8] fc
tiles), the first layer's base vector and the background activations of conv2..conv5.  All of that is a pure function of the
checkpoint's bytes, so it is kept as ONE flat float32 blob in a file whose name carries a digest of those bytes:

    <dir>/<checkpoint name>.svx-packed-<digest16>.bin      dir = $SVX_CACHE_DIR, else the checkpoint's directory when it is
                                                           writable, else ~/.cache/svision_amd

    "SVXPACK1" | u64 header bytes | JSON {format, names: {name: [shape, offset in floats]}, floats} | pad to 4096 | float32 data

Loading = mmap + one host-to-device copy + views.  A digest that does not match (another checkpoint under the same name, an edited
file), another format version or a short file -> the cache is ignored and rewritten.  SVX_WEIGHT_CACHE=0 turns it off.
The digest covers every byte of the bundle (xxh3 when the module is there: ~25 ms for 228 MB from the page cache; blake2b
otherwise), the pack format and the library's ABI version: a cache can never outlive the layouts it was written for.
"""
import json
import os
import struct

import numpy as np

MAGIC = b"SVXPACK1"
FORMAT = 1                       # bump when a packed layout or the set of cached tensors changes
_ALIGN = 64                      # floats: every tensor starts on a 256-byte boundary of the blob


def enabled():
    return os.environ.get("SVX_WEIGHT_CACHE", "1") != "0"


def bundle_files(prefix):
    """The files of a TF tensor bundle ``prefix`` (.index + .data-* shards), or [prefix] for anything else that exists."""
    d, base = os.path.dirname(prefix) or ".", os.path.basename(prefix)
    files = []
    if os.path.exists(prefix + ".index"):
        files.append(prefix + ".index")
        files += sorted(os.path.join(d, f) for f in os.listdir(d) if f.startswith(base + ".data-"))
    elif os.path.exists(prefix):
        files.append(prefix)
    return files


def checkpoint_digest(prefix, abi=0):
    """Hex digest of the bundle's bytes (+ pack format + ABI version); None when the bundle cannot be read."""
    files = bundle_files(prefix)
    if not files:
        return None
    try:
        import xxhash
        h = xxhash.xxh3_128()
    except ImportError:                                       # noqa: PERF203
        import hashlib
        h = hashlib.blake2b(digest_size=16)
    h.update(b"svx-pack-format-%d-abi-%d" % (FORMAT, int(abi)))
    try:
        for path in files:
            h.update(os.path.basename(path).encode() + b"\0")
            size = os.path.getsize(path)
            if size == 0:
                continue
            with open(path, "rb") as f:
                mm = np.memmap(f, dtype=np.uint8, mode="r")
                for lo in range(0, size, 64 << 20):
                    h.update(mm[lo:lo + (64 << 20)])
                del mm
    except OSError:
        return None
    return h.hexdigest()


def cache_path(prefix, digest):
    name = "%s.svx-packed-%s.bin" % (os.path.basename(prefix), digest[:16])
    d = os.environ.get("SVX_CACHE_DIR")
    if not d:
        own = os.path.dirname(os.path.abspath(prefix))
        d = own if os.access(own, os.W_OK) else os.path.join(os.path.expanduser("~"), ".cache", "svision_amd")
    return os.path.join(d, name)


def save(path, tensors):
    """tensors: {name: float32 array-like (numpy or CPU torch)} -> the blob file (written beside and renamed: never half a file).
    Returns True when written; a directory that cannot be written is not an error."""
    names, off, arrays = {}, 0, []
    for name, t in tensors.items():
        a = np.ascontiguousarray(t.numpy() if hasattr(t, "numpy") else t, dtype=np.float32)
        names[name] = [list(a.shape), off]
        arrays.append((off, a))
        off += (a.size + _ALIGN - 1) // _ALIGN * _ALIGN
    header = json.dumps({"format": FORMAT, "names": names, "floats": off}).encode()
    data_at = (len(MAGIC) + 8 + len(header) + 4095) // 4096 * 4096
    tmp = "%s.tmp%d" % (path, os.getpid())
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(tmp, "wb") as f:
            f.write(MAGIC + struct.pack("<Q", len(header)) + header)
            f.write(b"\0" * (data_at - f.tell()))
            blob = np.zeros(off, np.float32)
            for o, a in arrays:
                blob[o:o + a.size] = a.reshape(-1)
            f.write(blob.tobytes())
        os.replace(tmp, path)
        return True
    except OSError:
        try:
            os.remove(tmp)
        except OSError:
            pass
        return False


def load(path):
    """-> (float32 memmap of the data region, {name: (shape, offset in floats)}) or None (absent, foreign, short, another format)."""
    try:
        size = os.path.getsize(path)
        with open(path, "rb") as f:
            head = f.read(len(MAGIC) + 8)
            if len(head) < len(MAGIC) + 8 or head[:len(MAGIC)] != MAGIC:
                return None
            (hlen,) = struct.unpack("<Q", head[len(MAGIC):])
            if hlen > 1 << 20:
                return None
            meta = json.loads(f.read(hlen).decode())
        if meta.get("format") != FORMAT:
            return None
        data_at = (len(MAGIC) + 8 + hlen + 4095) // 4096 * 4096
        if size < data_at + 4 * int(meta["floats"]):
            return None
        blob = np.memmap(path, dtype=np.float32, mode="r", offset=data_at, shape=(int(meta["floats"]),))
        return blob, {k: (tuple(v[0]), int(v[1])) for k, v in meta["names"].items()}
    except (OSError, ValueError, KeyError, TypeError):
        return None
