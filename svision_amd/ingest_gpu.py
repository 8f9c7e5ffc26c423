"""Device-side BAM ingestion (the default engine, svision_amd.ingest.ChromosomeFeed): the compressed file goes to the GPU as
it is -- through a ring of pinned 64 MB slots --, the BGZF blocks are inflated there (svx_bgzf_inflate_fast_on: parallel Huffman
decoding per block, then the LZ copies; the round-3 kernels by name: SVX_INFLATE_VARIANT), the CRC32 of every block is checked,
the records are located through
the .bai linear index and their fixed fields, QNAMEs and CIGAR words are packed on the device (svx_bam_walk_*).  The packed
CIGARs -- the input of svx_cigar_scan -- never leave HBM; the host receives the small per-record arrays its collection step
needs and hands them to the helper processes by WRITING them to shared-memory files (it never maps them: DESIGN.md section 5).

Why: DEFLATE decoding is the cost of ingestion (a HiFi BAM inflates to ~22 KB per read, ~0.6 GB/s per host core), and the
GPU boxes this was built on give a container the CPU time of 16 cores (svision_amd.ingest.effective_cpus): ~10 GB/s of
inflated data, a quarter of what the device pipeline consumes.  The same data inflates at 75 GB/s on the MI355X (two kernels,
136 and 170 GB/s).
Replaces pysam's AlignmentFile.fetch (run_collection.py:23-26) like the host reader (io.bam.BamStream), which stays the
engine for files without a linear index, for --hash / --graph (read bases wanted) and for SVX_INGEST=cpu.
"""
import os

import numpy as np
import torch

from . import _lib, kernels, streams
from .io.bam import AlignmentTable, read_bai_linear

FIRST_GROUP_BYTES = 192 << 20            # the first launch is small (~7 k blocks: the wave-per-block kernel): the pipeline starts after ~0.1 s
STAGE_BYTES = 64 << 20                   # a pinned staging slot of the pipelined reader (ring of eight)
PIPE_GROUP_BYTES = 400 << 20             # the SECOND group: two slices (~14 k blocks) -- a ramp 192 / 400 / 600 MB.  Round 6, measured (profiles/r06_group_sweep.txt):
                                         # the five-window chr21 job (BASELINE configs[1]) is cut [1, 2, 2] instead of [1, 4] and takes 0.142 s instead of 0.172
                                         # (its second and third window no longer wait for the inflate of all four); 20 windows, cfg1, ONT: unchanged (+-1 %)
                                         # parts_pipelined, the groups behind the first two: ~21 k blocks.  (Round 3 / early round 4: 768 MB, then 2.5 GB
LARGE_GROUP_BYTES = 600 << 20            # -- a launch of the lane-per-block kernel cost 60-90 ms whatever it held.  The two-kernel inflate is
LARGE_GROUP_BLOCKS = 94_000              # proportional to the launch, and once the read-backs no longer blocked each other (launch(), streams.py)
                                         # a steady flow of small groups beat the large ones: the CNN behind never runs out of chromosomes.)
GROUP_BYTES = 24 << 30                   # serial form (groups / decode_group): later groups as large as they come
SLICE_BYTES = 256 << 20                  # a chromosome is handed over in slices of whole collection windows of about this many compressed bytes
MIN_MARGIN = 128 << 10                   # reference bases a slice's records reach beyond its windows on either side, at least (plan_units)


# The pinned staging slots outlive a decoder: a process that reads a second file (a service, a bench's warm-up pass) finds them
# allocated -- hipHostMalloc costs ~0.1 s per GB and every other HIP call of the process waits for it.
_STAGING = {"lock": None, "slots": [], "busy": False}
_REACH_CACHE = {}                                             # (file identity) -> DeviceDecoder.estimate_reach: ~7 ms of zlib, once per file and process


def _staging_ring():
    """-> (slots, release): the process-wide ring of pinned slots if no other decoder holds it, else a private one."""
    import threading
    if _STAGING["lock"] is None:
        _STAGING["lock"] = threading.Lock()
    with _STAGING["lock"]:
        if not _STAGING["busy"]:
            _STAGING["busy"] = True

            def release():
                with _STAGING["lock"]:
                    _STAGING["busy"] = False
            return _STAGING["slots"], release
    return [], (lambda: None)


class DeviceIngestError(RuntimeError):
    """``tids``: the references the failure belongs to (None: unknown -- the reference the consumer waits for is blamed)."""
    tids = None

    def __init__(self, msg, tids=None):
        super().__init__(msg)
        self.tids = list(tids) if tids is not None else None


class LazyCigar:
    """Stand-in for ``AlignmentTable.cigar`` while the words exist on the device only.  ``attach(array)`` (owner process,
    after the spill) or ``path`` + ``ready`` (helper processes: the file the owner spills to and its completion flag)."""

    def __init__(self, n_words, path=None, ready=None):
        self.size, self.nbytes, self.path, self.ready, self._arr = int(n_words), 4 * int(n_words), path, ready, None
        self.event = None                    # owner process: set when the spill of this table is through (ingest.ChromosomeFeed)

    def attach(self, arr):
        self._arr = arr

    def _get(self):
        if self._arr is None:
            import time
            if self.path is None:
                # the owner process itself reads the words (-t 1: the by-value comparison of duplicated records): the spill runs
                # right behind the hand-over on a thread of its own -- wait for it (round 6: with the hand-over 10 ms earlier the
                # first window's collection could get here first, and the window was skipped as "failed")
                if self.event is not None:
                    self.event.wait(timeout=120)
                if self._arr is not None:
                    return self._arr
                raise RuntimeError("the CIGAR words of this table are on the device only")
            t0 = time.time()
            while not os.path.exists(self.ready):              # the owner spills them right after the hand-over
                if time.time() - t0 > 120:
                    raise RuntimeError("the CIGAR spill %s never completed" % self.path)
                time.sleep(0.001)
            self._arr = np.memmap(self.path, dtype=np.uint32, mode="r", shape=(self.size,)) if self.size else np.empty(0, np.uint32)
        return self._arr

    def __getitem__(self, key):
        return self._get()[key]

    def __len__(self):
        return self.size

    def __array__(self, dtype=None, copy=None):
        a = self._get()
        return np.asarray(a, dtype) if dtype is not None else np.asarray(a)


def spill_cigar(table):
    """Owner process, off the critical path: the device CIGAR words of a device-decoded table -> its shared-memory slot."""
    d_cigar, alloc, lazy = table._d_cigar, table._alloc, table.cigar
    if hasattr(alloc, "put"):                                   # a shared-memory slot: read back into private memory, then written to the file
        host = np.empty(lazy.size, np.uint32)
        if lazy.size:
            torch.from_numpy(host.view(np.int32)).copy_(d_cigar[:lazy.size])
        alloc.put("cigar", host)
    else:
        host = alloc("cigar", np.uint32, lazy.size)
        if lazy.size:
            torch.from_numpy(host.view(np.int32)).copy_(d_cigar[:lazy.size])
            if hasattr(host, "flush"):
                host.flush()
    lazy.attach(host)
    if getattr(alloc, "dir", None) is not None:                # the flag the helper processes wait for (LazyCigar._get)
        with open(os.path.join(alloc.dir, "cigar.ready"), "w"):
            pass


class Unit:
    """What the decoder hands over in one piece: the records of reference ``tid`` that the collection windows [lo, hi) of the
    job can touch -- every record overlapping [lo - reach, hi + reach), reach = the longest alignment's span + the longest
    read + 1000 (sample.Sample.reach: cluster extents, the coverage count and the genotyper's +-1000 bp stay inside) -- as
    the file range [vlo, vhi) of virtual offsets, both ends at linear-index entries (or the ends of the reference's records).
    A whole chromosome is one unit with lo = 0, hi = its length.  ``left_edge``: every record that reaches beyond this
    coordinate lies at or behind ``vlo`` (None: vlo is the reference's first record)."""
    __slots__ = ("tid", "lo", "hi", "vlo", "vhi", "left_edge", "to_end", "first", "last", "windows")

    def __init__(self, tid, lo, hi, vlo, vhi, left_edge=None, to_end=True, first=True, last=True, windows=None):
        self.tid, self.lo, self.hi, self.vlo, self.vhi = int(tid), int(lo), int(hi), int(vlo), int(vhi)
        self.left_edge, self.to_end, self.first, self.last = left_edge, bool(to_end), bool(first), bool(last)
        self.windows = list(windows) if windows is not None else [(self.lo, self.hi)]

    def __repr__(self):
        return "Unit(tid %d, [%d, %d)%s%s)" % (self.tid, self.lo, self.hi, "" if self.left_edge is None else ", records from %d" % self.left_edge,
                                                 "" if self.to_end else ", cut")


class MarginError(RuntimeError):
    """A slice turned out not to hold every record its windows can touch: ``needed`` = the reach its own records have."""

    def __init__(self, unit, needed):
        super().__init__("%r: its records reach %d bases, more than the margin it was cut with" % (unit, needed))
        self.unit, self.needed = unit, int(needed)


def voff_at(span, coord):
    """Linear-index look-up: the virtual offset of the first record of the reference overlapping the 16 kb bin of ``coord`` or
    a later one (SAMv1 5.1.3) -- every record that reaches beyond the start of that bin lies at or behind it --, clamped to
    the reference's records [lo, hi).  Entries of 0 (bins a writer left unfilled) are skipped."""
    lo, hi, linear = span
    if coord <= 0:
        return int(lo)
    b = int(coord) >> 14
    if b >= linear.size:
        return int(hi)
    tail = linear[b:]
    nz = tail[tail != 0]
    if nz.size == 0:
        return int(hi)
    return int(min(max(int(nz[0]), int(lo)), int(hi)))


def cut_groups(tids, size_of, first_limits, limit, merge_last=True):
    """Chromosomes in file order -> the groups whose blocks are inflated in one launch: the first groups of at most
    ``first_limits`` compressed bytes (the very first is small: the pipeline behind starts after one short launch), the others of at
    most ``limit``; a chromosome larger than its limit is a group of its own.  The two-kernel inflate is proportional to the launch
    and the read-backs of the groups no longer block each other, so a steady flow of small groups keeps the CNN behind supplied
    (the lane-per-block era made them as large as the chip: a launch cost 60-90 ms whatever it held).  ``merge_last``: a remainder
    of at most half a group joins the group in front of it -- launched on its own it is the last launch of the job, its LZ copies
    (latency-bound) run next to a CNN that has its full backlog by then (3-4 x slower than alone: 100 ms for 7 k blocks), and
    everything waits for that one chromosome."""
    groups, cur, cur_bytes = [], [], 0
    for t in tids:
        lim = first_limits[len(groups)] if len(groups) < len(first_limits) else limit
        if cur and cur_bytes + size_of(t) > lim:
            groups.append(cur)
            cur, cur_bytes = [], 0
        cur.append(t)
        cur_bytes += size_of(t)
    if cur:
        groups.append(cur)
    # (round 6: a QUARTER of a group, half until then -- with the LZ launches of round 5 a remainder of a third to a half of a group is
    # better off on its own: cfg1's [1, 2, 3, 2] slices per launch take 0.193 s, [1, 2, 5] 0.210; profiles/r06_group_sweep.txt)
    if merge_last and len(groups) > 2 and sum(size_of(t) for t in groups[-1]) * 4 <= limit:
        groups[-2:] = [groups[-2] + groups[-1]]
    return groups


class DeviceDecoder:
    def __init__(self, path, index, references, lengths, header_text, device, threads=8, alloc_for=None):
        import time
        self._t0, self.trace = time.perf_counter(), []          # (seconds since construction, what) of the first events (SVX_TIMING)
        self.path, self.references, self.lengths, self.header_text = path, list(references), list(lengths), header_text
        self.device, self.threads = torch.device(device), max(1, int(threads))
        self.alloc_for = alloc_for                               # callable() -> alloc(name, dtype, n) of the next part (shared memory)
        self.lib = _lib.load()
        self.spans = read_bai_linear(index)
        self.size = os.path.getsize(path)
        self.pinned = None
        self.stats = {"read_s": 0.0, "h2d_inflate_s": 0.0, "walk_s": 0.0, "d2h_s": 0.0, "names_s": 0.0, "blocks": 0, "bytes_in": 0, "bytes_inflated": 0}
        self._mark("index read")
        import threading
        self.first_handover = threading.Event()                 # set by the consumer once the first chromosome's scan is through

    def _mark(self, what):
        import time
        if len(self.trace) < 400:
            self.trace.append((round(time.perf_counter() - self._t0, 4), what))

    def usable(self, tids):
        return all(t < len(self.spans) and (self.spans[t] is None or self.spans[t][2].size > 0) for t in tids)

    def groups(self, tids):
        """Chromosomes that have records, in file order, cut into runs of about FIRST_GROUP_BYTES / GROUP_BYTES compressed bytes."""
        have = sorted((self.spans[t][0], t) for t in tids if t < len(self.spans) and self.spans[t] is not None)
        out, cur, cur_bytes, limit = [], [], 0, FIRST_GROUP_BYTES
        for _v, t in have:
            lo, hi, _lin = self.spans[t]
            nbytes = (hi >> 16) - (lo >> 16) + 65536
            if cur and cur_bytes + nbytes > limit:
                out.append(cur)
                cur, cur_bytes, limit = [], 0, GROUP_BYTES
            cur.append(t)
            cur_bytes += nbytes
        if cur:
            out.append(cur)
        return out

    def whole_units(self, tids):
        """One unit per reference that has records, in file order: whole chromosomes."""
        have = sorted((self.spans[t][0], t) for t in tids if t < len(self.spans) and self.spans[t] is not None)
        return [Unit(t, 0, self.lengths[t] if t < len(self.lengths) else 1 << 62, self.spans[t][0], self.spans[t][1]) for _v, t in have]

    def plan_units(self, tids, windows_of=None, margin=None, slice_bytes=None, resume=None, min_span_margins=None):
        """The references of ``tids`` that have records, in file order, each cut into slices of whole collection windows:
        ``windows_of(tid)`` -> its windows [(start, end), ...] ascending (None / a single window: the whole reference is one
        unit).  A slice is a run of windows of about ``slice_bytes`` compressed bytes (at least one window) and reads the
        records from the linear-index entry of ``start - margin`` to that of ``end + margin``: a real 30x chromosome is 3 GB
        of file, and its first window should not wait for all of it (the reference fetches window by window,
        run_collection.py:23-26).  ``margin`` is a guess (:meth:`estimate_reach`); the consumer checks every slice against the
        reach of its own records (ingest.ChromosomeFeed: MarginError -> planned again with a larger one).  ``resume`` = (tid,
        coordinate): leave out that reference's windows in front of the coordinate and every reference in front of it."""
        margin = int(margin if margin is not None else self.estimate_reach(tids))
        slice_bytes = int(slice_bytes or int(os.environ.get("SVX_SLICE_BYTES", "0")) or SLICE_BYTES)      # (the variable: experiments, tests)
        min_span = int(os.environ.get("SVX_SLICE_MIN_MARGINS", "13") if min_span_margins is None else min_span_margins) * margin
        have = sorted((self.spans[t][0], t) for t in tids if t < len(self.spans) and self.spans[t] is not None)
        units, skipping = [], resume is not None
        for _v, t in have:
            span = self.spans[t]
            wins = windows_of(t) if windows_of is not None else None
            wins = sorted((int(a), int(b)) for a, b in wins) if wins else None        # (a reference the job names no window of: whole)
            if skipping:
                if t != resume[0]:
                    continue
                skipping = False
                if wins is not None:
                    wins = [w for w in wins if w[1] > resume[1]]
                    if not wins:
                        continue
            if wins is None or os.environ.get("SVX_SLICES", "1") == "0":
                units.append(Unit(t, 0, self.lengths[t], span[0], span[1]))        # the whole chromosome (a resumed one: what is left of it is served by it)
                continue
            # a slice reads its margins twice (once with each neighbour): behind the reference's first slice -- one window, whatever
            # it costs: the pipeline waits for it -- a slice spans at least ~13 margins' worth of reference, so that long reads (ONT: a
            # margin of a megabase) do not turn a fifth of the file into overlap
            i = 0
            while i < len(wins):
                j = i + 1
                v0 = voff_at(span, wins[i][0])
                while j < len(wins) and (((voff_at(span, wins[j][1]) >> 16) - (v0 >> 16)) <= slice_bytes
                                         or (i > 0 and wins[j - 1][1] - wins[i][0] < min_span)):
                    j += 1
                lo, hi = wins[i][0], wins[j - 1][1]
                # (on the right TWICE the margin: the entry of a bin is the first record that overlaps it, which may start a whole read
                # length in front of it -- and the slice holds the records in front of THAT one)
                vlo, vhi = voff_at(span, lo - margin), (span[1] if j == len(wins) and hi >= self.lengths[t] else voff_at(span, hi + 2 * margin))
                vhi = max(vhi, vlo)
                if vhi == vlo and vlo < span[1]:
                    # Both look-ups found the SAME record: either a coverage gap (it starts behind the slice) or ONE record that spans
                    # the whole slice, margins included (a small --window_size under a long read or contig) -- then it, and every
                    # record behind it that starts inside the windows, must not be dropped as "a slice without a record" (ADVICE r5).
                    # The slice takes the records up to the next entry of the index that lies behind this one: it is never empty
                    # here, and the consumer's completeness check (ChromosomeFeed._slice_complete) sees the spanning record.
                    tail = span[2][max(0, (hi + 2 * margin) >> 14):]
                    later = tail[tail > vlo]
                    vhi = int(min(int(later[0]), int(span[1]))) if later.size else int(span[1])
                units.append(Unit(t, lo, hi, vlo, vhi, left_edge=None if vlo == span[0] else max(0, (lo - margin) >> 14 << 14), to_end=vhi == span[1],
                                  first=i == 0, last=j == len(wins), windows=wins[i:j]))
                i = j
        return units

    def estimate_reach(self, tids=None, sample=1 << 20, blocks=12):
        """A first guess of how far a slice must reach beyond its windows (:meth:`plan_units`): the records of a few 16 kb bins in
        the middle of the first reference are inflated on the host (zlib, a dozen blocks: ~9 ms) and walked -> the
        largest (reference span + read length) a log-normal fit of the sample expects among the reads crossing a window edge; at
        least MIN_MARGIN, a multiple of 16 kb.  Only a guess: reads are not of one length; every slice is checked against its own
        records afterwards."""
        if getattr(self, "_reach", None) is not None:
            return self._reach
        # (per file and process: a second pass over the same file -- a service, bench.py's legs behind its warm-up pass -- finds it)
        ident = (self.path, self.size, os.stat(self.path).st_mtime_ns, None if tids is None else min(tids, default=None))
        if ident in _REACH_CACHE:
            self._reach = _REACH_CACHE[ident]
            return self._reach
        import struct
        import zlib
        reach, logs = 0, []
        try:
            tid = min((s[0], t) for t, s in enumerate(self.spans) if s is not None and (tids is None or t in tids))[1]
            # (from the middle of the reference: the reads at its start are cut to it -- all of them short)
            v0 = voff_at(self.spans[tid], self.lengths[tid] // 2) if tid < len(self.lengths) else self.spans[tid][0]
            if v0 >= self.spans[tid][1]:
                v0 = self.spans[tid][0]
            start = v0 >> 16
            n = int(min(sample, self.size - start))
            buf = np.empty(n, np.uint8)
            if n > 0 and self.lib.svx_read_range(self.path.encode(), start, n, buf.ctypes.data, 1) == 0:
                raw, at, parts = buf.tobytes(), 0, []
                for _ in range(blocks):
                    if at + 18 > n or raw[at:at + 4] != b"\x1f\x8b\x08\x04":
                        break
                    xlen = struct.unpack_from("<H", raw, at + 10)[0]
                    bsize = struct.unpack_from("<H", raw, at + 16)[0] + 1          # (htslib writes BC as the only extra subfield)
                    if at + bsize > n:
                        break
                    parts.append(zlib.decompress(raw[at + 12 + xlen:at + bsize - 8], -15))
                    at += bsize
                data = b"".join(parts)
                span_of = np.asarray([1, 0, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0], np.int64)
                p = v0 & 0xFFFF
                while p + 36 <= len(data):
                    size, _ref, _pos, l_name, _mq, _bin, n_cig, _flag, l_seq = struct.unpack_from("<iiiBBHHHi", data, p)
                    if size < 32:
                        break
                    if p + 36 + l_name + 4 * n_cig <= len(data):         # (the CIGAR is in hand although the record's bases may not be)
                        words = np.frombuffer(data, "<u4", n_cig, p + 36 + l_name)
                        each = int(((words >> 4).astype(np.int64) * span_of[words & 15]).sum()) + max(l_seq, 0)
                        reach = max(reach, each)
                        logs.append(np.log(max(each, 1)))
                    p += 4 + size
        except Exception:                                       # noqa: BLE001 -- a guess: the default stands
            reach, logs = 0, []
        # what a window needs beyond its edges is set by the longest of the reads that cross an edge -- a few dozen at 30 x, drawn
        # with a bias towards the long ones.  A log-normal fit of (reference span + read length) over the sample -- mu, sigma of the
        # logs -- puts the largest of ~60 length-biased draws near exp(mu + sigma^2 + 2.7 sigma): 1.4 x the median for reads of one
        # length (HiFi, sigma ~0.13), ~11 x for an ONT-like tail (sigma 0.7); never less than 1.25 x the largest seen.
        guess = 1.25 * reach
        if len(logs) > 3:
            mu, sg = float(np.mean(logs)), float(np.std(logs))
            guess = min(max(guess, float(np.exp(mu + sg * sg + 2.7 * sg))), 2.0 * reach)      # (a few dozen reads: the fit is kept within 1.25-2 x the largest)
        self._reach = max(MIN_MARGIN, (int(guess) + 1000 + 16383) >> 14 << 14)
        _REACH_CACHE[ident] = self._reach
        self._mark("reach guessed from a dozen blocks in mid-reference: %d" % self._reach)
        return self._reach

    def _block_bytes(self, tid, sample=1 << 20):
        """Mean size of a BGZF block in the file (a megabyte sampled at the start of reference ``tid``): the large groups are
        cut to the number of blocks one round of the lane-per-block kernel holds, whatever the file's compression ratio."""
        start = self.spans[tid][0] >> 16
        n = int(min(sample, self.size - start))
        if n <= 0:
            return 28_000.0
        buf = np.empty(n, np.uint8)
        if self.lib.svx_read_range(self.path.encode(), start, n, buf.ctypes.data, 1) != 0:
            return 28_000.0
        cap = n // 28 + 16
        so, co = np.empty(cap, np.uint64), np.empty(cap, np.uint64)
        sl, isz = np.empty(cap, np.uint32), np.empty(cap, np.uint32)
        used = np.zeros(1, np.uint64)
        k = int(self.lib.svx_bgzf_index(buf.ctypes.data, n, start, cap, so.ctypes.data, sl.ctypes.data, isz.ctypes.data, co.ctypes.data, used.ctypes.data))
        return float(used[0]) / k if k > 0 else 28_000.0

    def _pinned(self, n):
        if self.pinned is None or self.pinned.numel() < n:
            self.pinned = torch.empty(max(n, 64 << 20), dtype=torch.uint8, pin_memory=True)
        return self.pinned

    # ---- pipelined form: a reader thread, up to `depth` groups in flight on streams of their own ------------------------
    def parts_pipelined(self, tids, depth=2):
        """:meth:`units_pipelined` over whole chromosomes: yields (finish, device arrays) per reference that has records."""
        for _unit, finish, arrays in self.units_pipelined(self.whole_units(tids), depth):
            yield finish, arrays

    def units_pipelined(self, units, depth=2):
        """Generator over the units (:meth:`plan_units` / :meth:`whole_units`: whole chromosomes or slices of them, in file
        order): (unit, finish, (d_cigar int32, d_cig_off int64 [n+1], d_pos int32)) where ``finish()`` -> AlignmentTable on the
        host (the QNAME ids are computed there: host work the caller can overlap with the next unit's device work); (unit, None,
        None) for a unit without a record.  The steps are overlapped: a reader thread fills the device buffers of the groups
        ahead (through the staging ring); every group (cut_groups: ~600 MB of file) is inflated and counted without a host
        synchronisation -- its tokens kernel on the process's "tokens" stream, one group after the other, the rest on one of
        ``depth`` group streams (svision_amd/streams.py: hardware queues of their own) --; per group ONE read-back (block
        status + walk counts: issued when the inflate is through, never queued behind it), per unit ONE (the packed arrays).
        Consecutive slices of one chromosome overlap by their margins: inside a group the overlap is inflated once (the
        group's bytes are one range of the file), across a group boundary twice (~2 % of a group)."""
        import collections
        import queue
        import threading
        import time
        lib, dev = self.lib, self.device
        units = list(units)

        def size_of(u):
            return (u.vhi >> 16) - (u.vlo >> 16) + 65536

        def mb(name, default):                                 # (experiments: SVX_FIRST_GROUP_MB / SVX_PIPE_GROUP_MB / SVX_LARGE_GROUP_MB)
            v = os.environ.get(name)
            return int(v) << 20 if v else default
        large = mb("SVX_LARGE_GROUP_MB", LARGE_GROUP_BYTES)
        large = min(large, int(LARGE_GROUP_BLOCKS * self._block_bytes(units[0].tid))) if units else large
        groups = cut_groups(units, size_of, [mb("SVX_FIRST_GROUP_MB", FIRST_GROUP_BYTES), mb("SVX_PIPE_GROUP_MB", PIPE_GROUP_BYTES)], large,
                            merge_last=os.environ.get("SVX_MERGE_LAST", "1") != "0")
        self._mark("groups cut: %s" % [len(g) for g in groups])
        q = queue.Queue(maxsize=1)
        stop = threading.Event()
        # Staging: a ring of eight pinned 64 MB slots (four, until round 4: a slot's "copy done" event sits in a hardware queue its
        # stream shares with long kernels and completes tens of ms after the copy itself -- with four slots the reader waited 0.15 s
        # of a 0.5 s job for slots whose copies had long finished, with eight 0.02 s; a stream with a hardware queue of its own
        # -- one of the low priority class, which nothing else of the process uses -- changed nothing).  A group's compressed bytes go to the device slot by slot -- read (8
        # pread threads), index the BGZF blocks the slot holds, copy them to their place in the group's device buffer on a
        # copy stream, reuse the slot once its copy is done.  (First version: one pinned buffer per group in flight --
        # 3.7 GB of hipHostMalloc at 0.1 s per GB inside the run, during which every other HIP call of the process waited.)
        ring, release_ring = _staging_ring()
        ring[:] = [r for r in ring if r[0].numel() == STAGE_BYTES]
        for r in ring:
            r[1] = None
        ring_at = [0]
        # (low class: the copies run on the DMA engines, and nothing else of the process uses that class's hardware queues: streams.py)
        copy_stream = streams.get("copy", dev)
        n_slots = max(2, int(os.environ.get("SVX_STAGE_SLOTS", "8")))

        def slot():
            if ring_at[0] < n_slots and len(ring) < n_slots:
                ring_at[0] += 1
                ring.append([torch.empty(STAGE_BYTES, dtype=torch.uint8, pin_memory=True), None])
                return ring[-1]
            s_ = ring[ring_at[0] % len(ring)]
            ring_at[0] += 1
            if s_[1] is not None:
                t_w = time.perf_counter()
                s_[1].synchronize()
                self.stats["slot_wait_s"] = self.stats.get("slot_wait_s", 0.0) + (time.perf_counter() - t_w)
            return s_

        def read_group(group):
            spans = [(u.vlo, u.vhi, self.spans[u.tid][2]) for u in group]
            tids_of = sorted({u.tid for u in group})
            c0 = min(s[0] >> 16 for s in spans)
            c1 = min(self.size, max(s[1] >> 16 for s in spans) + 65536 + 64)
            nbytes = c1 - c0
            t0 = time.perf_counter()
            self._mark("read %s: start" % group[:2])
            d_comp = torch.empty((nbytes + 31) // 16 * 16, dtype=torch.uint8, device=dev)
            off, tables, copied = 0, [], None
            while off < nbytes:
                want = min(STAGE_BYTES, nbytes - off)
                st_ = slot()
                pin = st_[0]
                t_p = time.perf_counter()
                if lib.svx_read_range(self.path.encode(), c0 + off, want, pin.data_ptr(), self.threads) != 0:
                    raise DeviceIngestError(lib.svx_bam_error().decode(), tids_of)
                self.stats["pread_s"] = self.stats.get("pread_s", 0.0) + (time.perf_counter() - t_p)
                cap = want // 28 + 16
                so, co = np.empty(cap, np.uint64), np.empty(cap, np.uint64)
                sl, isz = np.empty(cap, np.uint32), np.empty(cap, np.uint32)
                used = np.zeros(1, np.uint64)
                k = int(lib.svx_bgzf_index(pin.data_ptr(), want, c0 + off, cap, so.ctypes.data, sl.ctypes.data, isz.ctypes.data, co.ctypes.data,
                                           used.ctypes.data))
                if k < 0 or (k == 0 and off == 0):
                    raise DeviceIngestError("no BGZF block at file offset %d" % (c0 + off), tids_of)
                if k == 0:
                    break                                      # what is left of the range is the head of a block that ends behind it
                u = int(used[0])
                with torch.cuda.stream(copy_stream):
                    d_comp[off:off + u].copy_(pin[:u], non_blocking=True)
                    copied = torch.cuda.Event()
                    copied.record()
                st_[1] = copied
                tables.append((so[:k] + np.uint64(off), sl[:k], isz[:k], co[:k]))
                off += u                                       # (a block cut by the end of the slot is read again, at the head of the next one)
            d_comp.record_stream(copy_stream)
            src_off, src_len, isize, coff = (np.concatenate([t[i] for t in tables]) for i in range(4))
            nb = int(src_off.size)
            dst = np.zeros(nb + 1, np.uint64)
            dst[1:] = np.cumsum(isize.astype(np.uint64))

            def inflated_offset(voffs):
                c = voffs >> np.uint64(16)
                idx = np.searchsorted(coff, c)
                at_end = idx >= nb
                idx = np.minimum(idx, nb - 1)
                if not (at_end | (coff[idx] == c)).all():
                    raise DeviceIngestError("the index points between two BGZF blocks")
                return np.where(at_end, dst[nb], dst[idx] + (voffs & np.uint64(0xFFFF)))
            starts = []
            for u_, (lo, hi, linear) in zip(group, spans):
                seeds = linear[(linear >= np.uint64(lo)) & (linear < np.uint64(hi))]
                voffs = np.unique(np.concatenate([np.asarray([lo], np.uint64), seeds, np.asarray([hi], np.uint64)]))
                try:
                    starts.append(np.unique(inflated_offset(voffs)))
                except DeviceIngestError as exc:
                    raise DeviceIngestError(str(exc), [u_.tid]) from None
            # one pinned block of small tables: payload offsets, payload sizes, inflated offsets, then every chromosome's starts
            n_starts = [int(a.size) - 1 for a in starts]
            words = 3 * nb + 1 + sum(a.size for a in starts) + 8
            tab = torch.empty(words, dtype=torch.int64, pin_memory=True)
            tv = tab.numpy()
            tv[:nb] = src_off.view(np.int64)
            tv[nb:2 * nb] = src_len.astype(np.int64)
            tv[2 * nb:3 * nb + 1] = dst.view(np.int64)
            at, start_at = 3 * nb + 1, []
            for a in starts:
                tv[at:at + a.size] = a.view(np.int64)
                start_at.append(at)
                at += a.size
            self.stats["read_s"] += time.perf_counter() - t0
            self.stats["blocks"] += nb
            self.stats["bytes_in"] += int(nbytes)
            self.stats["bytes_inflated"] += int(dst[nb])
            self._mark("read: done, %d blocks" % nb)
            return {"group": group, "d_comp": d_comp, "copied": copied, "nbytes": nbytes, "nb": nb, "total": int(dst[nb]), "tab": tab, "start_at": start_at,
                    "n_starts": n_starts}

        def reader():
            def hand(item):                                     # stop-aware, the end marker and an exception included: an abandoned run
                while not stop.is_set():                        # (a failed group, a plan that is cut again) leaves nobody to take them,
                    try:                                        # and a reader blocked for ever keeps the process-wide staging ring busy
                        q.put(item, timeout=0.2)
                        return True
                    except queue.Full:
                        continue
                return False
            try:
                for g in groups:
                    if not hand(read_group(g)):
                        return
                hand(None)
            except BaseException as exc:                         # noqa: BLE001
                hand(exc)

        def launch(item, stream):
            nb = item["nb"]
            self._mark("launch %s: start" % item["group"][:2])
            # The group's large buffers -- the inflated bytes, the inflate's workspace -- are taken on THIS thread's (default)
            # stream and handed to the group's stream with record_stream: the caching allocator keeps its free blocks per
            # stream, so buffers allocated on the ingest streams (new ones every run) never met a cached block and were
            # hipMalloc'ed fresh -- 0.2 ms per GB on most boxes, 15 ms per GB on some: a 0.24 s stall in front of the
            # largest group's launch (two of nine bench runs).
            variant = kernels.inflate_variant_for(nb)
            d_raw = torch.empty(max(item["total"], 16), dtype=torch.uint8, device=dev)
            d_ws = kernels.inflate_workspace(lib, variant, item["total"], nb, dev)
            # (the allocator may hand out a block that default-stream work freed and is still using: the group's streams
            # order themselves behind whatever the default stream holds at this point -- normally nothing)
            allocated = torch.cuda.Event()
            allocated.record(torch.cuda.default_stream(dev))
            if tokens_stream is not None:
                tokens_stream.wait_event(allocated)
            with torch.cuda.stream(stream):
                stream.wait_event(allocated)
                stream.wait_event(item["copied"])              # the last slot of the group's compressed bytes is on the device
                d_comp = item["d_comp"]
                d_comp.record_stream(stream)
                d_raw.record_stream(stream)
                if d_ws is not None:
                    d_ws.record_stream(stream)
                d_tab = item["tab"].to(dev, non_blocking=True)
                d_status = torch.zeros(nb, dtype=torch.int32, device=dev)
                d_len = d_tab[nb:2 * nb].to(torch.int32)
                st = kernels._stream_ptr(dev)
                # every group's tokens kernel on ONE stream, in launch order (include/svx.h, svx_bgzf_inflate_fast_on): two of them
                # side by side share the chip and finish together -- late; in a row, the first group's chromosomes are out a
                # whole tokens launch earlier and its LZ copies (latency-bound) run next to the second group's tokens
                if tokens_stream is not None and variant == "fast":
                    for t_ in (d_comp, d_ws, d_tab, d_len, d_status):
                        t_.record_stream(tokens_stream)
                kernels.launch_inflate(lib, variant, d_comp.data_ptr(), d_tab.data_ptr(), d_len.data_ptr(), d_tab[2 * nb:].data_ptr(), nb,
                                       d_raw.data_ptr(), d_status.data_ptr(), item["total"], dev, ws=d_ws,
                                       tokens_stream=tokens_stream if variant == "fast" else None)
                if kernels.bgzf_crc_wanted():                  # the footers' CRC32 (htslib checks it on every block): status 9 where one differs
                    _lib.check(lib.svx_bgzf_crc32(d_raw.data_ptr(), d_tab[2 * nb:].data_ptr(), d_comp.data_ptr(), d_tab.data_ptr(), d_len.data_ptr(), nb,
                                                  d_status.data_ptr(), st), "svx_bgzf_crc32")
                total_starts = sum(item["n_starts"])
                d_counts = torch.empty((total_starts + 1, 4), dtype=torch.int64, device=dev)
                row = 0
                for at, n in zip(item["start_at"], item["n_starts"]):
                    _lib.check(lib.svx_bam_walk_count(d_raw.data_ptr(), d_tab[at:].data_ptr(), n, d_counts[row:].data_ptr(), st), "svx_bam_walk_count")
                    row += n
                d_counts[total_starts, 0] = d_status.max()
                # NO read-back is enqueued here.  A device-to-host copy goes to a DMA engine's queue at once, with a wait for
                # the kernels in front of it -- and the engine serves its queue in order: the counts' copy sat there for the
                # whole inflate (40-120 ms) and every other read-back of the process (the previous group's packed arrays,
                # the scans' results, the CNN's predictions) waited behind it.  finish_group() copies once the event is through.
                ev = torch.cuda.Event()
                ev.record()
            item.update(d_raw=d_raw, d_tab=d_tab, d_counts=d_counts, event=ev, stream=stream)
            self._mark("launched %s" % item["group"][:2])
            return item

        def finish_group(item):
            t0 = time.perf_counter()
            item["event"].synchronize()
            self._mark("inflate + count done %s" % item["group"][:2])
            item["d_comp"] = None
            self.stats["h2d_inflate_s"] += time.perf_counter() - t0
            d_counts = item.pop("d_counts")
            h_counts = torch.empty(tuple(d_counts.shape), dtype=torch.int64, pin_memory=True)
            with torch.cuda.stream(item["stream"]):
                h_counts.copy_(d_counts, non_blocking=True)
                ev_c = torch.cuda.Event()
                ev_c.record()
            ev_c.synchronize()
            counts = h_counts.numpy()
            if int(counts[-1, 0]) != 0:
                raise DeviceIngestError("corrupt BGZF blocks in %s" % item["group"], sorted({u.tid for u in item["group"]}))
            d_raw, d_tab, stream = item["d_raw"], item["d_tab"], item["stream"]
            pending, row = [], 0
            t0 = time.perf_counter()
            # Sizes first, then ONE device buffer and ONE pinned buffer per group for the packed arrays of all its chromosomes
            # (and one device buffer for their CIGAR words): every first-time hipMalloc / hipHostMalloc of a run costs
            # milliseconds during which the other threads' HIP calls -- and their page faults -- wait; per chromosome that
            # was six of them.
            plan, pack_at, word_at = [], 0, 0
            for unit_, at, n_starts in zip(item["group"], item["start_at"], item["n_starts"]):
                c = counts[row:row + n_starts]
                bad = c[:, 3] != 0
                if bad.any():
                    code = int(c[bad, 3][0])
                    raise DeviceIngestError({1: "the linear index does not match the records", 2: "malformed BAM record"}.get(code, "walk error %d" % code), [unit_.tid])
                n, words, name_bytes = (int(v) for v in c[:, :3].sum(axis=0)) if n_starts else (0, 0, 0)
                # [cig_off n+1][name_off n+1][tid n][pos n][l_seq n][flag n][mapq n][names]: everything the host wants
                sect = [8 * (n + 1), 8 * (n + 1), 4 * n, 4 * n, 4 * n, 2 * n, n, name_bytes]
                offs = np.zeros(len(sect) + 1, np.int64)
                offs[1:] = np.cumsum([(v + 15) // 16 * 16 for v in sect])
                size = (int(offs[-1]) + 16 + 255) // 256 * 256
                plan.append((at, n_starts, row, n, words, name_bytes, offs, pack_at, size, word_at, unit_))
                pack_at += size
                word_at += (max(words, 1) + 63) // 64 * 64       # (svx_cigar_scan reads 16-byte quads: every chromosome starts aligned)
                row += n_starts
            base_all = torch.zeros((max(row, 1), 3), dtype=torch.int64, pin_memory=True)
            for at, n_starts, r0, *_rest in plan:
                if n_starts > 1:
                    base_all.numpy()[r0 + 1:r0 + n_starts] = np.cumsum(counts[r0:r0 + n_starts - 1, :3], axis=0)
            d_pack_all = torch.empty(max(pack_at, 256), dtype=torch.uint8, device=dev)        # (default stream: see launch())
            d_cigar_all = torch.empty(max(word_at, 64), dtype=torch.int32, device=dev)
            allocated = torch.cuda.Event()
            allocated.record(torch.cuda.default_stream(dev))
            with torch.cuda.stream(stream):
                stream.wait_event(allocated)
                st = kernels._stream_ptr(dev)
                d_pack_all.record_stream(stream)
                d_cigar_all.record_stream(stream)
                d_base_all = base_all.to(dev, non_blocking=True)
                h_pack_all = torch.empty(max(pack_at, 256), dtype=torch.uint8, pin_memory=True)
                for at, n_starts, r0, n, words, name_bytes, offs, p0, size, w0, unit_ in plan:
                    if n == 0:                                   # a slice (or a reference) without a record: nothing to extract
                        pending.append((None, None, offs, 0, 0, 0, None, None, None, base_all, None, unit_))
                        continue
                    d_pack = d_pack_all[p0:p0 + size]
                    d_base = d_base_all[r0:r0 + n_starts]

                    def view(k, dtype, count, d_pack=d_pack, offs=offs):
                        return d_pack[int(offs[k]):int(offs[k]) + count * torch.empty(0, dtype=dtype).element_size()].view(dtype)
                    d_cig_off, d_name_off = view(0, torch.int64, n + 1), view(1, torch.int64, n + 1)
                    d_tid, d_pos, d_lseq = view(2, torch.int32, n), view(3, torch.int32, n), view(4, torch.int32, n)
                    d_flag, d_mapq, d_names = view(5, torch.int16, n), view(6, torch.uint8, n), view(7, torch.uint8, max(name_bytes, 1))
                    d_cigar = d_cigar_all[w0:w0 + max(words, 1)]
                    _lib.check(lib.svx_bam_walk_extract(d_raw.data_ptr(), d_tab[at:].data_ptr(), n_starts, d_base.data_ptr(), d_tid.data_ptr(),
                                                        d_pos.data_ptr(), d_flag.data_ptr(), d_mapq.data_ptr(), d_lseq.data_ptr(), d_cig_off.data_ptr(),
                                                        d_cigar.data_ptr(), d_name_off.data_ptr(), d_names.data_ptr(), n, st), "svx_bam_walk_extract")
                    h_pack = h_pack_all[p0:p0 + size]
                    h_pack.copy_(d_pack, non_blocking=True)
                    # svx_cigar_scan reads the offsets and positions where they are: views of the group's pack buffer, which
                    # lives as long as one of them does (until round 4: two clones and two fills of the CSR arrays' closing
                    # entries per chromosome -- four tiny launches, each a few hundred microseconds of waiting for room on a
                    # chip that is full of inflate and CNN waves; the extract kernel writes the closing entries itself now).
                    # The consumer scans on another stream and orders itself behind this one through the event only.
                    ev = torch.cuda.Event()
                    ev.record()
                    pending.append((ev, h_pack, offs, n, words, name_bytes, d_cigar, d_cig_off, d_pos, base_all, d_pack, unit_))
            self.stats["walk_s"] += time.perf_counter() - t0
            yield None                                          # every chromosome's extraction is enqueued: the caller may launch the next group
            for ev, h_pack, offs, n, words, name_bytes, d_cigar, d_cig_off, d_pos, _base, _d_pack, unit_ in pending:
                if ev is None:
                    yield unit_, None, None
                    continue
                t0 = time.perf_counter()
                ev.synchronize()
                self.stats["d2h_s"] += time.perf_counter() - t0
                self._mark("packed read-back done")
                yield unit_, self._make_finish(h_pack.numpy(), offs, n, words, name_bytes, d_cigar), (d_cigar, d_cig_off, d_pos)
            item["d_raw"] = item["d_tab"] = None

        th = threading.Thread(target=reader, name="svx-read", daemon=True)
        th.start()
        # high priority: the ingest kernels and the CNN share the device and do not overlap; whatever the order, the device
        # does the same work, but chromosomes that arrive early give the pipeline behind a backlog (and the per-chromosome
        # kernels here are small: behind queued graph replays they would wait for tens of ms)
        depth = int(os.environ.get("SVX_INGEST_DEPTH", depth))          # (experiments)
        depth = max(1, min(depth, 2))                              # (the high class has four hardware queues: tokens, two groups, scans)
        tokens_stream = streams.get("tokens", dev) if os.environ.get("SVX_TOKENS_STREAM", "1") != "0" else None
        group_streams = [streams.get("ingest%d" % i, dev) for i in range(depth)]
        inflight = collections.deque()
        state = {"done": False, "k": 0, "finished": 0, "hold_until": float("inf")}
        # (SVX_FIRST_HOLD=1: the round-3 rule -- nothing is launched behind the first group until its chromosome is through.
        # It protected that chromosome's small kernels from waiting behind the second group's inflate in a shared hardware
        # queue; with queues of their own (streams.py) it only kept the second group's tokens kernel out of the start-up, when
        # the device has nothing else to do: without it the second group's chromosomes arrive 20-30 ms earlier, 0.41 -> 0.40 s.)
        hold_first = os.environ.get("SVX_FIRST_HOLD", "0") == "1"

        def pump(block):
            """Launch what the reader has ready, up to `depth` groups in flight; block only when nothing is in flight."""
            while not state["done"] and len(inflight) < depth:
                # Nothing is launched behind the very first group until its chromosome is through (extracted here, scanned by the
                # consumer; at most 0.15 s): kernels do not pre-empt each other, these steps are a few small kernels, and behind the
                # second group's inflate launch they would wait 60-90 ms -- while the pipeline waits for exactly that chromosome.
                if hold_first and state["k"] == 1 and not (state["finished"] and self.first_handover.is_set()) and time.perf_counter() < state["hold_until"]:
                    if inflight or not block:
                        return                                  # (the caller goes on to finish the group in flight)
                    time.sleep(0.0005)
                    continue
                try:
                    item = q.get(block=block and not inflight, timeout=None)
                except queue.Empty:
                    return
                if item is None:
                    state["done"] = True
                    return
                if isinstance(item, BaseException):
                    # a LATER group failed in the reader: the groups in flight in front of it are healthy -- they are finished and
                    # their chromosomes yielded first (drive() raises this once nothing is in flight any more), so that the
                    # consumer's count of finished chromosomes points at the failing group and nothing decoded is thrown away
                    state["error"], state["done"] = item, True
                    return
                inflight.append(launch(item, group_streams[state["k"] % depth]))
                state["k"] += 1
                if state["k"] == 1:
                    state["hold_until"] = time.perf_counter() + 0.15

        # The launches are driven from a thread of their own: the consumer of this generator does host work per chromosome
        # (QNAME ids, shared-memory copies, the upload for the scan: 3-80 ms) and a generator that launches only between two
        # of its yields left the device without inflate work for as long.
        out_q = queue.Queue(maxsize=8)

        def put(x):
            while not stop.is_set():
                try:
                    out_q.put(x, timeout=0.2)
                    return True
                except queue.Full:
                    continue
            return False

        def drive():
            try:
                while not stop.is_set():
                    pump(block=True)
                    if not inflight:
                        if state.get("error") is not None:
                            raise state["error"]
                        break
                    head = inflight[0]
                    while not head["event"].query():           # keep launching while the oldest group is still on the device ...
                        pump(block=False)
                        time.sleep(0.0005)
                    inflight.popleft()
                    for part in finish_group(head):
                        if part is None:
                            state["finished"] += 1              # (its extraction is enqueued: what is launched now runs behind it)
                        elif not put(part):
                            return
                        pump(block=False)                       # (a group of a dozen chromosomes takes tens of ms to finish)
                    self._mark("group %s finished" % head["group"][:2])
                put(None)
            except BaseException as exc:                         # noqa: BLE001 -- re-raised in the consumer's thread
                put(exc)

        def warm():
            """The device buffers of the first two groups, allocated while the first read is still on its way (the caching allocator
            hands them out again): their hipMalloc calls would sit on the path to the first chromosome.  Only those: on some
            boxes a first hipMalloc costs 15 ms per GB instead of 0.2 -- 0.3 s for the 22 GB of all groups, during which no
            other HIP call of the process returns -- and the large groups' buffers are better allocated when their turn comes,
            next to device work that is already queued."""
            for g in groups[:2]:                               # in the order they will be asked for: compressed bytes, inflated bytes
                nbytes = sum(size_of(u) for u in g)
                if stop.is_set():
                    break
                for n in (nbytes + (1 << 20), 3 * nbytes + (1 << 20)):
                    torch.empty(n, dtype=torch.uint8, device=dev)      # allocated and released at once: the block stays in the allocator's cache
            self._mark("warm: done")

        threading.Thread(target=warm, name="svx-inflate-warm", daemon=True).start()
        driver = threading.Thread(target=drive, name="svx-inflate-driver", daemon=True)
        driver.start()
        try:
            while True:
                part = out_q.get()
                if part is None:
                    break
                if isinstance(part, BaseException):
                    raise part
                yield part
        finally:
            stop.set()
            try:
                th.join(timeout=10)                            # (an abandoned run: the reader may be in the middle of a group)
                copy_stream.synchronize()                      # the slots go back to the process-wide ring: no copy may still read them
            finally:
                if not th.is_alive():
                    release_ring()

    def _make_finish(self, hp, offs, n, words, name_bytes, d_cigar):
        """The host side of one device-decoded chromosome: packed read-back -> shared-memory arrays, QNAME ids, table."""
        lib = self.lib

        def sect(k, dtype, count):
            return hp[int(offs[k]):int(offs[k]) + count * np.dtype(dtype).itemsize].view(dtype)

        def finish():
            import time
            t0 = time.perf_counter()
            alloc = self.alloc_for() if self.alloc_for is not None else (lambda _name, dtype, k: np.empty(k, dtype))

            put = getattr(alloc, "put", None)                   # a shared-memory slot (ingest._slot_alloc): arrays are written to its files

            def keep(name, dtype, src):
                if put is not None:
                    return put(name, np.asarray(src, dtype))        # (a view of the group's pinned read-back buffer, which lives as long as its views)
                out = alloc(name, dtype, src.size)
                if src.size:
                    out[:] = src
                return out
            cig_off_h = keep("cig_off", np.int64, sect(0, np.int64, n + 1))
            name_off_h = sect(1, np.int64, n + 1)
            tid_h, pos_h, l_seq_h = keep("tid", np.int32, sect(2, np.int32, n)), keep("pos", np.int32, sect(3, np.int32, n)), keep("l_seq", np.int32, sect(4, np.int32, n))
            flag_h, mapq_h = keep("flag", np.uint16, sect(5, np.uint16, n)), keep("mapq", np.uint8, sect(6, np.uint8, n))
            names_h = np.ascontiguousarray(sect(7, np.uint8, name_bytes))
            name_id = np.empty(n, np.int32) if put is not None else alloc("name_id", np.int32, n)
            uniq = np.empty(max(name_bytes, 1), np.uint8)
            ub = np.zeros(1, np.uint64)
            name_off_c = np.ascontiguousarray(name_off_h)
            t1 = time.perf_counter()
            self.stats["finish_copy_s"] = self.stats.get("finish_copy_s", 0.0) + (t1 - t0)
            n_unique = int(lib.svx_name_ids(names_h.ctypes.data, name_off_c.ctypes.data, n, name_id.ctypes.data, uniq.ctypes.data, ub.ctypes.data))
            t2 = time.perf_counter()
            self.stats["finish_ids_s"] = self.stats.get("finish_ids_s", 0.0) + (t2 - t1)
            if put is not None:
                put("name_id", name_id)
                blob = put("names", uniq[:int(ub[0])].copy())
            else:
                blob = alloc("names", np.uint8, int(ub[0]))
                blob[:] = uniq[:int(ub[0])]
            name_list = blob.tobytes().decode().split("\n")[:-1] if n_unique else []
            self.stats["finish_list_s"] = self.stats.get("finish_list_s", 0.0) + (time.perf_counter() - t2)
            self.stats["names_s"] += time.perf_counter() - t0
            self._mark("finish(): slot copies + QNAME ids %.1f ms" % ((time.perf_counter() - t0) * 1e3))
            table = AlignmentTable(self.references, self.lengths, tid_h, pos_h, flag_h, mapq_h, l_seq_h, name_id, name_list, np.empty(0, np.uint32),
                                   cig_off_h, self.header_text)
            table.cigar = LazyCigar(words)
            table._names_blob = blob
            table._alloc = alloc
            table._shm_dir = getattr(alloc, "dir", None)
            table._d_cigar = d_cigar
            return table
        return finish

    def decode_group(self, tids):
        """Generator over the chromosomes of one group: (finish, (d_cigar int32, d_cig_off int64 [n+1], d_pos int32)) where
        ``finish()`` -> AlignmentTable on the host (the QNAME ids are computed there: host work the caller can overlap
        with the next chromosome's device work).  The group's blocks are inflated in ONE launch before the first yield."""
        import time
        lib, dev = self.lib, self.device
        spans = [self.spans[t] for t in tids]
        c0 = min(s[0] >> 16 for s in spans)
        c1 = min(self.size, max(s[1] >> 16 for s in spans) + 65536 + 64)
        nbytes = c1 - c0
        t0 = time.perf_counter()
        pin = self._pinned(nbytes + 64)
        if lib.svx_read_range(self.path.encode(), c0, nbytes, pin.data_ptr(), self.threads) != 0:
            raise DeviceIngestError(lib.svx_bam_error().decode())
        pin[nbytes:nbytes + 64].zero_()
        cap = nbytes // 28 + 16
        src_off, coff = np.empty(cap, np.uint64), np.empty(cap, np.uint64)
        src_len, isize = np.empty(cap, np.uint32), np.empty(cap, np.uint32)
        used = np.zeros(1, np.uint64)
        nb = int(lib.svx_bgzf_index(pin.data_ptr(), nbytes, c0, cap, src_off.ctypes.data, src_len.ctypes.data, isize.ctypes.data,
                                    coff.ctypes.data, used.ctypes.data))
        if nb <= 0:
            raise DeviceIngestError("no BGZF block at file offset %d" % c0)
        src_off, src_len, isize, coff = src_off[:nb], src_len[:nb], isize[:nb], coff[:nb]
        t1 = time.perf_counter()
        self.stats["read_s"] += t1 - t0
        padded = (nbytes + 31) // 16 * 16
        d_comp = torch.empty(padded, dtype=torch.uint8, device=dev)
        d_comp.copy_(pin[:padded], non_blocking=True)
        d_raw, d_status = kernels.bgzf_inflate(d_comp, src_off, src_len, isize)
        if int(d_status.max().item()) != 0:
            raise DeviceIngestError("%d corrupt BGZF blocks" % int(d_status.ne(0).sum().item()))
        del d_comp
        t2 = time.perf_counter()
        self.stats["h2d_inflate_s"] += t2 - t1
        self.stats["blocks"] += nb
        self.stats["bytes_in"] += int(nbytes)
        self.stats["bytes_inflated"] += int(d_raw.numel())
        dst = np.zeros(nb + 1, np.uint64)
        dst[1:] = np.cumsum(isize.astype(np.uint64))

        def inflated_offset(voffs):
            c = voffs >> np.uint64(16)
            idx = np.searchsorted(coff, c)
            at_end = idx >= nb                                   # the virtual offset of the end of the data: behind the last block
            idx = np.minimum(idx, nb - 1)
            ok = at_end | (coff[idx] == c)
            if not ok.all():
                raise DeviceIngestError("the index points between two BGZF blocks")
            return np.where(at_end, dst[nb], dst[idx] + (voffs & np.uint64(0xFFFF)))

        for t, (lo, hi, linear) in zip(tids, spans):
            seeds = linear[(linear >= np.uint64(lo)) & (linear < np.uint64(hi))]
            voffs = np.unique(np.concatenate([np.asarray([lo], np.uint64), seeds, np.asarray([hi], np.uint64)]))
            starts = np.unique(inflated_offset(voffs))           # (two virtual offsets of one byte: a block boundary)
            yield self._walk(t, d_raw, starts)

    def _walk(self, tid, d_raw, starts):
        import time
        lib, dev = self.lib, self.device
        st = kernels._stream_ptr(dev)
        t0 = time.perf_counter()
        n_starts = int(starts.size) - 1
        d_starts = torch.from_numpy(starts.view(np.int64)).to(dev)
        d_counts = torch.empty((n_starts, 4), dtype=torch.int64, device=dev)
        _lib.check(lib.svx_bam_walk_count(d_raw.data_ptr(), d_starts.data_ptr(), n_starts, d_counts.data_ptr(), st), "svx_bam_walk_count")
        counts = d_counts.cpu().numpy()
        bad = counts[:, 3] != 0
        if bad.any():
            code = int(counts[bad, 3][0])
            raise DeviceIngestError({1: "the linear index does not match the records", 2: "malformed BAM record"}.get(code, "walk error %d" % code))
        base = np.zeros((n_starts, 3), np.uint64)
        base[1:] = np.cumsum(counts[:-1, :3], axis=0).astype(np.uint64)
        n, words, name_bytes = (int(v) for v in counts[:, :3].sum(axis=0))
        d_base = torch.from_numpy(base.view(np.int64)).to(dev)
        d_tid = torch.empty(n, dtype=torch.int32, device=dev)
        d_pos = torch.empty(n, dtype=torch.int32, device=dev)
        d_flag = torch.empty(n, dtype=torch.int16, device=dev)
        d_mapq = torch.empty(n, dtype=torch.uint8, device=dev)
        d_lseq = torch.empty(n, dtype=torch.int32, device=dev)
        d_cig_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
        d_cigar = torch.empty(max(words, 1), dtype=torch.int32, device=dev)
        d_name_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
        d_names = torch.empty(max(name_bytes, 1), dtype=torch.uint8, device=dev)
        _lib.check(lib.svx_bam_walk_extract(d_raw.data_ptr(), d_starts.data_ptr(), n_starts, d_base.data_ptr(), d_tid.data_ptr(), d_pos.data_ptr(),
                                            d_flag.data_ptr(), d_mapq.data_ptr(), d_lseq.data_ptr(), d_cig_off.data_ptr(), d_cigar.data_ptr(),
                                            d_name_off.data_ptr(), d_names.data_ptr(), n, st), "svx_bam_walk_extract")
        t1 = time.perf_counter()
        alloc = self.alloc_for() if self.alloc_for is not None else (lambda _name, dtype, k: np.empty(k, dtype))

        def to_host(name, d, dtype):
            host = alloc(name, dtype, d.numel())
            if d.numel():
                torch.from_numpy(host).copy_(d)                  # (staged through the runtime's pinned bounce buffer)
            return host
        tid_h = to_host("tid", d_tid, np.int32)
        pos_h = to_host("pos", d_pos, np.int32)
        l_seq_h = to_host("l_seq", d_lseq, np.int32)
        flag_h = alloc("flag", np.uint16, n)
        if n:
            torch.from_numpy(flag_h.view(np.int16)).copy_(d_flag)
        mapq_h = to_host("mapq", d_mapq, np.uint8)
        cig_off_h = to_host("cig_off", d_cig_off, np.int64)
        # the CIGAR words stay in HBM (svx_cigar_scan reads them there).  The host has one rare use for them -- comparing
        # duplicated records by value (collection.classes.Seg.same_value) -- and gets its copy off the critical path:
        # spill_cigar() below, called by the feed after the chromosome has been handed to the pipeline
        cigar_h = LazyCigar(words)
        name_off_h = d_name_off.cpu().numpy()
        names_h = d_names[:name_bytes].cpu().numpy()
        t2 = time.perf_counter()
        self.stats["walk_s"] += t1 - t0
        self.stats["d2h_s"] += t2 - t1

        def finish():
            t2 = time.perf_counter()
            name_id = alloc("name_id", np.int32, n)
            uniq = np.empty(max(name_bytes, 1), np.uint8)
            ub = np.zeros(1, np.uint64)
            n_unique = int(lib.svx_name_ids(names_h.ctypes.data, name_off_h.ctypes.data, n, name_id.ctypes.data, uniq.ctypes.data, ub.ctypes.data))
            blob = alloc("names", np.uint8, int(ub[0]))
            blob[:] = uniq[:int(ub[0])]
            name_list = blob.tobytes().decode().split("\n")[:-1] if n_unique else []
            self.stats["names_s"] += time.perf_counter() - t2
            table = AlignmentTable(self.references, self.lengths, tid_h, pos_h, flag_h, mapq_h, l_seq_h, name_id, name_list, np.empty(0, np.uint32),
                                   cig_off_h, self.header_text)
            table.cigar = cigar_h                                # (the constructor wants an array)
            table._names_blob = blob
            table._alloc = alloc
            table._shm_dir = getattr(alloc, "dir", None)
            table._d_cigar = d_cigar
            return table
        return finish, (d_cigar[:max(words, 1)], d_cig_off, d_pos)
