"""Chromosome-by-chromosome ingestion of a BAM file, overlapped with the pipeline.

The reference opens the BAM in every pool worker and fetches one window at a time
(run_collection.py:23-26).  Here a rank's chromosomes are streamed once, in file order, by the
native reader's own threads (svx_bam_stream_*, libdeflate on host cores); each chromosome's packed
arrays land in shared memory, are uploaded and scanned on the device (svx_cigar_scan) by a feeder
thread on a stream of its own, and are handed to the forked host helpers by name -- while the
windows of the chromosome before are still in the pipeline.  Nothing of the file other than the
chromosomes in flight is resident.

Two feeds, one interface (``poll`` / ``get`` / ``release``):

  StaticFeed      a Sample that is already complete (tests, ``bench.py`` with the alignments resident in HBM)
  ChromosomeFeed  the file-driven one (the command line, ``bench.py --from-bam``)
"""
import itertools
import os
import queue
import shutil
import tempfile
import threading
import time

import numpy as np

from .io.bam import AlignmentTable, BamStream
from . import streams
from .sample import Sample

_KEYS = itertools.count()                                    # keys of the parts announced to the helper processes (pipeline.PooledHotPath)


def effective_cpus():
    """CPUs this process can actually use: its affinity mask, capped by the cgroup's CPU-time quota (a container that
    sees 256 CPUs may be allowed the time of 16: more runnable threads than that get the whole group throttled for the
    rest of every scheduling period, the GPU-feeding thread included).  -> (usable CPUs, visible CPUs)."""
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:              # cgroup v2: "<quota> <period>" or "max <period>"
            q, period = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, period = int(f.read()), int(g.read())
                if q > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    usable = visible if quota is None else max(1, min(visible, int(quota)))
    return usable, visible


def decode_threads(ranks_on_node=1, helpers=1):
    """Inflate threads of one rank: twice the node's usable CPUs divided by its ranks, at most 128.  The helpers wait for
    the decoder whenever it is the limit, so nothing is subtracted for them; a CPU-time quota is enforced per 100 ms period,
    and more runnable threads than CPUs use a period's budget earlier (measured on a 16-CPU quota, the same 9.3 GB of
    inflated data: 7 threads 2.2 s, 14 threads 1.11-1.4 s, 21 threads 1.05 s, 56 threads 0.95 s)."""
    usable, _visible = effective_cpus()
    return max(2, min(128, 2 * usable // max(ranks_on_node, 1)))


def empty_sample(references, lengths, fasta, min_sv, header_text=""):
    """A chromosome without records: its windows still exist as tasks (SVision:172-201) and yield nothing."""
    from . import kernels
    table = AlignmentTable(references, lengths, [], [], [], [], [], [], [], [], [0], header_text)
    return Sample.with_scan(table, fasta, min_sv, (np.empty(0, kernels.GAP_DTYPE), np.zeros(1, np.int64), np.empty((0, 4), np.int32)))


class StaticFeed:
    """Every chromosome is served by one resident Sample (key None: the helpers have it since their fork / attach_scan)."""

    def __init__(self, sample):
        self.sample = sample

    def poll(self, block=False):
        return False

    def take_fresh(self):
        return []

    def get(self, chrom, block=True, start=None):
        return None, self.sample

    def keys_of(self, chrom):
        return []

    def release(self, chrom):
        pass

    def close(self):
        pass


class ChromosomeFeed:
    """See the module docstring.  ``chroms``: this rank's chromosomes in task order; ``references`` / ``lengths``: the
    BAM header's dictionary.  ``stats`` afterwards: seconds the feeder thread spent waiting for the decoder, uploading +
    scanning, and blocked on a full hand-over queue; bytes of packed CIGAR uploaded."""

    def __init__(self, bam_path, fasta, options, chroms, references, lengths, device="cuda", index=None, threads=0, depth=2,
                 engine=None, header_text="", tasks=None):
        self.bam_path, self.fasta, self.options = bam_path, fasta, options
        # tasks: {chromosome: [[start, end], ...]} -- the job's collection windows (cli.build_tasks).  With them the device engine
        # hands a chromosome over in SLICES of whole windows (ingest_gpu.DeviceDecoder.plan_units) instead of in one piece: the
        # first window of a 3 GB chromosome starts after its first ~200 MB, as the reference's window-by-window fetch does
        # (run_collection.py:23-26).  None: whole chromosomes.
        self.tasks = None if tasks is None else {c: [(int(a), int(b)) for a, b in w] for c, w in tasks.items()}
        self.header_text = header_text
        # where the BGZF blocks are inflated: "gpu" = on the device (ingest_gpu.DeviceDecoder), "cpu" = libdeflate on host
        # threads (io.bam.BamStream), "auto" (the default; SVX_INGEST overrides) = the device whenever it can: the file has a
        # .bai with its linear index and the run needs no read bases (--hash / --graph).  On a 16-CPU GPU box the device engine
        # takes 0.6 s for the 3.9 GB of the bench's 20-window file, the host engine 1.15 s (DESIGN.md section 5 "Device-side
        # ingestion"); a reference the device engine cannot take (a linear index that does not match, a corrupt block) falls back to the
        # host engine chromosome by chromosome.
        self.engine = engine or os.environ.get("SVX_INGEST", "auto")
        self.references, self.lengths = list(references), list(lengths)
        self.chroms = list(chroms)
        self.device, self.index, self.threads = device, index, threads
        self.with_seq = bool(options.hash or getattr(options, "graph", False))
        shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        self.root = tempfile.mkdtemp(prefix="svx_feed_", dir=shm)
        self.handover = queue.Queue(maxsize=depth)           # feeder thread -> owner thread
        self.samples = {}                                      # chrom -> [[lo, hi, key, Sample, meta], ...] (ascending; a whole chromosome: one entry) once the owner has seen them
        self.fresh = []                                        # (key, meta) not yet announced to the helpers
        self.error = None
        self.finished = False
        self.stats = {"engine": None, "decode_wait_s": 0.0, "upload_scan_s": 0.0, "handover_wait_s": 0.0, "cigar_bytes": 0, "records": 0,
                      "first_ready_s": None, "last_ready_s": None, "slices": 0, "replans": 0}
        self._epoch, self._replan, self._acked = 0, None, (0, 0)   # slices cut again with a larger margin (_run -> _decode)
        self._t0 = time.perf_counter()
        self._stop = False
        self._b_done = False                                   # stage B has left its loop (whatever the reason): nothing will be acknowledged any more
        self._slot_lock, self._free_slots, self._n_slots = threading.Lock(), [], 0
        self.thread = threading.Thread(target=self._run, name="svx-feed", daemon=True)
        self.thread.start()

    # ---- shared-memory slots ---------------------------------------------------------------------------------------
    # A chromosome's arrays live in the files of one slot directory; a released slot is handed to a later chromosome and
    # its files are overwritten in place: their pages stay allocated (a fresh tmpfs page costs a fault and a memset --
    # writing 270 MB of new files took 0.19 s, a sixth of a 20-window job).
    def _slot_alloc(self):
        with self._slot_lock:
            d = self._free_slots.pop() if self._free_slots else None
            if d is None:
                d = os.path.join(self.root, "s%d" % self._n_slots)
                self._n_slots += 1
                os.makedirs(d)
        flag = os.path.join(d, "cigar.ready")
        if os.path.exists(flag):
            os.remove(flag)
        arrays = {}

        def alloc(name, dtype, n):
            dtype = np.dtype(dtype)
            arrays[name] = (dtype.str, int(n))
            if n == 0:
                return np.empty(0, dtype)
            path = os.path.join(d, name + ".bin")
            need = int(n) * dtype.itemsize
            mode = "r+" if os.path.exists(path) and os.path.getsize(path) >= need else "w+"
            if mode == "w+" and os.path.exists(path):
                os.remove(path)
            return np.memmap(path, dtype=dtype, mode=mode, shape=(int(n),))

        def put(name, src):
            """The device engine's way into a slot: the array is WRITTEN to the slot's file and the caller keeps using its own
            copy -- no mapping is created in this process.  np.memmap here cost the first file-driven pass of a process two
            stalls of 0.1-0.2 s: mmap() needs the address-space lock for writing, and while the process's first hipMalloc /
            hipHostMalloc calls and the runtime's page pinning hold it, the caller -- the one thread every chromosome passes
            through -- stood in mmap() (periodic stack dumps: numpy memmap.__new__, four in a row).  The helpers map the
            files in their own processes, which have no HIP runtime."""
            src = np.ascontiguousarray(src).reshape(-1)
            arrays[name] = (src.dtype.str, int(src.size))
            if src.size:
                path = os.path.join(d, name + ".bin")
                reuse = os.path.exists(path) and os.path.getsize(path) >= src.nbytes
                with open(path, "r+b" if reuse else "wb") as f:
                    f.write(src.view(np.uint8))
            return src
        alloc.dir, alloc.arrays, alloc.put = d, arrays, put
        return alloc

    def _slot_free(self, d):
        with self._slot_lock:
            self._free_slots.append(d)

    def _run(self):
        """Stage B of the feeder (this thread): QNAME ids, upload (host engine) + device scan, the check that a slice holds
        every record its windows can touch, hand-over.  Stage A (a thread of its own, :meth:`_decode`) reads / inflates /
        packs the next chromosomes / slices meanwhile."""
        import torch
        decoded = queue.Queue(maxsize=2)
        try:
            tids = [self.references.index(c) for c in self.chroms]
            want = list(tids)                                  # chromosomes not completely handed over yet, in task order
            covered = {}                                       # tid -> the coordinate up to which its windows have been handed over
            if not torch.cuda.is_available():
                raise RuntimeError("ChromosomeFeed needs the GPU (svx_cigar_scan); there is no CPU fallback")
            engine = self.engine
            if engine == "auto":
                engine = "gpu"
            if engine == "gpu" and (self.index is None or self.with_seq or not str(self.device).startswith("cuda")):
                engine = "cpu"
            self.stats["engine"] = engine
            stage_a = threading.Thread(target=self._decode, args=(engine, tids, decoded), name="svx-decode", daemon=True)
            stage_a.start()
            scan_stream = streams.get("scan", self.device)      # (a hardware queue of its own: svision_amd/streams.py)
            spill = queue.Queue()
            threading.Thread(target=self._spill, args=(spill,), name="svx-spill", daemon=True).start()
            inf = float("inf")
            while not self._stop:
                t0 = time.perf_counter()
                try:
                    item = decoded.get(timeout=0.2)
                except queue.Empty:
                    self.stats["decode_wait_s"] += time.perf_counter() - t0
                    continue
                self.stats["decode_wait_s"] += time.perf_counter() - t0
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                epoch, unit, table, arrays = item
                if epoch != self._epoch:                       # decoded before the slices were cut again (a margin that was too small)
                    continue
                t0 = time.perf_counter()
                dec = getattr(self, "decoder", None)
                if dec is not None:
                    dec._mark("consumer: got a part")
                    if os.environ.get("SVX_TIMING"):
                        from . import sample as _sample_mod
                        _sample_mod._MARK = dec._mark
                if callable(table):                            # device engine: the QNAME ids are still to be computed
                    table = table()
                tid = unit.tid if unit is not None else int(table.tid[0])
                while want and want[0] != tid:                 # chromosomes of this rank without a record in the file (or without one in the rest of it)
                    t_ = want.pop(0)
                    self._emit_empty(t_, covered.get(t_, 0))
                if not want:
                    break
                # what the part serves: a whole chromosome from the host engine = everything of it not handed over yet
                lo = covered.get(tid, 0) if unit is None else max(unit.lo, covered.get(tid, 0))
                last = unit is None or unit.last
                hi = inf if last else unit.hi
                if table is None:                              # a slice without a record
                    self._emit_empty(tid, lo, hi)
                    self._acked = (epoch, self._acked[1] + 1 if self._acked[0] == epoch else 1)
                    covered[tid] = hi
                    if last:
                        want.pop(0)
                    continue
                with torch.cuda.stream(scan_stream):
                    if arrays is None:
                        sample = Sample.from_table(table, self.fasta, self.options.min_sv_size, self.device)
                    else:
                        sample = Sample.from_device(table, self.fasta, self.options.min_sv_size, *arrays)
                alloc = table._alloc
                if dec is not None:
                    dec._mark("consumer: scanned")
                if unit is not None and not self._slice_complete(unit, sample):
                    # the records of this slice reach farther than the margin it was cut with: it may miss records its windows'
                    # clusters count or genotype with.  Nothing of it is handed over; stage A cuts the rest of the file again.
                    self.stats["replans"] += 1
                    if dec is not None:
                        dec._mark("slice %r rejected: its windows need %d bases beyond them" % (unit, self._slice_needs(unit, sample)))
                    self._replan = (epoch, tid, lo, self._slice_needs(unit, sample))
                    self._epoch = epoch + 1
                    sample.device_buffers = None
                    if getattr(alloc, "dir", None) is not None:
                        self._slot_free(alloc.dir)
                    continue
                for name, arr in (("gaps", sample.gaps), ("gap_off", sample.gap_off), ("stats", sample.stats)):
                    alloc.put(name, arr)
                self.stats["upload_scan_s"] += time.perf_counter() - t0
                self.stats["cigar_bytes"] += int(table.cigar.nbytes)
                self.stats["records"] += len(table)
                self.stats["slices"] += 1
                meta = {"dir": alloc.dir, "arrays": dict(alloc.arrays), "references": self.references, "lengths": self.lengths,
                        "min_sv": self.options.min_sv_size, "n": len(table), "with_seq": self.with_seq, "header_text": table.header_text,
                        "stats_shape": list(np.shape(sample.stats))}
                if arrays is not None:                         # device engine: the host copy of the CIGAR words follows later
                    meta["lazy_cigar"] = int(table.cigar.size)
                    meta["spilled"] = threading.Event()
                    if hasattr(table.cigar, "attach"):            # (LazyCigar: a reader in THIS process waits for the spill)
                        table.cigar.event = meta["spilled"]
                    spill.put((table, meta["spilled"], sample))
                else:
                    # host engine: the upload served the scan and nothing else (windows of a file-driven run are never scanned
                    # again: their chromosome's Sample is final) -- the slice's device arrays go back to the allocator now, not when
                    # the whole chromosome has been voted (a chromosome's slices held their buffers until release())
                    sample.device_buffers = None
                if dec is not None:
                    dec._mark("handed over %s [%s, %s) (scan + slot writes)" % (self.references[tid], lo, hi))
                    dec.first_handover.set()                    # (the decoder holds its second launch back for this, ingest_gpu.py)
                covered[tid] = hi
                if last:
                    want.pop(0)
                self._put((self.references[tid], lo, hi, sample, meta))
                self._acked = (epoch, self._acked[1] + 1 if self._acked[0] == epoch else 1)
            while want and not self._stop:
                t_ = want.pop(0)
                self._emit_empty(t_, covered.get(t_, 0))
            if getattr(self, "decoder", None) is not None:
                self.stats["device_decoder"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in self.decoder.stats.items()}
                self.stats["device_decoder"]["trace"] = ["%.3f %s" % (t + (self.decoder._t0 - self._t0), w) for t, w in self.decoder.trace[:400]]
        except BaseException as exc:                           # noqa: BLE001 -- surfaces in the owner thread (poll / get)
            self.error = exc
            self._stop = True
            try:
                while True:                                    # let stage A run into its stop flag
                    decoded.get_nowait()
            except queue.Empty:
                pass
        finally:
            self._b_done = True                                # stage A's settle() / put() must not wait for acknowledgements that cannot come
            try:
                spill.put(None)
            except NameError:
                pass
            self.handover.put(None)

    @staticmethod
    def _slice_complete(unit, sample):
        """Does the slice hold every record its windows can touch?  A window [a, b) collects on the records that overlap it; the
        coordinates of their signatures -- and so the extents of the clusters the window reports -- stay within those records'
        [pos - read length, end + read length] (inserted / re-placed pieces), the cluster coverage (classes.py:165-170) counts the
        records overlapping those extents and the genotyper looks 1000 bases farther (genotype.py:22-26).  So the window needs every
        record overlapping [min(pos - l_seq) - 1000, max(end + l_seq) + 1000) of ITS records.  On the left the slice starts at the
        linear-index entry of ``left_edge`` (every record reaching beyond that coordinate lies behind it), on the right it ends in
        front of a record that starts at or behind its last one (the file is sorted)."""
        table = sample.table
        if len(table) == 0:
            return unit.left_edge is None and unit.to_end
        tid = int(table.tid[0])
        ref_end = table.ref_end()
        last_pos = int(table.pos[-1])
        for a, b in unit.windows:
            rows = table.fetch(tid, a, b)
            if rows.size == 0:
                lo_need, hi_need = a - 1000, b + 1000
            else:
                l_seq = table.l_seq[rows].astype(np.int64)
                lo_need = min(a, int((table.pos[rows].astype(np.int64) - l_seq).min())) - 1000
                hi_need = max(b, int((ref_end[rows] + l_seq).max())) + 1000
            if unit.left_edge is not None and unit.left_edge > lo_need:
                return False
            if not unit.to_end and last_pos < hi_need:
                return False
        return True

    @staticmethod
    def _slice_needs(unit, sample):
        """The margin (bases beyond its windows) the slice turned out to need: what it is cut again with, doubled."""
        table = sample.table
        if len(table) == 0:
            return 0
        tid, ref_end, need = int(table.tid[0]), table.ref_end(), 0
        for a, b in unit.windows:
            rows = table.fetch(tid, a, b)
            if rows.size:
                l_seq = table.l_seq[rows].astype(np.int64)
                need = max(need, a - int((table.pos[rows].astype(np.int64) - l_seq).min()), int((ref_end[rows] + l_seq).max()) - b)
        return need + 1000

    def _spill(self, jobs):
        """Stage C: host copies of the device-decoded CIGAR words (ingest_gpu.spill_cigar), after the hand-over."""
        import torch
        from .ingest_gpu import spill_cigar
        stream = streams.get("spill", self.device)
        while True:
            job = jobs.get()
            if job is None:
                return
            table, done, sample = job
            try:
                with torch.cuda.stream(stream):
                    spill_cigar(table)
                    if sample.device_buffers is not None:      # (the arrays were last used on this stream: the spill's copy)
                        for t in sample.device_buffers[:3]:
                            t.record_stream(stream)
                table._d_cigar = None
                sample.device_buffers = None                   # the device arrays of the slice: scanned, spilled, never read again
            except BaseException as exc:                       # noqa: BLE001 -- a helper that needs the words would wait for ever
                self.error = self.error or exc
            finally:
                done.set()

    def _decode(self, engine, tids, decoded):
        """Stage A: the rank's chromosomes -- whole, or in slices of collection windows (device engine with ``tasks``) -- as
        (epoch, unit or None, table or a callable finishing it, device arrays or None) into ``decoded``."""
        import logging

        def put(item):
            while not self._stop and not self._b_done:
                try:
                    decoded.put(item, timeout=0.2)
                    return True
                except queue.Full:
                    continue
            return False

        state = {"epoch": 0, "put": 0}

        def host_parts(which):                                 # BGZF inflate on host threads (libdeflate): whole chromosomes
            slot = {"alloc": None, "used": True, "arrays": {}}

            def alloc(name, dtype, n):                         # one slot per part: "tid" is the first array of a part (io.bam._table_from_handle)
                if name == "tid" and slot["used"]:            # (a part the stream skipped leaves its slot to the next one)
                    slot["alloc"], slot["used"], slot["arrays"] = self._slot_alloc(), False, {}
                arr = slot["arrays"][name] = np.empty(int(n), dtype)       # the decoder fills this process's own memory ...
                return arr
            stream = BamStream(self.bam_path, with_seq=self.with_seq, threads=self.threads, tids=which, index=self.index, alloc=alloc)
            try:
                for table in stream:
                    for name, arr in slot["arrays"].items():   # ... and the slot's files are written from it (no mapping in this process: _slot_alloc.put)
                        slot["alloc"].put(name, arr)
                    table._alloc, slot["used"] = slot["alloc"], True
                    if not put((state["epoch"], None, table, None)):
                        return False
                    state["put"] += 1
            finally:
                stream.close()
            return True

        def replan_wanted():
            rp = self._replan
            return rp if rp is not None and rp[0] == state["epoch"] else None

        def settle():
            """Wait until stage B has looked at everything put in this epoch: the LAST slice may be the one it rejects."""
            while not self._stop and not self._b_done and replan_wanted() is None and self._acked != (state["epoch"], state["put"]) and state["put"]:
                time.sleep(0.0005)
            return replan_wanted()

        try:
            if engine == "gpu":                                # BGZF inflate + record packing on the device
                from .ingest_gpu import DeviceDecoder, DeviceIngestError
                # pread threads of the device engine: copies out of the page cache, ~2 GB/s each (SVX_READ_THREADS overrides)
                n_read = int(os.environ.get("SVX_READ_THREADS", "0")) or min(8, max(1, self.threads))
                dec = self.decoder = DeviceDecoder(self.bam_path, self.index, self.references, self.lengths, self.header_text, self.device,
                                                   threads=n_read, alloc_for=self._slot_alloc)
                if not dec.usable(tids):
                    host_parts(tids)
                else:
                    windows_of = None
                    if self.tasks is not None and not getattr(self.options, "contig", False):
                        def windows_of(t):
                            return self.tasks.get(self.references[t])
                    # (SVX_SLICE_MARGIN: tests -- a margin that is too small must be noticed and the slices cut again)
                    margin = (int(os.environ.get("SVX_SLICE_MARGIN", "0")) or dec.estimate_reach(tids)) if windows_of is not None else 0
                    units = dec.plan_units(tids, windows_of, margin)
                    refusals = 0
                    while units and not self._stop:
                        done, outcome, exc_ = 0, "end", None
                        try:
                            for part in dec.units_pipelined(units):
                                if replan_wanted() is not None:
                                    outcome = "replan"
                                    break
                                if not put((state["epoch"],) + tuple(part)):
                                    return
                                state["put"] += 1
                                done += 1
                        except DeviceIngestError as exc:
                            outcome, exc_ = "error", exc
                        if outcome != "replan" and settle() is not None:
                            outcome = "replan"                   # (an error behind a rejected slice: cut again first, the error will come back)
                        if outcome == "replan":
                            # a slice's records reach farther than the margin guessed from the file's first reads (ONT: a log-normal
                            # tail): everything from that slice on is cut again with twice what it needed
                            _ep, tid_, coord, needed = replan_wanted()
                            margin = (max(2 * needed, 2 * margin) + 16383) >> 14 << 14
                            state["epoch"], state["put"] = state["epoch"] + 1, 0
                            logging.info("device ingestion: slices of %s cut again from %d on with a margin of %d bases", self.references[tid_], coord, margin)
                            units = dec.plan_units(tids, windows_of, margin, resume=(tid_, coord))
                            continue
                        if outcome == "end":
                            break
                        # an index that does not fit, a corrupt block: the host reader takes those references -- whole; what the
                        # device has handed over of them stays, the host's table serves the rest -- and the device engine goes on
                        # behind them -- three times; then the host reader takes the rest of the file
                        refusals += 1
                        rest = units[done:]
                        named = [i for i, u in enumerate(rest) if u.tid in (getattr(exc_, "tids", None) or [])]
                        hi = (max(named) + 1) if named else 1
                        bad = []
                        for u in (rest if refusals >= 3 else rest[:hi]):
                            if u.tid not in bad:
                                bad.append(u.tid)
                        logging.warning("device ingestion failed at reference %s (%s): decoding %s on the host",
                                        ", ".join(self.references[t] for t in bad[-max(1, len({rest[i].tid for i in named})):]), exc_,
                                        "the rest of the file" if refusals >= 3 else "%d reference(s)" % len(bad))
                        if not host_parts(bad):
                            return
                        units = [u for u in rest if u.tid not in bad]
                    settle()
            else:
                host_parts(tids)
            put(None)
        except BaseException as exc:                           # noqa: BLE001
            put(exc)

    def _emit_empty(self, tid, lo=0, hi=float("inf")):
        """Windows [lo, hi) of a chromosome without records there: they still exist as tasks (SVision:172-201) and yield nothing."""
        meta = {"dir": None, "references": self.references, "lengths": self.lengths, "min_sv": self.options.min_sv_size, "n": 0,
                "with_seq": self.with_seq, "header_text": ""}
        self._put((self.references[tid], lo, hi, empty_sample(self.references, self.lengths, self.fasta, self.options.min_sv_size), meta))

    def _put(self, item):
        t0 = time.perf_counter()
        self.handover.put(item)
        self.stats["handover_wait_s"] += time.perf_counter() - t0

    # ---- owner thread ----------------------------------------------------------------------------------------------
    def poll(self, block=False):
        """Move the chromosomes / slices the feeder has finished into ``samples`` (and onto the ``fresh`` list of what the
        helpers have not been told yet); ``block``: wait until at least one arrives or the stream ends.  -> whether any arrived."""
        got = False
        while not self.finished:
            try:
                item = self.handover.get(block=block and not got, timeout=0.25 if block and not got else None)
            except queue.Empty:
                if block and not got:
                    continue
                break
            if item is None:
                self.finished = True
                if self.error is not None:
                    raise self.error
                break
            chrom, lo, hi, sample, meta = item
            key = next(_KEYS)                                  # unique in the process: a helper may still hold a key of an earlier feed
            now = time.perf_counter() - self._t0
            if self.stats["first_ready_s"] is None:
                self.stats["first_ready_s"] = now
            self.stats["last_ready_s"] = now
            self.samples.setdefault(chrom, []).append([lo, hi, key, sample, meta])
            self.fresh.append((key, chrom, meta))
            got = True
        return got

    def _find(self, chrom, start):
        for ent in self.samples.get(chrom, ()):
            if start is None or ent[0] <= start < ent[1]:
                return ent
        return None

    def get(self, chrom, block=True, start=None):
        """-> (key, Sample) of the part of a chromosome that serves the collection window starting at ``start`` (None: its
        first part -- a whole chromosome is one part), or (None, None) when it is not ready and ``block`` is False."""
        while True:
            ent = self._find(chrom, start)
            if ent is not None:
                return ent[2], ent[3]
            if self.finished:
                raise KeyError("chromosome %s%s is not part of this feed" % (chrom, "" if start is None else " at %s" % start))
            self.poll(block=block)
            if not block and self._find(chrom, start) is None:
                return None, None

    def keys_of(self, chrom):
        return [ent[2] for ent in self.samples[chrom]]

    def take_fresh(self):
        """(key, chrom, meta) of the parts that arrived since the last call: what the helpers must be sent."""
        out, self.fresh = self.fresh, []
        # (the meta of a chromosome goes through a pipe: without the owner-side event)
        return [(k, c, None if m is None else {a: b for a, b in m.items() if a != "spilled"}) for k, c, m in out]

    def release(self, chrom):
        """The chromosome is done (its windows voted and stitched): free its device buffers and shared memory."""
        for ent in self.samples.get(chrom, ()):
            _lo, _hi, _key, sample, meta = ent
            if sample is None:
                continue
            sample.device_buffers = None
            ent[3], ent[4] = None, None
            if meta is not None and meta["dir"] is not None:
                if meta.get("spilled") is not None:
                    meta["spilled"].wait(timeout=120)          # the slot must not be reused under a running spill
                self._slot_free(meta["dir"])

    def close(self):
        self._stop = True
        try:
            while self.thread.is_alive():
                try:
                    self.handover.get_nowait()
                except queue.Empty:
                    time.sleep(0.005)
        finally:
            shutil.rmtree(self.root, ignore_errors=True)


def load_shared_sample(meta, fasta):
    """Helper-process side: the chromosome a ChromosomeFeed announced, mapped from shared memory (copy-on-write)."""
    d = meta["dir"]
    if d is None:
        return empty_sample(meta["references"], meta["lengths"], fasta, meta["min_sv"], meta["header_text"])
    arrays = meta["arrays"]

    def arr(name, dtype):
        if name not in arrays or arrays[name][1] == 0:
            return np.empty(0, dtype)
        return np.memmap(os.path.join(d, name + ".bin"), dtype=np.dtype(dtype), mode="c", shape=(arrays[name][1],))
    names_blob = arr("names", np.uint8)
    names = bytes(names_blob).decode().split("\n")[:-1] if names_blob.size else []
    cig_off = arr("cig_off", np.int64)
    if cig_off.size == 0:
        cig_off = np.zeros(1, np.int64)
    seq_packed = arr("seq_packed", np.uint8) if meta["with_seq"] else None
    seq_off = arr("seq_off", np.int64) if meta["with_seq"] else None
    table = AlignmentTable(meta["references"], meta["lengths"], arr("tid", np.int32), arr("pos", np.int32), arr("flag", np.uint16),
                           arr("mapq", np.uint8), arr("l_seq", np.int32), arr("name_id", np.int32), names, arr("cigar", np.uint32), cig_off,
                           meta["header_text"], seq_packed, seq_off)
    if meta.get("lazy_cigar") is not None:                     # device engine: the words arrive in the slot a little later
        from .ingest_gpu import LazyCigar
        table.cigar = LazyCigar(meta["lazy_cigar"], os.path.join(d, "cigar.bin"), os.path.join(d, "cigar.ready"))
    from . import kernels
    gaps = arr("gaps", kernels.GAP_DTYPE)
    stats = np.asarray(arr("stats", np.int32)).reshape(meta["stats_shape"])
    return Sample.with_scan(table, fasta, meta["min_sv"], (gaps, arr("gap_off", np.int64), stats))
