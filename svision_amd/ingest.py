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
import os
import queue
import shutil
import tempfile
import threading
import time

import numpy as np

from .io.bam import AlignmentTable, BamStream
from .sample import Sample


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
    """Inflate threads of one rank: the node's usable CPUs divided by its ranks, at most 128.  (The helpers wait for the
    decoder whenever it is the limit, so nothing is subtracted for them: measured on a 16-CPU quota, 24 threads 1.12 s,
    16 threads 1.18 s, 11 threads 1.63 s for the same 9.3 GB of inflated data.)"""
    usable, _visible = effective_cpus()
    return max(2, min(128, usable // max(ranks_on_node, 1)))


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

    def get(self, chrom, block=True):
        return None, self.sample

    def key_of(self, chrom):
        return None

    def release(self, chrom):
        pass

    def close(self):
        pass


class ChromosomeFeed:
    """See the module docstring.  ``chroms``: this rank's chromosomes in task order; ``references`` / ``lengths``: the
    BAM header's dictionary.  ``stats`` afterwards: seconds the feeder thread spent waiting for the decoder, uploading +
    scanning, and blocked on a full hand-over queue; bytes of packed CIGAR uploaded."""

    def __init__(self, bam_path, fasta, options, chroms, references, lengths, device="cuda", index=None, threads=0, depth=2,
                 engine=None, header_text=""):
        self.bam_path, self.fasta, self.options = bam_path, fasta, options
        self.header_text = header_text
        # where the BGZF blocks are inflated: "cpu" = libdeflate on host threads (io.bam.BamStream), "gpu" = on the device
        # (ingest_gpu.DeviceDecoder); default: the device when the host is short of CPU time and the file has a linear index
        self.engine = engine or os.environ.get("SVX_INGEST", "auto")
        self.references, self.lengths = list(references), list(lengths)
        self.chroms = list(chroms)
        self.device, self.index, self.threads = device, index, threads
        self.with_seq = bool(options.hash or getattr(options, "graph", False))
        shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        self.root = tempfile.mkdtemp(prefix="svx_feed_", dir=shm)
        self.handover = queue.Queue(maxsize=depth)           # feeder thread -> owner thread
        self.samples = {}                                      # chrom -> (key, Sample, meta) once the owner has seen it
        self.fresh = []                                        # (key, meta) not yet announced to the helpers
        self.error = None
        self.finished = False
        self.stats = {"engine": None, "decode_wait_s": 0.0, "upload_scan_s": 0.0, "handover_wait_s": 0.0, "cigar_bytes": 0, "records": 0,
                      "first_ready_s": None, "last_ready_s": None}
        self._t0 = time.perf_counter()
        self._stop = False
        self.thread = threading.Thread(target=self._run, name="svx-feed", daemon=True)
        self.thread.start()

    # ---- feeder thread ---------------------------------------------------------------------------------------------
    def _alloc_in(self, d):
        def alloc(name, dtype, n):
            if n == 0:
                return np.empty(0, dtype)
            return np.lib.format.open_memmap(os.path.join(d, name + ".npy"), mode="w+", dtype=dtype, shape=(n,))
        alloc.dir = d
        return alloc

    def _run(self):
        import torch
        try:
            tids = [self.references.index(c) for c in self.chroms]
            want = list(tids)
            seq = [0]

            def next_dir():
                d = os.path.join(self.root, "c%d" % seq[0])
                seq[0] += 1
                os.makedirs(d)
                self._dir = d
                return d

            class Alloc:                                       # one directory per part, created when the part arrives
                def __init__(self, feed):
                    self.feed, self.fn = feed, None

                def __call__(self, name, dtype, n):
                    if name == "tid":                          # first array of a part (io.bam._table_from_handle)
                        self.fn = self.feed._alloc_in(next_dir())
                    return self.fn(name, dtype, n)

            ingest_stream = torch.cuda.Stream(device=self.device, priority=-1) if torch.cuda.is_available() else None
            if ingest_stream is None:
                raise RuntimeError("ChromosomeFeed needs the GPU (svx_cigar_scan); there is no CPU fallback")

            def host_parts(which):                             # BGZF inflate on host threads
                stream = BamStream(self.bam_path, with_seq=self.with_seq, threads=self.threads, tids=which, index=self.index, alloc=Alloc(self))
                try:
                    for table in stream:
                        table._shm_dir = self._dir
                        yield table, None
                finally:
                    stream.close()

            def device_parts(which):                           # BGZF inflate + record packing on the device
                from .ingest_gpu import DeviceDecoder, DeviceIngestError

                def alloc_for():
                    return self._alloc_in(next_dir())
                dec = self.decoder = DeviceDecoder(self.bam_path, self.index, self.references, self.lengths, self.header_text, self.device,
                                                   threads=min(8, max(1, self.threads)), alloc_for=alloc_for)
                if not dec.usable(which):
                    yield from host_parts(which)
                    return
                for group in dec.groups(which):
                    try:
                        with torch.cuda.stream(ingest_stream):
                            parts = dec.decode_group(group)
                    except DeviceIngestError as exc:           # CG-tag CIGARs, an index that does not fit: the host reader takes the group
                        import logging
                        logging.warning("device ingestion of references %s failed (%s): decoding them on the host", group, exc)
                        yield from host_parts(group)
                        continue
                    for table, arrays in parts:
                        yield table, arrays

            engine = self.engine
            if engine == "auto":
                usable, _visible = effective_cpus()
                engine = "gpu" if (self.index is not None and not self.with_seq and usable < 48) else "cpu"
            if engine == "gpu" and (self.index is None or self.with_seq):
                engine = "cpu"
            self.stats["engine"] = engine
            it = device_parts(tids) if engine == "gpu" else host_parts(tids)
            while not self._stop:
                t0 = time.perf_counter()
                item = next(it, None)
                self.stats["decode_wait_s"] += time.perf_counter() - t0
                if item is None:
                    break
                table, arrays = item
                tid = int(table.tid[0])
                while want and want[0] != tid:                 # chromosomes of this rank without a record in the file
                    self._emit_empty(want.pop(0))
                if not want:
                    break
                want.pop(0)
                t0 = time.perf_counter()
                with torch.cuda.stream(ingest_stream):
                    if arrays is None:
                        sample = Sample.from_table(table, self.fasta, self.options.min_sv_size, self.device)
                    else:
                        sample = Sample.from_device(table, self.fasta, self.options.min_sv_size, *arrays)
                d = table._shm_dir
                np.save(os.path.join(d, "gaps.npy"), sample.gaps)
                np.save(os.path.join(d, "gap_off.npy"), sample.gap_off)
                np.save(os.path.join(d, "stats.npy"), sample.stats)
                self.stats["upload_scan_s"] += time.perf_counter() - t0
                self.stats["cigar_bytes"] += int(table.cigar.nbytes)
                self.stats["records"] += len(table)
                meta = {"dir": d, "references": self.references, "lengths": self.lengths, "min_sv": self.options.min_sv_size,
                        "n": len(table), "with_seq": self.with_seq, "header_text": table.header_text}
                self._put((self.references[tid], sample, meta))
            while want and not self._stop:
                self._emit_empty(want.pop(0))
            if getattr(self, "decoder", None) is not None:
                self.stats["device_decoder"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in self.decoder.stats.items()}
        except BaseException as exc:                           # noqa: BLE001 -- surfaces in the owner thread (poll / get)
            self.error = exc
        finally:
            self.handover.put(None)

    def _emit_empty(self, tid):
        meta = {"dir": None, "references": self.references, "lengths": self.lengths, "min_sv": self.options.min_sv_size, "n": 0,
                "with_seq": self.with_seq, "header_text": ""}
        self._put((self.references[tid], empty_sample(self.references, self.lengths, self.fasta, self.options.min_sv_size), meta))

    def _put(self, item):
        t0 = time.perf_counter()
        self.handover.put(item)
        self.stats["handover_wait_s"] += time.perf_counter() - t0

    # ---- owner thread ----------------------------------------------------------------------------------------------
    def poll(self, block=False):
        """Move the chromosomes the feeder has finished into ``samples`` (and onto the ``fresh`` list of what the helpers
        have not been told yet); ``block``: wait until at least one arrives or the stream ends.  -> whether any arrived."""
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
            chrom, sample, meta = item
            key = len(self.samples)
            now = time.perf_counter() - self._t0
            if self.stats["first_ready_s"] is None:
                self.stats["first_ready_s"] = now
            self.stats["last_ready_s"] = now
            self.samples[chrom] = (key, sample, meta)
            self.fresh.append((key, chrom, meta))
            got = True
        return got

    def get(self, chrom, block=True):
        """-> (key, Sample) of a chromosome, or (None, None) when it is not ready and ``block`` is False."""
        while chrom not in self.samples:
            if self.finished:
                raise KeyError("chromosome %s is not part of this feed" % chrom)
            self.poll(block=block)
            if not block and chrom not in self.samples:
                return None, None
        key, sample, _meta = self.samples[chrom]
        return key, sample

    def key_of(self, chrom):
        return self.samples[chrom][0]

    def take_fresh(self):
        """(key, chrom, meta) of the chromosomes that arrived since the last call: what the helpers must be sent."""
        out, self.fresh = self.fresh, []
        return out

    def release(self, chrom):
        """The chromosome is done (its windows voted and stitched): free its device buffers and shared memory."""
        key, sample, meta = self.samples[chrom]
        if sample is None:
            return
        sample.device_buffers = None
        self.samples[chrom] = (key, None, None)
        if meta is not None and meta["dir"] is not None:
            shutil.rmtree(meta["dir"], ignore_errors=True)

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

    def arr(name, dtype):
        path = os.path.join(d, name + ".npy")
        return np.load(path, mmap_mode="c") if os.path.exists(path) else np.empty(0, dtype)
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
    gaps, gap_off, stats = (np.load(os.path.join(d, k + ".npy"), mmap_mode="c") for k in ("gaps", "gap_off", "stats"))
    return Sample.with_scan(table, fasta, meta["min_sv"], (gaps, gap_off, stats))
