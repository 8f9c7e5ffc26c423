"""A sequenced sample resident on the device: the decoded BAM as a structure of arrays,
its packed CIGAR words in HBM, and the result of the device CIGAR scan.

This is the object the reference's ``sample_path`` argument stands for
(run_collection.py:15-26 re-opens the BAM in every worker; here the file is decoded
once per process, uploaded once, and every alignment's long gaps / reference span /
clip lengths come from one ``svx_cigar_scan`` launch sequence).
"""
import numpy as np

from .io.bam import Fasta, read_bam

_CACHE = {}
_MARK = None                                                 # (SVX_TIMING: ingest._run points this at the decoder's trace)
_PINNED = []                                                 # pinned int32 scratch buffers of the scans' read-backs (grow-only, reused)


def _pinned_scratch(words):
    import torch
    for i, buf in enumerate(_PINNED):
        if buf.numel() >= words:
            return _PINNED.pop(i)
    return torch.empty(max(int(words), 1 << 20), dtype=torch.int32, pin_memory=True)


class Sample:
    def __init__(self, table, fasta, gaps, gap_off, stats, min_sv, device_buffers=None):
        self.table = table
        self.fasta = fasta
        self.gaps = gaps
        self.gap_off = np.asarray(gap_off).astype(np.int64)
        self.min_sv = min_sv
        self.device_buffers = device_buffers
        self.last_window_scan = None
        self.stats = np.asarray(stats)
        table.attach_scan(self.stats)

    # -- construction -----------------------------------------------------------------
    @classmethod
    def from_table(cls, table, fasta, min_sv, device="cuda"):
        """Upload the packed CIGARs and run the device scan (the product path of a host-decoded table).  The upload goes
        through pinned memory on the caller's current stream, like the read-back (:meth:`_scan_resident`): a copy from or to
        pageable memory is staged by the runtime on an internal queue of normal priority, where it waits behind the CNN's
        queued launches (tens of ms per chromosome in a file-driven run)."""
        import torch
        dev = torch.device(device)
        cigar = np.ascontiguousarray(table.cigar).view(np.int32).reshape(-1)
        cig_off = np.ascontiguousarray(table.cig_off, np.int64)
        pos = np.ascontiguousarray(table.pos, np.int32)
        n, words = int(pos.size), int(cigar.size)
        pad = (-words) % 4                                       # (svx_cigar_scan reads 16-byte quads)
        stage = _pinned_scratch(words + pad + 2 * (n + 1) + 2 + n)
        h = stage.numpy()
        h[:words] = cigar
        h[words:words + pad] = 0
        at_off = (words + pad + 1) // 2 * 2                     # int64 view: an even word index
        h[at_off:at_off + 2 * (n + 1)].view(np.int64)[:] = cig_off
        at_pos = at_off + 2 * (n + 1)
        h[at_pos:at_pos + n] = pos
        d_all = stage[:at_pos + n].to(dev, non_blocking=True)
        d_cigar = d_all[:max(words, 1)] if words else torch.zeros(4, dtype=torch.int32, device=dev)
        d_off = d_all[at_off:at_off + 2 * (n + 1)].view(torch.int64)
        d_pos = d_all[at_pos:at_pos + n]
        out = cls._scan_resident(table, fasta, min_sv, d_cigar, d_off, d_pos)
        _PINNED.append(stage)                                    # (the scan's event is through: the upload has been consumed)
        return out

    @classmethod
    def from_device(cls, table, fasta, min_sv, d_cigar, d_off, d_pos):
        """The packed arrays are already in HBM (device-side ingestion, svision_amd/ingest_gpu.py): scan them in place."""
        import torch
        # the arrays were produced (and allocated) on the decoder's stream, whose event the caller has waited for; they are
        # used on the caller's current stream from here on: the caching allocator must not hand their blocks out behind that
        # stream's back when they are freed
        for t in (d_cigar, d_off, d_pos):
            t.record_stream(torch.cuda.current_stream(t.device))
        return cls._scan_resident(table, fasta, min_sv, d_cigar, d_off, d_pos)

    @classmethod
    def _scan_resident(cls, table, fasta, min_sv, d_cigar, d_off, d_pos):
        import torch
        from . import kernels
        # The result comes back through PINNED memory, copies and event on the caller's (high-priority) stream.  A read-back
        # into pageable memory (.item(), .cpu()) is staged by the runtime through a copy kernel on an internal queue of
        # normal priority: behind the CNN's queued launches and a tokens launch that owns every CU it took 40-90 ms instead
        # of 1-3, several times per job -- the hand-over of a whole group of chromosomes waited for another group's inflate.
        n = int(d_pos.numel())
        cap = max(1024, n // 2)
        while True:
            res = kernels.cigar_scan(d_cigar, d_off, d_pos, min_sv, gaps_cap=cap)
            if _MARK: _MARK("from_device: scan enqueued")
            need = (n + 1) + cap * 6 + n * 4
            host = _pinned_scratch(need)
            host[:n + 1].copy_(res.gap_off, non_blocking=True)
            host[n + 1:n + 1 + cap * 6].copy_(res.gaps[:cap * 6], non_blocking=True)
            host[n + 1 + cap * 6:need].copy_(res.stats.view(-1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            if _MARK: _MARK("from_device: copies enqueued")
            ev.synchronize()
            if _MARK: _MARK("from_device: event done")
            h = host.numpy()
            total = kernels.check_scan_total(int(h[:n + 1].view(np.uint32)[n]))
            if total <= cap:
                break
            _PINNED.append(host)
            cap = total                                      # gap_off[n] holds the full count: once more with the exact capacity
        gap_off = h[:n + 1].view(np.uint32).copy()
        gaps = h[n + 1:n + 1 + total * 6].copy().view(kernels.GAP_DTYPE) if total else np.empty(0, kernels.GAP_DTYPE)
        stats = h[n + 1 + cap * 6:need].reshape(n, 4).copy()
        _PINNED.append(host)
        from .segmentplot import run_hash_lineplot
        run_hash_lineplot.DEVICE = d_cigar.device            # this process owns a GPU: --hash re-alignment seeds on the device
        return cls(table, fasta, gaps, gap_off, stats, min_sv, device_buffers=(d_cigar, d_off, d_pos, res))

    @classmethod
    def with_scan(cls, table, fasta, min_sv, scan):
        """Attach an externally computed scan (tests inject the oracle's here)."""
        gaps, gap_off, stats = scan
        return cls(table, fasta, gaps, gap_off, stats, min_sv)

    @classmethod
    def open(cls, bam_path, genome_path, min_sv, device="cuda", with_seq=False):
        key = (bam_path, genome_path, int(min_sv), str(device), bool(with_seq))
        if key not in _CACHE:
            _CACHE[key] = cls.from_table(read_bam(bam_path, with_seq=with_seq), Fasta(genome_path), min_sv, device)
        return _CACHE[key]

    def rescan_window(self, chrom, start, end):
        """Re-run the device scan for the block of rows whose start lies in [start, end) of
        ``chrom`` and refresh the host copies (streaming use: each window's rows are scanned
        when the window is processed).  Rows form a partition of the table over windows."""
        return self.finish_rescan(self.rescan_window_async(chrom, start, end))

    def rescan_window_async(self, chrom, start, end):
        """Enqueue the scan of a window's rows and the copy of its result to pinned host memory on the (high priority)
        scan stream; returns a handle for :meth:`finish_rescan`, or None for a window without rows.  Issued a few windows
        ahead by the pipeline: a blocking read-back right after the launch waited 12 ms per window behind the queued CNN
        batches (a tiny kernel arriving behind ~40 queued graph replays is dispatched late, whatever its stream), which
        made the GPU-owning thread -- not the device -- the limit of the whole pipeline."""
        import torch
        from . import kernels
        table = self.table
        b = table._bounds()
        tid = table.get_tid(chrom)
        lo = int(b[tid]) + int(np.searchsorted(table.pos[b[tid]:b[tid + 1]], start, side="left"))
        hi = int(b[tid]) + int(np.searchsorted(table.pos[b[tid]:b[tid + 1]], end, side="left"))
        if hi <= lo:
            return None
        d_cigar, d_off, d_pos, _ = self.device_buffers
        n = hi - lo
        cap = int(self.gap_off[hi] - self.gap_off[lo])
        capd = max(cap, 16)
        if getattr(self, "_scan_stream", None) is None:
            from . import streams
            self._scan_stream = streams.get("scan", d_cigar.device)   # ONE high-priority scan stream per process (streams.py)
            self._scan_pinned = []
        need = (n + 1) + capd * 6 + n * 4
        host = None
        for i, buf in enumerate(self._scan_pinned):
            if buf.numel() >= need:
                host = self._scan_pinned.pop(i)
                break
        if host is None:
            host = torch.empty(max(need, 1 << 20), dtype=torch.int32, pin_memory=True)
        with torch.cuda.stream(self._scan_stream):
            res = kernels.cigar_scan(d_cigar, d_off[lo:hi + 1], d_pos[lo:hi], self.min_sv, gaps_cap=capd,
                                     n_words=int(table.cig_off[hi]), span_words=int(table.cig_off[hi] - table.cig_off[lo]))
            host[:n + 1].copy_(res.gap_off, non_blocking=True)
            host[n + 1:n + 1 + capd * 6].copy_(res.gaps[:capd * 6], non_blocking=True)
            host[n + 1 + capd * 6:need].copy_(res.stats.view(-1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        import time
        return (lo, hi, cap, capd, host, ev, res, time.perf_counter())

    def finish_rescan(self, handle):
        """Wait for an enqueued window scan, check it against the resident scan and install it; -> number of rows."""
        from . import kernels
        if handle is None:
            self.last_window_scan = None
            return 0
        lo, hi, cap, capd, host, ev, _res = handle[:7]
        ev.synchronize()
        n = hi - lo
        h = host.numpy()
        gap_off = h[:n + 1].view(np.uint32).copy()
        gaps = h[n + 1:n + 1 + cap * 6].copy().view(kernels.GAP_DTYPE) if cap else np.empty(0, kernels.GAP_DTYPE)
        stats = h[n + 1 + capd * 6:n + 1 + capd * 6 + n * 4].reshape(n, 4).copy()
        self._scan_pinned.append(host)
        if int(gap_off[-1]) != cap or not np.array_equal(gap_off.astype(np.int64) + self.gap_off[lo], self.gap_off[lo:hi + 1]):
            raise RuntimeError("window rescan disagrees with the resident scan")
        gaps["aln"] += lo
        self.apply_window_scan(lo, hi, gaps, stats)
        self.last_window_scan = (lo, hi, gaps, stats)           # what a helper process needs to follow (apply_window_scan)
        return hi - lo

    def apply_window_scan(self, lo, hi, gaps, stats):
        """Install the scan of the rows [lo, hi) (gaps with absolute alignment indices, stats [hi-lo, 4])."""
        table = self.table
        self.gaps[self.gap_off[lo]:self.gap_off[hi]] = gaps
        changed = not np.array_equal(table.ref_span[lo:hi], stats[:, 0])
        table.ref_span[lo:hi], table.lead_clip[lo:hi], table.trail_clip[lo:hi] = stats[:, 0], stats[:, 1], stats[:, 2]
        if changed:
            table._ref_end = None                               # reference ends (and their sorted copies) are derived from ref_span
            table._max_span = None

    def reach(self):
        """How far anything a collection window does can lie outside the window: a cluster reported by a window is built from
        records that overlap it, its signatures' coordinates stay within the reference span of those records plus one read
        length (inserted / re-placed pieces), and the genotyper looks 1000 bases farther (genotype.py:22-26).  The longest
        span + the longest read + 1000 of the whole table: an upper bound for every window of it."""
        m = getattr(self, "_reach", None)
        if m is None:
            t = self.table
            m = self._reach = (int(t.ref_span.max()) + int(t.l_seq.max()) + 1000) if len(t) else 0
        return m

    # -- accessors used by the collection step ----------------------------------------
    def gaps_of(self, aln):
        return self.gaps[self.gap_off[aln]:self.gap_off[aln + 1]]

    def chrom_of(self, tid):
        return self.table.references[tid]

    def fetch_ref(self, chrom, start, end):
        """pysam FastaFile.fetch semantics (analyze_reads.py:42-46); bytes for cheap indexing."""
        if start < 0 or end < start:
            raise ValueError("invalid coordinates: start (%d) > stop (%d)" % (start, end))
        return self.fasta.fetch_bytes(chrom, start, end)

    def fetch_ref_view(self, chrom, start, end):
        """:meth:`fetch_ref` without copying the span (the left-alignment of analyze_gap reads a few bases of tens of kb)."""
        if start < 0 or end < start:
            raise ValueError("invalid coordinates: start (%d) > stop (%d)" % (start, end))
        return self.fasta.fetch_view(chrom, start, end)

    def fetch_ref_str(self, chrom, start, end):
        """The same as text (the --hash re-aligner works on str)."""
        return self.fetch_ref(chrom, start, end).decode()


def register(path, sample):
    """Make ``run_detect(options, path, ...)`` resolve to an already prepared Sample."""
    _CACHE[("registered", path)] = sample


def resolve(sample_or_path, options):
    if isinstance(sample_or_path, Sample):
        return sample_or_path
    reg = _CACHE.get(("registered", sample_or_path))
    if reg is not None:
        return reg
    return Sample.open(sample_or_path, options.genome, options.min_sv_size, with_seq=bool(options.hash or getattr(options, "graph", False)))
