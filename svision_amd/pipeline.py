"""In-memory streaming form of the hot path: one collection window at a time, no
intermediate files; the host stages run in helper processes while the GPU-owning process
only feeds the device.

Same stages and the same (parity-tested) functions as the file-based driver (cli.py):

  device  svx_cigar_scan over the window's alignment block          (collection, analyze_reads.py:828-853)
  host    reads -> segments -> signatures -> clusters -> pair lines   (run_collection.py:15-47)
  device  encode + AlexNet (svx_encode_conv1 ...), batches of `batch_size` images  (create_batch.py:88, predict.py:206-210)
  host    per-site vote -> VCF body lines + scores                    (predict.py:213-300, output.py:469)

The reference fans windows out to a ``multiprocessing.Pool`` (SVision:261-281); here the pool
workers keep only the Python glue (collection + vote) and hand packed int32 records to the one
process that owns the GPU.  The device stage replays a captured HIP graph per batch (rasterise
-> AlexNet -> pack) on a few streams, so the owner process spends microseconds per batch.
"""
import collections
import io
import os
import multiprocessing as mp
import multiprocessing.connection as mpc

import logging

import numpy as np
import torch

from . import kernels
from .collection.output_clusters import collect_pair_lines, iter_pair_lines
from .collection.run_collection import detect_window
from .network.create_batch import PAD_DATA, parse_data_fields
from .network.predict import Predict, SiteVoter

_PAD_REC = parse_data_fields(PAD_DATA.split("_"))


class WindowResult:
    __slots__ = ("chrom", "start", "end", "lines", "n_images", "packed", "vcf", "scores", "n_sites", "n_records",
                 "records", "done_event", "t_device", "wid", "tsv", "head", "tail", "edges", "sample", "classes", "probs")


def _collect_lines(sample, options, chrom, start, end):
    """The TSV lines of one window.  As in the reference's run_detect (run_collection.py:15-47) a window whose
    collection raises -- e.g. pysam's ValueError for a reference fetch that starts before 0 -- contributes nothing:
    the error is turned into a message there (and dropped by the driver, SVision:273); here it is logged."""
    try:
        _sigs, clusters = detect_window(options, sample, chrom, start, end)
        return collect_pair_lines(clusters, options)
    except Exception as exc:                                  # noqa: BLE001 -- the reference catches everything
        logging.error("[ERROR]: %s. window %s:%s-%s skipped (reference behaviour)", exc, chrom, start, end)
        return []


def _collect_parts(sample, options, chrom, start, end, emit, granule=256):
    """:func:`_collect_lines` for the streaming pipeline: the window's lines are handed to ``emit`` in parts of at least
    ``granule`` lines (whole clusters) while the later clusters are still being worked on.  -> (all lines, ok); a window
    whose collection raises anywhere contributes nothing (ok False: parts already handed on are to be dropped)."""
    lines, sent = [], 0
    try:
        _sigs, clusters = detect_window(options, sample, chrom, start, end)
        for part in iter_pair_lines(clusters, options):
            lines.extend(part)
            if len(lines) - sent >= granule:
                emit(lines[sent:])
                sent = len(lines)
        if len(lines) > sent:
            emit(lines[sent:])
        return lines, True
    except Exception as exc:                                  # noqa: BLE001 -- the reference catches everything
        logging.error("[ERROR]: %s. window %s:%s-%s skipped (reference behaviour)", exc, chrom, start, end)
        return [], False


def edge_margin(sample):
    """A site farther than this from a window boundary cannot be collected by the neighbouring window as well: a
    cluster reported by a window is built from records that overlap that window, and a signature's coordinates stay
    within the reference span of its records plus one read length (inserted / re-placed pieces)."""
    return sample.reach()


class WindowVote:
    """Vote of ONE window [start, end), fed in TSV order as the predictions arrive (``feed``) and closed by ``finish`` ->
    (VCF text, score text of its interior sites, n_sites, head, tail).  The first / last site is returned unwritten
    (predict.SiteVoter(hold_edges=True)) when it lies within :func:`edge_margin` of the window's start / end and a
    neighbouring window exists: such a site can span the boundary, and the chromosome's
    :class:`~svision_amd.network.predict.ChromosomeVote` writes it once, as the reference's vote over the concatenated
    TSV does (predict.py:235-247)."""

    def __init__(self, sample, options, chrom, lines, start=None, end=None):
        self.lines, self.fed = lines, 0
        self.vcf, self.score = io.StringIO(), io.StringIO()
        self.voter = None
        if lines:
            hold = None
            if start is not None:
                m = edge_margin(sample)
                clen = sample.table.lengths[sample.table.get_tid(chrom)]
                hold = (start + m if start > 0 else float("-inf"), end - m if end < clen else float("inf"))
            self.voter = SiteVoter(Predict(chrom, None), self.vcf, self.score, options, sample, hold_edges=True, hold_range=hold)

    def feed(self, classes, probs):
        """Predictions of the next ``len(classes)`` lines."""
        k = len(classes)
        if k:
            self.voter.feed_batch([ln.label() for ln in self.lines[self.fed:self.fed + k]], classes, probs)
            self.fed += k

    def finish(self):
        n_sites, head, tail = 0, None, None
        if self.voter is not None:
            if self.fed != len(self.lines):
                raise RuntimeError("window vote closed after %d of %d predictions" % (self.fed, len(self.lines)))
            self.voter.finish()
            head, tail = self.voter.head, self.voter.tail
            # candidate sites = distinct region keys of the segment TSV, whatever the CNN says
            n_sites = len({ln.region for ln in self.lines})
        return self.vcf.getvalue(), self.score.getvalue(), n_sites, head, tail


def _vote(sample, options, chrom, lines, classes, probs, start=None, end=None):
    """:class:`WindowVote` in one go."""
    vote = WindowVote(sample, options, chrom, lines, start, end)
    if lines:
        vote.feed(classes[:len(lines)], probs[:len(lines)])
    return vote.finish()


def _edge_regions(lines):
    """(region of the window's first TSV line, of its last one) -- what :func:`distinct_sites` compares across a boundary."""
    return (lines[0].region, lines[-1].region) if lines else (None, None)


def distinct_sites(results):
    """Candidate sites of WindowResults in task order = distinct region keys of the chromosomes' concatenated TSVs
    (SURVEY 8(d)): a cluster that ends window k and opens window k+1 of a chromosome is in both windows' counts and is
    one site."""
    total, prev = 0, None
    for res in results:
        total += res.n_sites
        if prev is not None and prev.chrom == res.chrom and prev.edges[1] is not None and prev.edges[1] == res.edges[0]:
            total -= 1
        prev = res
    return total


def stitch_windows(results, options, sample):
    """WindowResults of any number of chromosomes in task order -> {chrom: (VCF body text, score text)}: the
    per-chromosome vote over the windows' held-back edge sites and their interior texts.  ``sample``: the Sample, or a
    callable (chromosome, window start) -> the Sample that served that window (file-driven runs hold one Sample per
    chromosome -- or per slice of it -- in flight, svision_amd/ingest.py): an edge site is genotyped on the records of the
    window that reported it."""
    from .network.predict import ChromosomeVote
    out, cur, bufs = {}, None, None
    for res in results:
        smp = sample(res.chrom, res.start) if callable(sample) else sample
        if res.chrom not in out:
            if cur is not None:
                cur.finish()
            bufs = (io.StringIO(), io.StringIO())
            out[res.chrom] = bufs
            cur = ChromosomeVote(res.chrom, bufs[0], bufs[1], options, smp)
        elif out[res.chrom] is not bufs:
            raise ValueError("windows of %s are not contiguous in task order" % res.chrom)
        cur.add(res.head, res.vcf, res.scores, res.tail, sample=smp)
    if cur is not None:
        cur.finish()
    return {c: (v.getvalue(), sc.getvalue()) for c, (v, sc) in out.items()}


class DeviceStage:
    """records [n,12] -> (softmax[5], class) per image, each launch a replay of one captured graph:
    svx_encode_conv1 -> active sets -> conv2..5 -> fc6/7 -> svx_fc8_softmax.

    ``batch`` is the reference's batch (the padding granule of BatchGenerator, create_batch.py:54-59); a launch carries
    ``launch_batches`` of them while that many are left, then half as many, ... down to single batches.  Every image is
    independent of its neighbours in every kernel (fixed k order per output element, tests/test_gpu_pipeline.py), so the
    grouping changes no result; it divides the fc6 / fc7 weight traffic per image (218 MB per launch whatever its size) and
    the share of partly filled tile rounds of the convolutions."""

    def __init__(self, net, batch, device, n_streams=2, use_graph=True, launch_batches=4, lazy=False):
        self.net, self.batch, self.device = net, batch, torch.device(device)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(n_streams)]
        self.use_graph = use_graph
        sizes, k = {batch}, max(1, int(launch_batches))
        while k > 1:                                          # launch_batches, half of it, ... one batch: a window's tail uses the largest that fits
            sizes.add(batch * k)
            k //= 2
        self.sizes = sorted(sizes, reverse=True)
        # per stream: {images per launch: (records, packed out, graph)}.  ``lazy``: a slot is built -- two eager passes, then the
        # capture -- when its first launch comes (a command line that classifies one small chromosome uses two or three of the
        # nine; a bench or a service builds them all up front so that no launch of a timed region pays for a capture)
        self.slots = [{} for _ in self.streams]
        if not lazy:
            for k in range(n_streams):
                for size in self.sizes:
                    self._slot(k, size)

    def _slot(self, k, size):
        slot = self.slots[k].get(size)
        if slot is None:
            s = self.streams[k]
            rec = torch.zeros((size, 12), dtype=torch.int32, device=self.device)
            rec[:] = torch.tensor(_PAD_REC, dtype=torch.int32, device=self.device)
            out = torch.empty((size, 12), dtype=torch.float32, device=self.device)   # softmax[5], class, logits[5], 0
            graph = None
            s.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(s):
                for _ in range(2 if self.use_graph else 1):      # the model's background tensors and the allocator's blocks before capture
                    self._body(rec, out)
            s.synchronize()
            if self.use_graph:
                graph = torch.cuda.CUDAGraph()
                # thread_local: the feeder thread of a file-driven run (ingest.ChromosomeFeed) uploads and synchronises on
                # its own stream while this thread captures; in the default global mode any such call invalidates the capture
                with torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):
                    self._body(rec, out)
            slot = self.slots[k][size] = (rec, out, graph)
        return slot

    def _body(self, rec, out):
        self.net.predict_records_packed(rec, out=out)        # no image tensor: encoding is fused into the first layer

    def run(self, d_rec, out, after=None, timing=None):
        """d_rec: int32 [n_padded,12] on the device (n_padded % batch == 0); out: float32 [n_padded,6].
        Enqueues everything asynchronously.

        ``after is None``: the side streams wait for the caller's current stream and that stream waits for them
        afterwards (simple, but consecutive calls are separated by a full drain of all streams).
        ``after`` = an event behind which ``d_rec`` is valid: the side streams wait for it only, nothing waits for them;
        returned are the events (one per stream used) behind which ``out`` is complete -- consecutive calls flow into
        each other, launch after launch, with no drain between two windows.  ``timing``: list that receives one
        (start, end, images) triple per launch, the events recorded on the launch's stream."""
        main = torch.cuda.current_stream(self.device)
        used = set()
        n_streams = len(self.streams)
        lo, n = 0, int(d_rec.shape[0])
        if n % self.batch:
            raise ValueError("DeviceStage.run: %d records are not a multiple of the batch (%d); pad with the reference's pad record" % (n, self.batch))
        while lo < n:
            b = next(size for size in self.sizes if size <= n - lo)
            k = self._next = (getattr(self, "_next", -1) + 1) % n_streams     # round-robin continues across calls
            s = self.streams[k]
            rec, o, graph = self._slot(k, b)
            if k not in used:
                if after is None:
                    s.wait_stream(main)
                else:
                    s.wait_event(after)
                used.add(k)
            with torch.cuda.stream(s):
                if timing is not None:
                    t0 = torch.cuda.Event(enable_timing=True)
                    t0.record()
                rec.copy_(d_rec[lo:lo + b], non_blocking=True)
                if graph is not None:
                    graph.replay()
                else:
                    self._body(rec, o)
                out[lo:lo + b].copy_(o[:, :6], non_blocking=True)
                if timing is not None:
                    t1 = torch.cuda.Event(enable_timing=True)
                    t1.record()
                    timing.append((t0, t1, b))
            lo += b
        if after is None:
            for k in used:
                main.wait_stream(self.streams[k])
            return out
        done = []
        for k in used:
            ev = torch.cuda.Event()
            ev.record(self.streams[k])
            done.append(ev)
            d_rec.record_stream(self.streams[k])
            out.record_stream(self.streams[k])
        return done


class HotPath:
    """Single-process form: collect -> device -> vote, with the device work of window k overlapped
    with the host collection of window k+1."""

    def __init__(self, sample, options, net, device="cuda", n_streams=2, use_graph=True, launch_batches=4, lazy_graphs=False):
        self.sample, self.options, self.net = sample, options, net
        self.device = torch.device(device)
        self.batch = options.batch_size
        self.stage = DeviceStage(net, self.batch, self.device, n_streams, use_graph, launch_batches, lazy=lazy_graphs)
        self.batch_events = []           # (start, end) event pair of every batch, on the batch's stream
        self.record_timing = bool(os.environ.get("SVX_TIMING"))      # bench.py switches it on; a plain run records no timing events
        self.device_images = 0           # images (padding included) launched since reset_timing()
        self._t_ref = None

    def collect(self, chrom, start, end, rescan=True):
        res = WindowResult()
        res.chrom, res.start, res.end = chrom, start, end
        if rescan:
            self.sample.rescan_window(chrom, start, end)
        res.lines = _collect_lines(self.sample, self.options, chrom, start, end)
        res.n_images = len(res.lines)
        res.records = np.asarray([ln.record() for ln in res.lines], np.int32).reshape(-1, 12)
        res.packed = None
        return res

    def empty(self, chrom, start, end):
        res = WindowResult()
        res.chrom, res.start, res.end, res.lines, res.n_images, res.packed = chrom, start, end, [], 0, None
        res.records = np.empty((0, 12), np.int32)
        return res

    def _pinned(self, rows):
        """A pinned host buffer of at least ``rows`` x 6 floats from a small free list (hipHostMalloc is slow)."""
        pool = self.__dict__.setdefault("_pinned_free", [])
        for i, buf in enumerate(pool):
            if buf.shape[0] >= rows:
                return pool.pop(i)
        return torch.empty((max(rows, 4096), 6), dtype=torch.float32, pin_memory=True)

    def launch(self, res):
        """Enqueue encode + CNN for the window's records and the copy of the packed predictions to pinned host memory on
        a stream of its own; returns immediately.  Nothing here -- and nothing in :meth:`fetch_predictions` -- synchronises
        with the launch stream: a blocking read-back there would wait for every window queued behind this one."""
        n = res.n_images
        res.t_device = None
        if n == 0:
            return res
        pad = (-n) % self.batch
        recs = np.concatenate([res.records, np.tile(np.asarray(_PAD_REC, np.int32), (pad, 1))]) if pad else res.records
        d_rec = torch.from_numpy(np.ascontiguousarray(recs)).to(self.device, non_blocking=True)
        out = torch.empty((n + pad, 6), dtype=torch.float32, device=self.device)
        up = torch.cuda.Event(enable_timing=self.record_timing)
        up.record()                                                   # behind the upload of the records
        if self.record_timing and getattr(self, "_t_ref", None) is None:
            self._t_ref = up                                          # time origin of batch_events
        done = self.stage.run(d_rec, out, after=up, timing=self.batch_events if self.record_timing else None)
        if getattr(self, "_d2h", None) is None:
            self._d2h = torch.cuda.Stream(device=self.device)
        host = self._pinned(n + pad)
        e2 = torch.cuda.Event()
        with torch.cuda.stream(self._d2h):
            for ev in done:
                self._d2h.wait_event(ev)
            host[:n + pad].copy_(out, non_blocking=True)
            e2.record()
        out.record_stream(self._d2h)
        res.packed, res.done_event, res.t_device = host, e2, None
        self.device_images += n + pad
        return res

    def device_busy_ms(self):
        """Milliseconds during which at least one batch was in flight on the device since the last reset_timing()
        (union of the per-batch [start, end] intervals recorded on the streams; call after a synchronize)."""
        if not self.batch_events:
            return 0.0
        ref = self._t_ref
        iv = sorted((ref.elapsed_time(ev[0]), ref.elapsed_time(ev[1])) for ev in self.batch_events)
        busy, cur_lo, cur_hi = 0.0, iv[0][0], iv[0][1]
        for lo, hi in iv[1:]:
            if lo > cur_hi:
                busy += cur_hi - cur_lo
                cur_lo, cur_hi = lo, hi
            else:
                cur_hi = max(cur_hi, hi)
        return busy + (cur_hi - cur_lo)

    def reset_timing(self):
        self.batch_events, self.device_images, self._t_ref = [], 0, None

    def fetch_predictions(self, res):
        if res.n_images == 0:
            return np.empty(0, np.int64), np.empty((0, 5), np.float32)
        res.done_event.synchronize()                                  # the copy stream's event: normally already complete
        packed = res.packed[:res.n_images].numpy().copy()
        self._pinned_free.append(res.packed)
        res.packed = None
        return packed[:, 5].astype(np.int64), packed[:, :5]

    def finish(self, res):
        classes, probs = self.fetch_predictions(res)
        res.vcf, res.scores, res.n_sites, res.head, res.tail = _vote(self.sample, self.options, res.chrom, res.lines, classes, probs, res.start, res.end)
        res.n_records = res.vcf.count("\n")
        res.edges = _edge_regions(res.lines)
        return res

    def run_windows(self, windows):
        prev = None
        for chrom, start, end in windows:
            cur = self.collect(chrom, start, end)
            if prev is not None:
                yield self.finish(prev)
            prev = self.launch(cur)
        if prev is not None:
            yield self.finish(prev)


# --------------------------------------------------------------------------------------------------
# Multi-process host: helper processes run collection and vote; the owner process runs the device.
_POOL_STATE = {}


def _worker_main(conn):
    """Helper process: never touches the GPU.  Protocol on the duplex pipe:
       owner -> ("win", wid, key, chrom, start, end, scan of the window's rows or None)
                    helper -> ("part", wid, records int32[k,12]) ... as the clusters are worked through, then
                    helper -> ("rec", wid, n images of the window, ok)
       owner -> ("pred", wid, classes, probs, last)  the window's predictions, in TSV order, in one or more messages
                    helper -> ("done", wid, vcf, scores, n_sites, n_images, tsv, head, tail, ...)   after the last one
       owner -> ("scan", min_sv, gaps.npy, gap_off.npy, stats.npy)   (HelperPool.attach_scan: helpers forked before the scan)
       owner -> ("chrom", key, meta) / ("drop", key)   a chromosome of a file-driven run arrives in / leaves shared memory
                                                        (ingest.ChromosomeFeed); key None = the Sample of the fork / "scan"
       owner -> ("opt", name, value)   a pool option changed (HelperPool.set_option: want_tsv on for bench.py's parity leg)
       owner -> ("stop",)"""
    sample, options = _POOL_STATE["sample"], _POOL_STATE["options"]
    from .segmentplot import run_hash_lineplot
    run_hash_lineplot.DEVICE = None           # helpers never touch the GPU
    if sample is not None:
        sample.device_buffers = None          # a helper forked from a live owner: its copy of the Sample is host-only
    # The cyclic collector finds nothing to free here (segments, signatures and lines die by reference count) but its
    # young-generation passes cost 20 % of a window and a full pass over the alignment table's objects ~70 ms: it runs
    # by hand, rarely.
    import gc
    import time
    gc.freeze()                               # (everything inherited from the owner, its garbage included, is never collected here: HelperPool.__init__)
    gc.disable()
    held = {}
    n_done = 0
    samples = {None: sample}
    header_dict = None
    while True:
        msg = conn.recv()
        if msg[0] == "stop":
            return
        if msg[0] == "chrom":
            meta = msg[2]
            if "references" in meta:                          # the header's dictionary travels with a helper's first part of a file only
                header_dict = (meta["references"], meta["lengths"])
            elif header_dict is None:
                raise RuntimeError("helper: part %r announced without the header's sequence dictionary" % (msg[1],))
            else:
                meta = dict(meta, references=header_dict[0], lengths=header_dict[1])
            if msg[1] not in samples:
                from .ingest import load_shared_sample
                samples[msg[1]] = load_shared_sample(meta, sample.fasta if sample is not None else _POOL_STATE.get("fasta"))
            continue
        if msg[0] == "drop":
            samples.pop(msg[1], None)
            continue
        if msg[0] == "opt":
            _POOL_STATE[msg[1]] = msg[2]
            continue
        if msg[0] == "scan":                                  # forked before the device scan existed: build the Sample now
            from .sample import Sample
            _t, min_sv, gaps, gap_off, stats = msg
            gaps, gap_off, stats = (np.load(p, mmap_mode="c") for p in (gaps, gap_off, stats))     # copy-on-write: windows' rescans land here
            sample = samples[None] = Sample.with_scan(_POOL_STATE["table"], _POOL_STATE["fasta"], min_sv, (gaps, gap_off, stats))
            continue
        if msg[0] == "win":
            _t, wid, key, chrom, start, end, scan = msg
            t0 = time.perf_counter()
            smp = samples[key]
            if scan is not None:
                smp.apply_window_scan(*scan)

            def emit(part, wid=wid):
                conn.send(("part", wid, np.asarray([ln.record() for ln in part], np.int32).reshape(-1, 12)))

            lines, ok = _collect_parts(smp, options, chrom, start, end, emit)
            held[wid] = [WindowVote(smp, options, chrom, lines, start, end), time.perf_counter() - t0, 0.0]
            conn.send(("rec", wid, len(lines), ok))
        elif msg[0] == "pred":
            _t, wid, classes, probs, last = msg
            state = held[wid]
            vote = state[0]
            t0 = time.perf_counter()
            if vote.lines:                                    # (a window whose collection failed has none: whatever was predicted is dropped)
                vote.feed(classes, probs)
            if not last:
                state[2] += time.perf_counter() - t0
                continue
            vcf, scores, n_sites, head, tail = vote.finish()
            lines = vote.lines
            tsv = "".join(ln.text() for ln in lines) if _POOL_STATE.get("want_tsv") else None
            del held[wid]
            conn.send(("done", wid, vcf, scores, n_sites, len(lines), tsv, head, tail, (state[1], state[2] + time.perf_counter() - t0), _edge_regions(lines)))
            n_done += 1
            if n_done % 128 == 0:
                gc.collect()


class HelperPool:
    """Forked host helpers.  Forking a process that owns a live GPU context is expensive on this stack (the driver
    evicts and restores the parent's queues around the copy-on-write protection of its pinned ranges: the first device
    work after 16 forks stalled for 3.5 s, and for over a minute in a long-lived process holding gigabytes of device allocations), so a command-line run creates the pool *before* the first HIP call, with the
    decoded table and the reference only, and sends the device scan's result afterwards (``attach_scan``).  Given a
    complete ``sample`` the helpers are usable at once (bench: the fork cost falls into the warm-up)."""

    def __init__(self, n_workers, options, sample=None, table=None, fasta=None, want_tsv=False):
        _POOL_STATE.update(sample=sample, table=table, fasta=fasta, options=options, want_tsv=want_tsv)
        ctx = mp.get_context("fork")                         # helpers inherit the resident host arrays copy-on-write
        # Cyclic garbage that holds device objects (a traceback that kept a failed call's tensors and events alive, say) is
        # collected HERE, by the process that owns the device: a forked helper that collected it would free them through a
        # runtime it must not touch (a segmentation fault in the helpers' first collection, seen once a test in the same process
        # had raised out of kernels.cigar_scan).
        import gc
        gc.collect()
        self.conns, self.procs = [], []
        for _ in range(n_workers):
            a, b = ctx.Pipe(duplex=True)
            p = ctx.Process(target=_worker_main, args=(b,), daemon=True)
            p.start()
            b.close()
            self.conns.append(a)
            self.procs.append(p)
        _POOL_STATE.clear()

    def attach_scan(self, sample):
        """Hand the device scan's result to the helpers through one set of files in shared memory (mapped read-only by
        every helper) rather than through N pipes."""
        import tempfile
        shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        self._scan_dir = tempfile.mkdtemp(prefix="svx_scan_", dir=shm)
        paths = []
        for name, arr in (("gaps", sample.gaps), ("gap_off", sample.gap_off), ("stats", sample.stats)):
            path = os.path.join(self._scan_dir, name + ".npy")
            np.save(path, np.ascontiguousarray(arr))
            paths.append(path)
        for c in self.conns:
            c.send(("scan", sample.min_sv, *paths))

    def set_option(self, name, value):
        """Change a pool option (``want_tsv``) in every helper; takes effect with the next window a helper finishes."""
        for c in self.conns:
            c.send(("opt", name, value))

    def close(self):
        for c in self.conns:
            try:
                c.send(("stop",))
            except (BrokenPipeError, OSError):
                pass
        for p in self.procs:
            p.join(timeout=5)
        self.conns, self.procs = [], []
        if getattr(self, "_scan_dir", None):
            import shutil
            shutil.rmtree(self._scan_dir, ignore_errors=True)
            self._scan_dir = None


class ImageQueue:
    """The images of all windows that wait for the device, in arrival order: the helpers' parts go in (``add``), launch groups
    of any size come out (``take``) together with where every image belongs -- (window id, offset in the window, offset in the
    group, count) -- so that the predictions find their way back to the windows' votes."""

    def __init__(self):
        self.parts = collections.deque()      # [wid, offset in the window, records, arrival time]
        self.images = 0
        self.seen = {}                        # wid -> images of the window queued so far

    def add(self, wid, records, now=0.0):
        n = int(records.shape[0])
        if n:
            self.parts.append([wid, self.seen.get(wid, 0), records, now])
            self.seen[wid] = self.seen.get(wid, 0) + n
            self.images += n

    def oldest(self):
        return self.parts[0][3] if self.parts else None

    def take(self, n):
        """The first ``n`` queued images -> (records [n, 12], [(wid, window offset, group offset, count)])."""
        if n > self.images or n <= 0:
            raise ValueError("ImageQueue.take(%d) with %d images queued" % (n, self.images))
        recs, mapping, off = [], [], 0
        while off < n:
            part = self.parts[0]
            k = min(n - off, int(part[2].shape[0]))
            recs.append(part[2][:k])
            mapping.append((part[0], part[1], off, k))
            off += k
            if k == part[2].shape[0]:
                self.parts.popleft()
            else:
                part[1] += k
                part[2] = part[2][k:]
        self.images -= n
        return (recs[0] if len(recs) == 1 else np.concatenate(recs)), mapping

    def drop(self, wid):
        """Forget what is queued of a window (its collection failed after parts had left)."""
        keep = collections.deque(p for p in self.parts if p[0] != wid)
        self.images -= sum(int(p[2].shape[0]) for p in self.parts if p[0] == wid)
        self.parts = keep
        self.seen.pop(wid, None)


class PooledHotPath(HotPath):
    """Owner process = device feeder; the helpers of a :class:`HelperPool` do the Python glue."""

    def __init__(self, sample, options, net, device="cuda", n_workers=8, n_streams=2, use_graph=True, max_inflight=3, launch_batches=4,
                 want_tsv=False, pool=None, feed=None, lazy_graphs=False):
        super().__init__(sample, options, net, device, n_streams, use_graph, launch_batches, lazy_graphs=lazy_graphs)
        self.pool = pool if pool is not None else HelperPool(n_workers, options, sample=sample, want_tsv=want_tsv)
        self.conns, self.procs = self.pool.conns, self.pool.procs
        self.max_inflight = max_inflight
        from .ingest import StaticFeed
        self._chrom_meta = {}                                          # key -> meta of the parts of a file-driven run that are alive
        self._helper_keys = [set() for _ in self.conns]                # per helper: the keys it has been told about
        self._helper_has_dict = [False] * len(self.conns)              # per helper: has it received the header's sequence dictionary
        self._feed = None
        self.feed = feed if feed is not None else StaticFeed(sample)   # where a chromosome's Sample comes from
        self.keep_predictions = False                                  # WindowResult.classes / .probs: the window's predictions in TSV order (bench.py's parity leg)

    @property
    def feed(self):
        return self._feed

    @feed.setter
    def feed(self, feed):
        """Another feed (bench.py: the file-driven legs and the resident one take turns; a service: the next file): whatever the
        helpers still hold of the previous one is dropped, and the next part every helper is told about carries the header's
        sequence dictionary again -- it belongs to the file, not to the pool.  (Keys are unique in the process,
        ingest._KEYS, so a part of the old feed can never be taken for one of the new.)"""
        if feed is self._feed:
            return
        for ci, c in enumerate(self.conns):
            for key in sorted(self._helper_keys[ci]):
                c.send(("drop", key))
            self._helper_keys[ci].clear()
        self._chrom_meta = {}
        self._helper_has_dict = [False] * len(self.conns)
        self._feed = feed

    def release(self, chrom):
        """A chromosome of a file-driven run is finished (voted, stitched): the helpers unmap its parts, the feed frees them."""
        try:
            keys = self.feed.keys_of(chrom)
        except KeyError:
            return
        for key in keys:
            self._chrom_meta.pop(key, None)
            for ci, c in enumerate(self.conns):               # only the helpers that were told about it (see _announce)
                if key in self._helper_keys[ci]:
                    self._helper_keys[ci].discard(key)
                    c.send(("drop", key))
        self.feed.release(chrom)

    def _announce(self, ci, key):
        """Tell helper ``ci`` where chromosome ``key`` lies in shared memory -- right in front of the first window of it that
        the helper is given, i.e. while it is idle and reading its pipe.  (Until round 4 every arriving chromosome was
        broadcast to every helper with blocking sends, busy ones included, its meta carrying the header's whole sequence
        dictionary: a helper in the middle of a large window does not read its pipe while it sends parts of that window
        to this thread -- a few such messages fill its socket buffer and both ends block for ever.  ADVICE r3.)"""
        if key is None or key in self._helper_keys[ci]:
            return
        meta = self._chrom_meta[key]
        if self._helper_has_dict[ci]:
            meta = {k: v for k, v in meta.items() if k not in ("references", "lengths")}
        self.conns[ci].send(("chrom", key, meta))
        self._helper_has_dict[ci] = True
        self._helper_keys[ci].add(key)

    def close(self):
        self.pool.close()
        self.conns, self.procs = [], []

    def run_windows(self, windows, rescan=True):
        """Yields WindowResults (completion order).  Each helper holds one window at a time.

        The device side is one **stream of images**: the helpers hand a window's records on in parts (whole clusters,
        >= 256 lines) while they work through the later clusters, the parts of all windows queue up in arrival order and
        leave in launches of 256 images whatever window they belong to -- every image is independent of its neighbours
        in every kernel (``DeviceStage``), so the grouping changes no result; what it removes is the per-window tail of
        128- / 64-image launches with their padding, and the wait for a window's last cluster before its first launch.
        Fewer than 256 images are launched (padded to the reference's batch) only when the device would otherwise idle,
        when a part has waited ``flush_after`` seconds, or when nothing more can arrive.

        ``self.owner_profile`` afterwards: where this (GPU-owning) thread spent its time -- seconds per phase of the
        loop, the scans found ready on entry, the host seconds the helpers report for collection and vote, and when the
        first launch / last read-back / last result happened (SVX_TIMING=1 python bench.py prints it)."""
        import time
        windows = list(windows)
        nxt = 0
        idle = list(range(len(self.conns)))
        busy = {}                     # conn index -> wid
        wins = {}                     # wid -> state of a window whose predictions are not complete yet
        pending = ImageQueue()        # the parts not launched yet
        inflight = collections.deque()  # (launch group as a WindowResult, [(wid, window offset, group offset, count)])
        inflight_images = 0
        collecting = 0                # windows sent to a helper whose last part has not arrived
        granule = 4 * self.batch if 4 * self.batch in self.stage.sizes else max(self.stage.sizes)
        cap_images = max(1, self.max_inflight) * 8 * granule
        flush_after = 0.002
        scans = collections.deque()     # handles of the window scans enqueued ahead (Sample.rescan_window_async)
        ahead = min(16, len(self.conns) + 2)    # every helper can turn idle in one burst; a scan takes ~15 ms on the saturated device
        remaining = len(windows)
        kept = {}                     # wid -> [(window offset, classes, probs)] (keep_predictions)
        prof = self.owner_profile = collections.defaultdict(float)
        clock = time.perf_counter
        t_loop = clock()

        def lap(key, since):
            now = clock()
            prof[key] += now - since
            return now

        def take(n):
            """The first ``n`` pending images as one launch group."""
            records, mapping = pending.take(n)
            group = WindowResult()
            group.records, group.n_images, group.lines, group.packed = records, n, None, None
            return group, mapping

        def forward(wid):
            """Send the predictions that have arrived for a window on to its helper, which votes as they come.  Only
            once the helper has reported the window's size: until then it is collecting, not reading its pipe, and a
            send could block this thread on a full pipe."""
            w = wins[wid]
            if w["total"] is None:
                return
            last = w["got"] == w["total"]
            chunks, w["chunks"] = w["chunks"], []
            if chunks:
                classes = chunks[0][1] if len(chunks) == 1 else np.concatenate([c[1] for c in chunks])
                probs = chunks[0][2] if len(chunks) == 1 else np.concatenate([c[2] for c in chunks])
            elif last:
                classes, probs = np.empty(0, np.int64), np.empty((0, 5), np.float32)
            else:
                return
            self.conns[w["ci"]].send(("pred", wid, classes, probs, last))
            if last:
                del wins[wid]

        while remaining:
            t = clock()
            while idle and nxt < len(windows):
                chrom, start, end = windows[nxt]
                key, smp = self.feed.get(chrom, block=False, start=start)     # file-driven runs: is the window's part of the chromosome decoded + scanned yet?
                for k, _c, meta in self.feed.take_fresh():            # where it lies in shared memory: told to a helper with its first window of it
                    self._chrom_meta[k] = meta
                if smp is None:
                    prof["feed.not_ready"] += 1
                    break
                ci = idle.pop()
                self._announce(ci, key)
                scan = None
                if rescan:                                            # device scan of the window's block: the helper collects on ITS result
                    t_s = clock()
                    want = ahead if nxt else 1                        # the very first window leaves before the scans of the next ones are enqueued (0.25 ms each)
                    while len(scans) < want and nxt + len(scans) < len(windows):     # enqueued several windows ahead, read back here
                        scans.append(self.sample.rescan_window_async(*windows[nxt + len(scans)]))
                    t_s = lap("scan.enqueue", t_s)
                    handle = scans.popleft()
                    if handle is not None:
                        prof["scan.n"] += 1
                        prof["scan.ready_on_entry"] += 1 if handle[5].query() else 0
                        prof["scan.age_s"] += clock() - handle[7]
                        handle[5].synchronize()
                    t_s = lap("scan.sync", t_s)
                    scan = self.sample.last_window_scan if self.sample.finish_rescan(handle) else None
                    lap("scan.apply", t_s)
                self.conns[ci].send(("win", nxt, key, chrom, start, end, scan))
                busy[ci] = nxt
                wins[nxt] = {"ci": ci, "total": None, "got": 0, "chunks": []}
                collecting += 1
                nxt += 1
            t = lap("scan+send", t)
            while pending.images and inflight_images < cap_images:
                room = cap_images - inflight_images
                if pending.images >= granule:
                    # small groups: a window's predictions return as soon as its own launches are done (and, when nothing
                    # is being collected any more, launch by launch: the last vote is what the job's end waits for)
                    n = min(pending.images, room, granule if collecting == 0 and nxt >= len(windows) else 2 * granule) // granule * granule
                    if n == 0:
                        break
                elif not inflight or collecting == 0 or clock() - pending.oldest() > flush_after:
                    n = pending.images                                # the device would idle / nothing else can arrive / it has waited
                    prof["launch.partial"] += 1
                else:
                    break
                group, mapping = take(n)
                prof.setdefault("first_launch_at", clock() - t_loop)
                inflight.append((self.launch(group), mapping))
                inflight_images += n
            t = lap("launch", t)
            while inflight and inflight[0][0].done_event.query():
                group, mapping = inflight.popleft()
                inflight_images -= group.n_images
                classes, probs = self.fetch_predictions(group)
                touched = {}
                for wid, w_off, g_off, k in mapping:
                    w = wins.get(wid)
                    if w is None or w.get("drop"):
                        continue
                    w["chunks"].append((w_off, classes[g_off:g_off + k], probs[g_off:g_off + k]))     # groups complete in launch order: window order
                    if self.keep_predictions:
                        kept.setdefault(wid, []).append((w_off, np.array(classes[g_off:g_off + k]), np.array(probs[g_off:g_off + k])))
                    w["got"] += k
                    touched[wid] = True
                for wid in touched:
                    forward(wid)
                prof["last_fetch_at"] = clock() - t_loop
            t = lap("fetch+send", t)
            waiting = [self.conns[ci] for ci in busy]
            if not waiting and not inflight and not pending.images and nxt < len(windows):
                self.feed.poll(block=True)                            # nothing to do but wait for the next chromosome
                lap("feed.wait", t)
                continue
            got = mpc.wait(waiting, timeout=0.0005 if inflight or pending.images else (0.002 if nxt < len(windows) and idle else 0.05))
            lap("wait", t)
            if not got and not inflight and not pending.images:
                dead = [ci for ci in busy if not self.procs[ci].is_alive()]
                if dead:
                    raise RuntimeError("host helper process %s died while holding window %s" % (dead, [busy[ci] for ci in dead]))
            for c in got:
                ci = self.conns.index(c)
                msg = c.recv()
                if msg[0] == "part":
                    pending.add(msg[1], msg[2], clock())
                elif msg[0] == "rec":
                    _t, wid, n_images, ok = msg
                    w = wins[wid]
                    collecting -= 1
                    if not ok:                                        # the window's collection failed after parts had left: drop them
                        pending.drop(wid)
                        w["chunks"], w["total"], w["got"] = [], 0, 0
                        forward(wid)                                  # an empty last message: the helper votes on no lines
                        wins[wid] = {"drop": True}                    # launches of it still in flight are ignored (entry never removed: wids are not reused)
                        continue
                    w["total"] = n_images
                    forward(wid)
                else:
                    _t, wid, vcf, scores, n_sites, n_images, tsv, head, tail, host_s, edges = msg
                    prof["helper.collect_s"] += host_s[0]             # host seconds inside the helpers
                    prof["helper.vote_s"] += host_s[1]
                    res = WindowResult()
                    res.chrom, res.start, res.end = windows[wid]
                    res.wid, res.tsv = wid, tsv
                    res.vcf, res.scores, res.n_sites, res.n_images = vcf, scores, n_sites, n_images
                    res.head, res.tail, res.edges = head, tail, edges
                    res.n_records = vcf.count("\n")
                    res.classes = res.probs = None
                    if self.keep_predictions:
                        parts = sorted(kept.pop(wid, []), key=lambda t: t[0])
                        res.classes = np.concatenate([t[1] for t in parts]) if parts else np.empty(0, np.int64)
                        res.probs = np.concatenate([t[2] for t in parts]) if parts else np.empty((0, 5), np.float32)
                    del busy[ci]
                    idle.append(ci)
                    remaining -= 1
                    prof["last_done_at"] = clock() - t_loop
                    yield res
