"""The ``SVision`` command line on MI355X: same arguments, same outputs.

Mirror of the reference driver (``/root/reference/SVision``): argument surface :27-106,
input checks :141-157, window tasking :164-234 (including the ``-c chr:a-b`` quirk that
windows restart at 0), Step 1 collection :259-294, Step 2 prediction :296-328, score range +
merge :331-339, cleanup :370-372.  The two process pools are replaced by one process per
GPU (chromosomes sharded across ranks when launched under torchrun, see dist.py).
"""
import argparse
import datetime
import logging
import os
import shutil
import sys
from time import localtime, strftime

import numpy as np

from . import __version__, dist as sdist
from .io.bam import Fasta, read_bam

REFERENCE_VERSION = "1.4"      # ##source line of the VCF the reference writes (src/version.py)


def parse_arguments(arguments=None):
    p = argparse.ArgumentParser(formatter_class=argparse.RawDescriptionHelpFormatter,
                                description="SVision (MI355X hot path %s)\n\nShort Usage: SVision [parameters] -o <output path> "
                                            "-b <input bam path> -g <reference> -m <model path>" % __version__)
    g = p.add_argument_group("Input/Output parameters")
    g.add_argument("-o", dest="out_path", type=os.path.abspath, required=True, help="Absolute path to output")
    g.add_argument("-b", dest="bam_path", type=os.path.abspath, required=True, help="Absolute path to bam file")
    g.add_argument("-m", dest="model_path", type=os.path.abspath, required=True, help="Absolute path to CNN predict model")
    g.add_argument("-g", dest="genome", type=os.path.abspath, required=True, help="Absolute path to your reference genome")
    g.add_argument("-n", dest="sample", type=str, required=True, help="Name of the BAM sample name")
    g = p.add_argument_group("Optional parameters")
    g.add_argument("-t", dest="thread_num", type=int, default=1, help="Thread numbers (default: %(default)s)")
    g.add_argument("-s", dest="min_support", type=int, default=5, help="Minimum support read number required for SV calling (default: %(default)s)")
    g.add_argument("-c", dest="chrom", type=str, default=None, help="Specific region (chr1:xxx-xxx) or chromosome (chr1) to detect")
    g.add_argument("--hash", action="store_true", default=False, help="Activate local realignment for unmapped sequences (default: %(default)s)")
    g.add_argument("--qname", action="store_true", default=False, help="Report support names for each events (default: %(default)s)")
    g.add_argument("--graph", action="store_true", default=False, help="Report graph for events (default: %(default)s)")
    g.add_argument("--contig", action="store_true", default=False, help="Activate contig mode (default: %(default)s)")
    g.add_argument("--debug", action="store_true", default=False, help="Activate debug mode and keep intermedia outputs (default: %(default)s)")
    g = p.add_argument_group("Collect parameters")
    g.add_argument("--min_mapq", type=int, default=10, help="Minimum mapping quality of reads to consider (default: %(default)s)")
    g.add_argument("--min_sv_size", type=int, default=50, help="Minimum SV size to detect (default: %(default)s)")
    g.add_argument("--max_sv_size", type=int, default=1000000, help="Maximum SV size to detect (default: %(default)s)")
    g.add_argument("--window_size", type=int, default=10000000, help="The sliding window size in segment collection (default: %(default)s)")
    g = p.add_argument_group("Cluster parameters")
    g.add_argument("--patition_max_distance", type=int, default=5000, help="Maximum distance to partition signatures (default: %(default)s)")
    g.add_argument("--cluster_max_distance", type=float, default=0.3, help="Clustering maximum distance for a partition (default: %(default)s)")
    g = p.add_argument_group("Predict parameters")
    g.add_argument("--batch_size", type=int, default=128, help="Batch size for the CNN prediction model (default: %(default)s)")
    g = p.add_argument_group("Genotype parameters")
    g.add_argument("--min_gt_depth", type=int, default=4, help="Minimum reads required for genotyping (default: %(default)s)")
    g.add_argument("--homo_thresh", type=float, default=0.8, help="Minimum variant allele frequency to be called as homozygous (default: %(default)s)")
    g.add_argument("--hete_thresh", type=float, default=0.2, help="Minimum variant allele frequency to be called as heterozygous (default: %(default)s)")
    g = p.add_argument_group("Hash table parameters")
    g.add_argument("--k_size", type=int, default=10, help="Size of kmer (default: %(default)s)")
    g.add_argument("--min_accept", type=int, default=50, help="Minimum match length for realignment (default: %(default)s)")
    g.add_argument("--max_hash_len", type=int, default=1000, help="Maximum length of unmapped sequence length for realignment (default: %(default)s)")
    return p.parse_args(sys.argv[1:] if arguments is None else arguments)


def build_tasks(options, references, lengths, fasta_refs):
    """{chrom: [[start, end], ...]} in BAM-header order (SVision:164-234)."""
    length_of = dict(zip(references, lengths))
    window = options.window_size
    tasks = {}
    if options.chrom is None:
        for chrom in references:
            n = length_of[chrom]
            if chrom not in fasta_refs:
                continue
            if options.contig:
                window = n
            if n < window:
                tasks.setdefault(chrom, []).append([0, n])
                continue
            pos = 0
            for _ in range(int(n / window)):
                tasks.setdefault(chrom, []).append([pos, pos + window])
                pos += window
            if pos < n:
                tasks.setdefault(chrom, []).append([pos, n])
        return tasks
    chrom = options.chrom
    if chrom in fasta_refs:
        start, end = 0, length_of[chrom]
    else:
        cords = chrom.split(":")[1]
        chrom, start, end = chrom.split(":")[0], int(cords.split("-")[0]), int(cords.split("-")[1])
    tasks[chrom] = []
    region_length = end - start + 1
    if region_length < window:
        tasks[chrom].append([start, end])
    else:
        pos = 0                                           # upstream restarts at 0, not at `start`
        for _ in range(int(region_length / window)):
            tasks[chrom].append([pos, pos + window])
            pos += window
        if pos < region_length:
            tasks[chrom].append([pos, region_length])
    return tasks


class _Ticker:
    """SVX_TIMING=1: wall time of the phases of a run on stdout."""

    def __init__(self):
        import time
        self.clock, self.last, self.on = time.time, time.time(), bool(os.environ.get("SVX_TIMING"))

    def __call__(self, what):
        now = self.clock()
        if self.on:
            print("%-36s %.3f s" % (what, now - self.last), flush=True)
        self.last = now


def load_rank_table(options, rank, ws):
    """The alignment records rank ``rank`` of ``ws`` needs.  With a ``.bai`` next to the BAM and more than one rank, a
    rank is a set of chromosomes: the header gives the task list, the index gives the byte range holding this rank's
    records, and only that range is read and inflated (upstream: AlignmentFile.fetch(chrom), run_collection.py:26)."""
    index = next((c for c in (options.bam_path + ".bai", os.path.splitext(options.bam_path)[0] + ".bai") if os.path.exists(c)), None)
    if ws == 1 or index is None:
        return read_bam(options.bam_path, with_seq=bool(options.hash or options.graph))
    head = read_bam(options.bam_path, tids=[], index=index)
    tasks = build_tasks(options, head.references, head.lengths, Fasta(options.genome).references)
    length_of = dict(zip(head.references, head.lengths))
    shard = sdist.shard_chromosomes(list(tasks), [length_of.get(c, 1) for c in tasks], ws)[rank]
    table = read_bam(options.bam_path, with_seq=bool(options.hash or options.graph), tids=[head.references.index(c) for c in shard], index=index)
    logging.info("rank %d/%d: %d records of %s decoded through %s", rank, ws, len(table), ",".join(shard) or "-", index)
    return table


def run(options, sample=None, classifier=None):
    """Whole pipeline; returns the merged VCF path (rank 0) or None."""
    from . import sample as _sample
    from .collection import run_collection
    from .network.output import cal_scores_max_min, merge_split_vcfs
    from .network.predict import Predict

    rank, ws = sdist.env_rank()              # the process group comes up after the host helpers are forked (below)
    work_dir = options.out_path
    os.makedirs(work_dir, exist_ok=True)
    graph_dir = os.path.join(work_dir, "graphs")
    if options.graph:                        # SVision:253-256; before the helpers are forked: they write the per-read graphs
        os.makedirs(graph_dir, exist_ok=True)
    fmt = logging.Formatter("%(asctime)s [%(levelname)-7.7s]  %(message)s")
    root = logging.getLogger()
    root.setLevel(logging.INFO)
    fh = logging.FileHandler("%s/SVision_%s%s.log" % (work_dir, strftime("%y%m%d_%H%M%S", localtime()),
                                                     "" if ws == 1 else ".rank%d" % rank), mode="w")
    fh.setFormatter(fmt)
    root.addHandler(fh)
    logging.info("******************** Start SVision, version %s (svision_amd %s) ********************", REFERENCE_VERSION, __version__)
    logging.info("CMD: %s", " ".join(sys.argv))
    logging.info("WORKDIR DIR: %s", os.path.abspath(work_dir))
    logging.info("CNN MODEL: %s", os.path.abspath(options.model_path))
    logging.info("INPUT BAM: %s", os.path.abspath(options.bam_path))

    pool = None
    feed = None
    _tick = _Ticker()
    if sample is None:
        # file-driven run: header now, the records chromosome by chromosome while the pipeline runs (ingest.ChromosomeFeed)
        from .io.bam import find_index, read_bam_header
        head = read_bam_header(options.bam_path)
        if head.sort_order != "coordinate":
            logging.error("This is not a coordinate sorted BAM file")
            raise SystemExit(1)
        fasta = Fasta(options.genome)
        references, lengths = head.references, head.lengths
    else:
        fasta = sample.fasta
        references, lengths = sample.table.references, sample.table.lengths
        _sample.register(options.bam_path, sample)
    if options.contig:
        options.min_support = 1
    tasks = build_tasks(options, references, lengths, fasta.references)
    if len(tasks) == 0:
        logging.error("No mapped reads in the BAM, please check your reference input!")
        raise SystemExit(1)
    chroms = list(tasks.keys())
    length_of = dict(zip(references, lengths))
    mine = sdist.shard_chromosomes(chroms, [length_of.get(c, 1) for c in chroms], ws)[rank]
    if classifier is None:
        if options.thread_num > 1 and sample is None:
            # -t N: fork the host helpers before the first HIP call (pipeline.HelperPool); they map every chromosome
            # from shared memory when the feed announces it
            from .pipeline import HelperPool
            pool = HelperPool(options.thread_num, options, fasta=fasta, want_tsv=True)
        from .build_host import compiled_state
        _compiled, _interp = compiled_state()
        if _interp:
            logging.warning("host modules running interpreted (2-4x slower collection and vote): %s -- build them with `python -m svision_amd.build_host`", ", ".join(_interp))
        _tick("header, FASTA index, fork helpers")
        # one process per GPU: every device tensor of this rank (scan buffers, weights, graphs) lives on its own GPU.
        # The process group comes up now (after the fork, before the first long phase), not when the first rank is done:
        # a late rendezvous would time out whenever the shards finish far apart.
        import torch
        if torch.cuda.is_available():
            torch.cuda.set_device(sdist.local_device_index())
        sdist.init_from_env()
        from .ingest import ChromosomeFeed, StaticFeed
        if sample is None:
            from .ingest import decode_threads
            threads = decode_threads(int(os.environ.get("LOCAL_WORLD_SIZE", ws)), options.thread_num)      # inflate threads of this rank
            feed = ChromosomeFeed(options.bam_path, fasta, options, [c for c in mine if c in references], references, lengths,
                                  device=torch.device("cuda", torch.cuda.current_device()), index=find_index(options.bam_path), threads=threads,
                                  tasks={c: tasks[c] for c in mine if c in tasks})
            logging.info("rank %d/%d: %s streamed from %s with %d decode threads", rank, ws, ",".join(mine) or "-", options.bam_path, threads)
        else:
            feed = StaticFeed(sample)

    seg_dir = os.path.join(work_dir, "segments")
    pred_dir = os.path.join(work_dir, "predict_results")
    os.makedirs(seg_dir, exist_ok=True)
    os.makedirs(pred_dir, exist_ok=True)
    t0 = datetime.datetime.now()
    if classifier is None:
        # device path: Step 1 and Step 2 streamed window by window (collection of window k+1 on the host while the
        # device classifies window k), one vote stream per chromosome in window order = the order of all.bed
        try:
            if options.thread_num > 1:
                _run_pooled(options, feed, tasks, mine, seg_dir, pred_dir, pool)
            else:
                _run_streaming(options, feed, tasks, mine, seg_dir, pred_dir)
        finally:
            feed.close()
        if _tick.on and hasattr(feed, "stats"):
            print("ingest: %s" % {k: (round(v, 3) if isinstance(v, float) else v) for k, v in feed.stats.items()}, flush=True)
        t1 = t2 = datetime.datetime.now()
        logging.info("[Coding + prediction finished]: streamed, Cost time: %s", (t2 - t0).seconds)
    else:
        logging.info("\n****************** Step1 Image coding and segmentation ******************")
        for chrom in mine:
            parts = []
            for part, (start, end) in enumerate(tasks[chrom]):
                err = run_collection.run_detect(options, options.bam_path, chrom, part, start, end)
                if err is not None:
                    logging.error("%s:%s-%s %s", chrom, start, end, err)      # upstream drops this string silently
                parts.append(os.path.join(seg_dir, "%s.segments.%d.bed" % (chrom, part)))
            with open(os.path.join(seg_dir, chrom + ".segments.all.bed"), "w") as out:   # `cat parts > all.bed`
                for p in parts:
                    if os.path.exists(p):
                        with open(p) as f:
                            shutil.copyfileobj(f, out)
        t1 = datetime.datetime.now()
        logging.info("[Coding finished]: Collect segment signatures, Cost time: %s", (t1 - t0).seconds)

        logging.info("\n****************** Step2 CNN prediction ******************")
        for chrom in mine:
            prefix = os.path.join(pred_dir, "%s.predict.s%s" % (chrom, options.min_support))
            Predict(chrom, os.path.join(seg_dir, chrom + ".segments.all.bed")).run(prefix, options, classifier=classifier, sample=sample)
        t2 = datetime.datetime.now()
        logging.info("[Prediction finished]: Predicting types, Cost time: %s", (t2 - t1).seconds)

    _tick("collection + encode + CNN + vote")
    # ---- the single cross-shard exchange: score range + record gather ----
    sdist.init_from_env()
    local_scores = cal_scores_max_min(pred_dir) if ws == 1 else _scores_of(pred_dir, mine, options)
    max_score, min_score = sdist.exchange_score_range(local_scores)
    if max_score is None:
        print("Empty output in the score file!!! Program exit")
        raise SystemExit(0)
    merged_path = os.path.join(options.out_path, "%s.svision.s%s.vcf" % (options.sample, options.min_support))
    grouped = sdist.world_initialized()      # ws > 1, or one rank with SVX_FORCE_DIST=1 (the RCCL path on a one-GPU box)
    if grouped:
        bodies = {}
        for chrom in mine:
            with open(os.path.join(pred_dir, "%s.predict.s%s.vcf" % (chrom, options.min_support))) as f:
                bodies[chrom] = f.read()
        bodies = sdist.gather_texts(bodies, dst=0)
        if rank == 0:
            for chrom, text in bodies.items():
                with open(os.path.join(pred_dir, "%s.predict.s%s.vcf" % (chrom, options.min_support)), "w") as f:
                    f.write(text)
        if options.graph:
            # the per-read graphs of a reported cluster are written by the rank that collected it (graphs/{chrom}-{start}-{end}/
            # {read}.gfa); step 3 below runs on rank 0, and the out_path need not be a filesystem the ranks share: the
            # graph texts travel with the VCF bodies
            mine_set, texts = set(mine), {}
            for name in sorted(os.listdir(graph_dir)):
                d = os.path.join(graph_dir, name)
                if os.path.isdir(d) and name.rsplit("-", 2)[0] in mine_set:
                    for fn in sorted(os.listdir(d)):
                        with open(os.path.join(d, fn)) as f:
                            texts[name + "/" + fn] = f.read()
            texts = sdist.gather_texts(texts, dst=0)
            if rank == 0:
                for rel, text in texts.items():
                    path = os.path.join(graph_dir, rel)
                    if not os.path.exists(path):
                        os.makedirs(os.path.dirname(path), exist_ok=True)
                        with open(path, "w") as f:
                            f.write(text)
    if rank == 0:
        options.source_version = REFERENCE_VERSION
        merge_split_vcfs(pred_dir, merged_path, max_score, min_score, chroms, options, fasta=fasta)
        if options.graph:                    # SVision:341-359: graph VCF + summaries; the plain VCF and the per-site folders go
            from .collection.graph import annotate_vcf_with_graphs
            logging.info("\n****************** Step3 Computing graphs ******************")
            annotate_vcf_with_graphs(graph_dir, merged_path, options)
            for name in os.listdir(graph_dir):
                if os.path.isdir(os.path.join(graph_dir, name)):
                    shutil.rmtree(os.path.join(graph_dir, name))
            os.remove(merged_path)
            merged_path = os.path.join(options.out_path, "%s.svision.s%s.graph.vcf" % (options.sample, options.min_support))
            logging.info("[Graph creation finished] Generate graphs")
        logging.info("[All steps finished] Total Cost time: %ss", (datetime.datetime.now() - t0).seconds)
    _tick("exchange + merge")
    if grouped:
        import torch.distributed as tdist
        if _tick.on:
            print("exchange backend %s, world %d" % (tdist.get_backend(), tdist.get_world_size()), flush=True)
        logging.info("cross-rank exchange over %s, world size %d", tdist.get_backend(), tdist.get_world_size())
        tdist.barrier()
    if not options.debug and rank == 0:
        shutil.rmtree(seg_dir, ignore_errors=True)
        shutil.rmtree(pred_dir, ignore_errors=True)
    root.removeHandler(fh)
    fh.close()
    return merged_path if rank == 0 else None


def _run_streaming(options, feed, tasks, chroms, seg_dir, pred_dir):
    """Steps 1 + 2 without the TSV round trip: same functions, same order of lines, same vote semantics as
    Predict.run over ``{chrom}.segments.all.bed`` (predict.py:206-300); the segment files are still written."""
    import sys
    import traceback
    from .network.predict import Predict, SiteVoter, load_network
    from .pipeline import HotPath
    import time as _time
    _t0 = _time.time()
    net = load_network(options.model_path)
    hot = HotPath(None, options, net, n_streams=3, lazy_graphs=True)      # a command line captures the launch shapes it meets (pipeline.DeviceStage)
    _t1 = _time.time()
    for chrom in chroms:
        prefix = os.path.join(pred_dir, "%s.predict.s%s" % (chrom, options.min_support))
        with open(prefix + ".score.txt", "w") as score_out, open(prefix + ".vcf", "w") as vcf_out, \
                open(os.path.join(seg_dir, chrom + ".segments.all.bed"), "w") as all_bed:
            voter = SiteVoter(Predict(chrom, None), vcf_out, score_out, options, None)
            logging.info("Predicting " + chrom)

            def feed_votes(res):
                classes, probs = hot.fetch_predictions(res)
                voter.next_sample = res.sample                    # a site is written on the records of the window that opened it
                voter.feed_batch([ln.label() for ln in res.lines], classes, probs)

            prev = None
            for part, (start, end) in enumerate(tasks[chrom]):
                # waits for the window's records (decoded ahead; the device engine hands a chromosome over in slices of windows)
                _key, hot.sample = feed.get(chrom, block=True, start=start)
                try:
                    cur = hot.collect(chrom, start, end, rescan=False)
                except Exception:                                 # run_collection.py:44-47: the window yields nothing
                    _t, value, trace = sys.exc_info()
                    logging.error("%s:%s-%s [ERROR]: %s. Locate At: %s", chrom, start, end, value, traceback.extract_tb(trace))
                    cur = hot.empty(chrom, start, end)
                cur.sample = hot.sample
                text = "".join(ln.text() for ln in cur.lines)
                with open(os.path.join(seg_dir, "%s.segments.%d.bed" % (chrom, part)), "w") as f:
                    f.write(text)
                all_bed.write(text)
                if prev is not None:
                    feed_votes(prev)
                prev = hot.launch(cur)
            if prev is not None:
                feed_votes(prev)
            voter.finish()
        feed.release(chrom)
    if os.environ.get("SVX_TIMING"):
        print("network + graphs %.3f, windows %.3f" % (_t1 - _t0, _time.time() - _t1), flush=True)


def _run_pooled(options, feed, tasks, chroms, seg_dir, pred_dir, pool=None):
    """``-t N`` (the reference's process-pool size, SVision:261,311): N forked helper processes run the collection and
    the vote of whole windows while this process feeds the device (pipeline.PooledHotPath); windows complete in any
    order; a chromosome is stitched and written (in task order inside it) as soon as its last window is done, so the
    files are those of the one-process path and only the chromosomes in flight are resident."""
    from .network.predict import load_network
    from .pipeline import PooledHotPath, stitch_windows
    net = load_network(options.model_path)
    windows = [(chrom, start, end) for chrom in chroms for start, end in tasks[chrom]]
    first = {}
    for wid, (chrom, _s, _e) in enumerate(windows):
        first.setdefault(chrom, wid)
    left = {chrom: len(tasks[chrom]) for chrom in chroms}
    import time as _time
    _t0 = _time.time()
    static = getattr(feed, "sample", None)
    hot = PooledHotPath(static, options, net, n_workers=options.thread_num, n_streams=3, max_inflight=6, want_tsv=True, pool=pool, feed=feed, lazy_graphs=True)
    _t1 = _time.time()
    done = {}

    def write_chromosome(chrom):
        wids = range(first[chrom], first[chrom] + len(tasks[chrom]))
        texts = stitch_windows([done[w] for w in wids], options,               # per-chromosome vote: edge sites written once
                               lambda c, start: feed.get(c, block=True, start=start)[1])
        vcf_text, score_text = texts.get(chrom, ("", ""))
        prefix = os.path.join(pred_dir, "%s.predict.s%s" % (chrom, options.min_support))
        logging.info("Predicting " + chrom)
        with open(prefix + ".score.txt", "w") as score_out, open(prefix + ".vcf", "w") as vcf_out, \
                open(os.path.join(seg_dir, chrom + ".segments.all.bed"), "w") as all_bed:
            for part, w in enumerate(wids):
                with open(os.path.join(seg_dir, "%s.segments.%d.bed" % (chrom, part)), "w") as f:
                    f.write(done[w].tsv)
                all_bed.write(done[w].tsv)
                del done[w]
            vcf_out.write(vcf_text)
            score_out.write(score_text)
        hot.release(chrom)

    try:
        for res in hot.run_windows(windows, rescan=False):
            done[res.wid] = res
            left[res.chrom] -= 1
            if left[res.chrom] == 0:
                write_chromosome(res.chrom)
    finally:
        _t2 = _time.time()
        hot.close()
    for chrom in chroms:                                          # chromosomes without a window (cannot happen) or never reached
        if left[chrom] and not os.path.exists(os.path.join(pred_dir, "%s.predict.s%s.vcf" % (chrom, options.min_support))):
            raise RuntimeError("chromosome %s was not completed" % chrom)
    if os.environ.get("SVX_TIMING"):
        print("pool up %.3f, windows %.3f, close %.3f, owner %s" % (_t1 - _t0, _t2 - _t1, _time.time() - _t2,
              {k: round(v, 3) for k, v in getattr(hot, "owner_profile", {}).items()}), flush=True)


def _scores_of(pred_dir, chroms, options):
    """Scores of this rank's chromosomes only (a shared out_path may hold other ranks' files)."""
    scores = []
    for chrom in chroms:
        path = os.path.join(pred_dir, "%s.predict.s%s.score.txt" % (chrom, options.min_support))
        if os.path.exists(path):
            with open(path) as f:
                scores += [float(l.strip()) for l in f if l.strip() != "0"]
    return scores


def main(arguments=None):
    run(parse_arguments(arguments))


if __name__ == "__main__":
    main()
