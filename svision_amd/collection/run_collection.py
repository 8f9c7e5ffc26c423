"""One collection task = one (chromosome, window): fetch -> signatures -> clusters -> TSV.

Host-side mirror of the reference's ``run_detect`` (src/collection/run_collection.py:15-47),
same arguments, same output file (``segments/{chrom}.segments.{part}.bed``), same error
behaviour: any exception is turned into an ``"[ERROR]: ..."`` string return value.
"""
import logging
import sys
import traceback

from .. import sample as _sample
from .cluster_signatures import partition_and_cluster
from .collect_signatures import analyze_alignments
from .output_clusters import writer_cluster_to_file


def detect_window(options, sample, chrom, start, end, part_num=0):
    """Signatures and clusters of one window (no file output)."""
    tid = sample.table.get_tid(chrom)
    rows = sample.table.fetch(tid, start, end)
    signatures = analyze_alignments(rows, sample, options, part_num)
    clusters = partition_and_cluster(signatures, chrom, sample, options)
    return signatures, clusters


def run_detect(options, sample_path, chrom, part_num, start, end):
    try:
        sample = _sample.resolve(sample_path, options)
        signatures, clusters = detect_window(options, sample, chrom, start, end, part_num)
        logging.info("Processing %s:%s-%s, %d segments write to: %s.segments.%s.bed", chrom, start, end,
                     len(signatures), chrom, part_num)
        writer_cluster_to_file(clusters, chrom, part_num, options)
        return None
    except Exception:
        _t, value, trace = sys.exc_info()
        return "[ERROR]: " + str(value) + ". Locate At: " + str(traceback.extract_tb(trace))
