"""Shared test helpers (CPU): options namespace, golden sample loading with the ORACLE scan
injected (the product path uses the GPU scan; CPU tests of the host logic inject the oracle's)."""
import gzip
import os
import types

import numpy as np

from svision_amd.io import bam
from svision_amd.sample import Sample

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def default_options(**over):
    """CLI defaults of the reference (SVision:27-106)."""
    o = types.SimpleNamespace(
        out_path=None, bam_path=None, model_path=None, genome=None, sample="sample", thread_num=1, min_support=5,
        chrom=None, hash=False, qname=False, graph=False, contig=False, debug=False, min_mapq=10, min_sv_size=50,
        max_sv_size=1000000, window_size=10000000, patition_max_distance=5000, cluster_max_distance=0.3,
        batch_size=128, min_gt_depth=4, homo_thresh=0.8, hete_thresh=0.2, k_size=10, min_accept=50, max_hash_len=1000)
    for k, v in over.items():
        setattr(o, k, v)
    return o


def load_golden_fasta(name="collect_small.fa.gz"):
    seqs, cur = {}, None
    with gzip.open(os.path.join(GOLDEN, name), "rb") as f:
        for line in f:
            if line.startswith(b">"):
                cur = line[1:].strip().decode()
            else:
                seqs[cur] = line.strip()
    return bam.Fasta(sequences=seqs)


def oracle_scan(table, min_sv):
    from oracle import cbind
    return cbind.cigar_scan(table.cigar, table.cig_off.astype(np.uint64), table.pos, min_sv)


def golden_sample(min_sv=50, device=None, name="collect_small"):
    table = bam.read_bam(os.path.join(GOLDEN, name + ".bam"))
    fasta = load_golden_fasta(name + ".fa.gz")
    if device is None:
        return Sample.with_scan(table, fasta, min_sv, oracle_scan(table, min_sv))
    return Sample.from_table(table, fasta, min_sv, device)
