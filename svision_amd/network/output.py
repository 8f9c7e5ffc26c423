"""Per-site VCF records, scores, and the merged VCF.

Mirror of the live part of the reference's ``src/network/output.py``:
``write_results_to_vcf`` :469-598, ``refine_type`` :352-467, ``cal_scores_max_min``
:601-612, ``merge_split_vcfs`` :251-348.  The score keeps the reference's mixed
float32/float64 NumPy arithmetic so that QUAL strings come out identical
for identical CNN outputs.
"""
import collections
import os

import numpy as np

from .. import __version__
from .genotype import genotyper


def refine_type(types, bkps, options):
    """Reconcile INS with DUP/tDUP lengths; DUP next to the insertion point becomes tDUP (:352-467)."""
    has_ins, has_dup, has_tdup = "INS" in types, "DUP" in types, "tDUP" in types
    if not (has_ins and (has_dup or has_tdup)):
        return types, bkps
    ins_len = dup_len = 0
    ins_pos = -1
    for i, t in enumerate(types):
        if t == "INS":
            ins_pos = int(bkps[i][0])
            ins_len += int(bkps[i][2])
        elif t == "tDUP":
            dup_len += int(bkps[i][2])
        elif t == "DUP":
            dup_len += int(bkps[i][2])
            if ins_pos != -1 and abs(ins_pos - int(bkps[i][1])) < 10:
                types[i] = "tDUP"
    if ins_len - dup_len > options.min_sv_size:
        for i, t in enumerate(types):
            if t == "INS":
                bkps[i][2] = ins_len - dup_len
                break
        return types, bkps
    keep = [i for i, t in enumerate(types) if t != "INS"]
    return [types[i] for i in keep], [bkps[i] for i in keep]


def write_results_to_vcf(vcf_out, score_out, region_svtypes, region, read_names, sig_types, sig_scores,
                         predict_scores, sig_mechanisms, options, sample):
    """One VCF body line (+ one score line) per supported type combination of a site (:469-598)."""
    if len(region_svtypes) == 0:
        return
    mean_score = np.mean(predict_scores)                      # float32 when the scores are float32
    avg_predict_score = (1 - round(mean_score, 2)) * 100
    chrom, start, end = region.split("+")[:3]
    start, end = int(start), int(end)
    kept = [sv for sv in region_svtypes if len(sv[1]) >= options.min_support]
    stat = collections.Counter(sig_types)
    flt = "Uncovered" if "sigUncovered" in stat and stat["sigUncovered"] >= 0.75 * len(sig_types) else "PASS"
    for sv_type, reads, bkps in kept:
        support = len(reads)
        names = [read_names[r] for r in reads]
        score_std = np.std([int(sig_scores[r]) for r in reads]) / int(str(support))
        sum_score = min(100, (score_std + avg_predict_score))
        types, bkps = refine_type(sv_type.split("+"), bkps, options)
        bk = ",".join("%s:%s-%s-%s" % (t, b[2], b[0], b[1]) for t, b in zip(types, bkps))
        info = "END=%d;SVLEN=%d;SVTYPE=%s;SUPPORT=%d;BKPS=%s" % (end, end - start, "+".join(types), support, bk)
        if options.qname:
            info += ";READS=" + ",".join(names)
        gt, dr, dv = genotyper((chrom, start, end, types), names, options, sample)
        alt = "<CSV>" if len(types) >= 2 else "<SV>"
        line = "\t".join([chrom, str(start), "0", "N", alt, str(sum_score), flt, info, "GT:DR:DV\t%s:%s:%s" % (gt, dr, dv)])
        print(sum_score, file=score_out)
        print(line, file=vcf_out)


def cal_scores_max_min(predict_path):
    """All per-record scores of a run, lines equal to '0' excluded (:601-612)."""
    scores = []
    for name in os.listdir(predict_path):
        if "score.txt" in name:
            with open(os.path.join(predict_path, name)) as f:
                for line in f:
                    if line.strip() == "0":
                        continue
                    scores.append(float(line.strip()))
    return scores


VCF_HEADER = """##CHROM=<CHROM=XXX,Description="Chromosome ID">
##POS=<POS=XXX,Description="Start position of the SV described in this region">
##ID=<ID=XXX,Description="ID of the SV described in this region">
##REF=<REF=N,Description="Ref's sequence in that region, default=N">
##QUAL=<QUAL=XXX,Description="The SV quality of the SV described in this region">
##ALT=<ID=SV,Description="Simple SVs">
##ALT=<ID=CSV,Description="Complex or nested SVs">
##FILTER=<ID=Covered,Description="Covered mean the SV is spanned by reads">
##FILTER=<ID=Uncovered,Description="UnCovered mean the SV is not spanned by reads">
##FILTER=<ID=Clustered,Description="Clustered mean the SV is not spanned by reads, but can be cluster together with others">
##INFO=<ID=END,Number=1,Type=Integer,Description="End position of the SV described in this region">
##INFO=<ID=SVLEN,Number=1,Type=Integer,Description="Difference in length between REF and ALT alleles">
##INFO=<ID=BKPS,Number=.,Type=String,Description="All breakpoints (length-start-end) in this region, where CSV might contain multiple breakpoints.">
##INFO=<ID=SVTYPE,Number=1,Type=String,Description="CNN predicted SV type, containing INS, DEL, DUP, tDUP (tandem duplication) and INV">
##INFO=<ID=SUPPORT,Number=1,Type=Integer,Description="SV support number in this region">
##INFO=<ID=READS,Number=.,Type=String,Description="SV support read names in this region">
"""
VCF_GRAPH_INFO = """##INFO=<ID=GraphID,Number=1,Type=String,Description="The corresponding graph id of isomorphic CSV graph structures">
##INFO=<ID=GFA_FILE_PREFIX,Number=1,Type=String,Description="File name of CSV corresponding GFA file">
##INFO=<ID=GFA_S,Number=1,Type=String,Description="Nodes contained in a CSV graph represented based on GFA format">
##INFO=<ID=GFA_L,Number=1,Type=String,Description="Links contained in a CSV graph represented based on GFA format">
"""                                                             # --graph only (:292-296)
VCF_FORMAT = """##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">
##FORMAT=<ID=DR,Number=1,Type=Integer,Description="high-quality reference reads">
##FORMAT=<ID=DV,Number=1,Type=Integer,Description="high-quality variant reads">
"""


def rescale_records(records, max_score, min_score, id_start=-1):
    """ID assignment + QUAL rescale of body lines (:305-344).  Returns (lines, last_id)."""
    out = []
    id_num = id_start
    prev_start, prev_end, sub = 0, 1, 1
    for rec in records:
        c = str(rec).split("\t")
        start, end = c[1], c[7].split(";")[0][4:]
        if start == prev_start and end == prev_end:
            c[2] = "%d_%d" % (id_num, sub)
            sub += 1
        else:
            prev_start, prev_end = start, end
            id_num += 1
            sub = 1
            c[2] = str(id_num)
        old = float(c[5])
        new = 100
        if max_score != min_score:
            new = int(100 - (round((old - min_score) / (max_score - min_score), 2) * 100))
        c[5] = str(new)
        out.append("\t".join(c))
    return out, id_num, (prev_start, prev_end, sub)


def merge_split_vcfs(in_dir, merged_vcf_path, max_score, min_score, spec_chroms, options, fasta=None):
    """Header + per-chromosome body files -> final VCF (:251-348)."""
    if fasta is None:
        from ..io.bam import Fasta
        fasta = Fasta(options.genome)
    with open(merged_vcf_path, "w") as out:
        out.write("##fileformat=VCFv4.3\n##source=SVision v%s\n" % getattr(options, "source_version", __version__))
        for chrom in fasta.references:
            out.write("##contig=<ID=%s,length=%d>\n" % (chrom, fasta.get_reference_length(chrom)))
        out.write(VCF_HEADER)
        if getattr(options, "graph", False):
            out.write(VCF_GRAPH_INFO)
        out.write(VCF_FORMAT)
        out.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s\n" % options.sample)
        id_num = -1
        for chrom in spec_chroms:
            with open(os.path.join(in_dir, "%s.predict.s%s.vcf" % (chrom, options.min_support))) as f:
                lines, id_num, _ = rescale_records(f.readlines(), max_score, min_score, id_num)
            out.writelines(lines)
