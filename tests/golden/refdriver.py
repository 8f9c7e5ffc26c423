"""Drives the REFERENCE implementation (imported from /root/reference, this container only)
with in-memory stand-ins for the third-party modules it needs (pysam, cv2, tensorflow, bs4),
so that golden input/output vectors can be generated.  Nothing here is shipped or imported
by the product; the fixtures it produces are plain data (tests/golden/*.json|tsv|bam).

Stand-in semantics (third-party behaviour the reference relies on, SURVEY 8(a')):
  * pysam: SAM-spec field derivations (reference_end, query_alignment_start/end,
    query_length), fetch = records overlapping [start, end) in file order;
  * cv2.line: oracle.encode_ref.cv_line (OpenCV clipLine + LineIterator restatement);
    cv2.resize to the same size: identity;
  * tensorflow: a permissive stub whose Session.run returns injected predictions.
"""
import os
import sys
import types

REF_ROOT = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import cigar_ref, encode_ref  # noqa: E402

DATASETS = {}      # path -> dict(table=AlignmentTable, )
FASTAS = {}        # path -> {name: bytes}


class AlignedSegment:
    """Attribute bag with pysam's derived properties."""

    def __init__(self):
        self.reference_id = -1
        self.reference_start = 0
        self.query_name = None
        self.is_supplementary = False
        self.is_reverse = False
        self.is_unmapped = False
        self.is_secondary = False
        self.query_sequence = None
        self.mapping_quality = 0
        self.cigarstring = None
        self.reference_name = None

    @property
    def mapq(self):
        return self.mapping_quality

    @property
    def qname(self):
        return self.query_name

    def _ops(self):
        return cigar_ref.parse_cigar(self.cigarstring)

    @property
    def reference_end(self):
        return self.reference_start + cigar_ref.alignment_stats(self._ops())[0]

    @property
    def query_length(self):
        return len(self.query_sequence) if self.query_sequence is not None else 0

    @property
    def query_alignment_start(self):
        # leading soft clips (hard clips were rewritten to S by the reference before this is read)
        lead = 0
        for o, n in self._ops():
            if o == 4:
                lead += n
            elif o == 5:
                continue
            else:
                break
        return lead

    @property
    def query_alignment_end(self):
        end = self.query_length
        for o, n in reversed(self._ops()):
            if o == 4:
                end -= n
            elif o == 5:
                continue
            else:
                break
        return end


class _LazySeq(str):
    """A query sequence whose content is irrelevant on the default path: only its length and
    slices are used (analyze_reads.py:667).  Real str of N's."""


def _record_from_table(table, i):
    a = AlignedSegment()
    a.reference_id = int(table.tid[i])
    a.reference_name = table.references[a.reference_id] if a.reference_id >= 0 else None
    a.reference_start = int(table.pos[i])
    a.query_name = table.names[int(table.name_id[i])]
    flag = int(table.flag[i])
    a.is_reverse = bool(flag & 0x10)
    a.is_supplementary = bool(flag & 0x800)
    a.is_secondary = bool(flag & 0x100)
    a.is_unmapped = bool(flag & 0x4)
    a.mapping_quality = int(table.mapq[i])
    words = table.cigar[int(table.cig_off[i]):int(table.cig_off[i + 1])]
    a.cigarstring = "".join("%d%s" % (int(w) >> 4, cigar_ref.OPS[int(w) & 15]) for w in words) if len(words) else None
    l_seq = int(table.l_seq[i])
    seq = table.query_sequence(i) if getattr(table, "seq_packed", None) is not None else None
    a.query_sequence = seq if seq is not None else (("N" * l_seq) if l_seq > 0 else None)
    return a


class AlignmentFile:
    def __init__(self, path, mode="r"):
        self.table = DATASETS[path]
        self.header = {"HD": {"SO": "coordinate"}}

    def _span(self, i):
        t = self.table
        words = t.cigar[int(t.cig_off[i]):int(t.cig_off[i + 1])]
        ops = [(int(w) & 15, int(w) >> 4) for w in words]
        span = cigar_ref.alignment_stats(ops)[0]
        return span if span > 0 and not (int(t.flag[i]) & 0x4) else 1

    def fetch(self, contig=None, start=None, stop=None, end=None):
        if stop is None:
            stop = end
        t = self.table
        tid = t.references.index(contig)
        start, stop = int(start), int(stop)            # old pysam truncates float coordinates
        for i in range(len(t)):
            if int(t.tid[i]) != tid:
                continue
            p = int(t.pos[i])
            if p >= stop:
                break
            if p + self._span(i) > start:
                yield _record_from_table(t, i)

    def get_tid(self, name):
        return self.table.references.index(name) if name in self.table.references else -1

    def getrname(self, tid):
        return self.table.references[tid]

    def get_reference_length(self, name):
        return self.table.lengths[self.table.references.index(name)]

    def check_index(self):
        return True

    def get_index_statistics(self):
        return [(r, 0, 0, 0) for r in self.table.references]


class FastaFile:
    def __init__(self, path):
        self._seqs = FASTAS[path]
        self.references = list(self._seqs)

    def fetch(self, chrom, start, end):
        s = self._seqs[chrom]
        if start < 0 or end < start:
            raise ValueError("invalid coordinates")
        return s[start:min(end, len(s))].decode()

    def get_reference_length(self, chrom):
        return len(self._seqs[chrom])


class _VariantRecord:
    """One body line of a VCF as pysam.VariantRecord shows it to graph.py:545-582."""

    def __init__(self, line):
        self._line = line.rstrip("\n")
        c = self._line.split("\t")
        self.contig = c[0]
        self.start = int(c[1]) - 1
        info = dict(kv.split("=", 1) for kv in c[7].split(";") if "=" in kv)
        self.stop = int(info["END"])                           # htslib: rlen from INFO/END
        self.info = {"SVTYPE": info["SVTYPE"], "SUPPORT": int(info["SUPPORT"])}
        if "READS" in info:                                    # written with --qname only (output.py:580)
            self.info["READS"] = tuple(info["READS"].split(","))

    def __str__(self):
        return self._line + "\n"                               # htslib prints back the fields it parsed, in their order


class _VariantHeader:
    def __init__(self, lines):
        self._lines = list(lines)
        if not any(l.startswith("##FILTER=<ID=PASS,") for l in self._lines):    # htslib defines PASS right after ##fileformat
            self._lines.insert(1, '##FILTER=<ID=PASS,Description="All filters passed">\n')

    def __str__(self):
        return "".join(self._lines)


class VariantFile:
    """pysam.VariantFile(path) for reading: .header (str()) and iteration over records.  Assumed third-party
    behaviour (pysam / htslib are not in this image): the text of header and records round-trips unchanged -- true for
    the reference's own VCF writer, whose INFO / FORMAT fields are all declared -- plus the PASS filter line."""

    def __init__(self, path, mode="r"):
        with open(path) as f:
            lines = f.readlines()
        self.header = _VariantHeader([l for l in lines if l.startswith("#")])
        self._records = [_VariantRecord(l) for l in lines if not l.startswith("#") and l.strip()]

    def __iter__(self):
        return iter(self._records)


class _Shape(list):
    def as_list(self):
        return list(self)


class _Anything:
    """Permissive stub: any attribute is a callable returning another stub."""

    def __getattr__(self, name):
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()

    def __enter__(self):
        return self

    def __rsub__(self, other):
        return _Anything()

    def __hash__(self):
        return id(self)

    def __iter__(self):                      # tf.split(...) results are zipped pairwise
        return iter([_Anything(), _Anything()])

    def __exit__(self, *a):
        return False

    def get_shape(self):
        return _Shape([1, 1, 1, 4])

    def as_list(self):
        return [1]


PREDICTOR = {"fn": None}     # fn(batch_images) -> (logits, argmax, softmax)


class _Session(_Anything):
    def run(self, fetches, feed_dict=None):
        if isinstance(fetches, list) and len(fetches) == 3:
            batch = [v for v in feed_dict.values() if hasattr(v, "shape") and getattr(v, "ndim", 0) == 4][0]
            return PREDICTOR["fn"](batch)
        return None


def install_stubs():
    pysam = types.ModuleType("pysam")
    pysam.AlignedSegment = AlignedSegment
    pysam.AlignmentFile = AlignmentFile
    pysam.FastaFile = FastaFile
    pysam.VariantFile = VariantFile
    cv2 = types.ModuleType("cv2")
    cv2.line = lambda img, p1, p2, color, thickness=1: encode_ref.cv_line(img, p1, p2, color)
    cv2.resize = lambda img, size: img
    cv2.flip = lambda img, code: img[:, ::-1]
    cv2.imwrite = lambda *a, **k: True
    tf = _Anything()
    tfm = types.ModuleType("tensorflow")
    tfm.__getattr__ = lambda name: getattr(tf, name)      # module-level __getattr__ (PEP 562)
    compat = _Anything()
    v1 = _Anything()
    v1.Session = lambda *a, **k: _Session()
    compat.v1 = v1
    tfm.compat = compat
    tfm.float32 = "float32"
    bs4 = types.ModuleType("bs4")
    bs4.BeautifulSoup = object
    bs4.__path__ = []
    bs4_element = types.ModuleType("bs4.element")
    bs4_element.NavigableString = str
    bs4.element = bs4_element
    sys.modules.update({"pysam": pysam, "cv2": cv2, "tensorflow": tfm, "bs4": bs4, "bs4.element": bs4_element})
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import warnings
    warnings.simplefilter("ignore")


def default_options(**over):
    """The reference CLI defaults (SVision:27-106)."""
    o = types.SimpleNamespace(
        out_path=None, bam_path=None, model_path=None, genome=None, sample="sample", thread_num=1, min_support=5,
        chrom=None, hash=False, qname=False, graph=False, contig=False, debug=False, min_mapq=10, min_sv_size=50,
        max_sv_size=1000000, window_size=10000000, patition_max_distance=5000, cluster_max_distance=0.3,
        batch_size=128, min_gt_depth=4, homo_thresh=0.8, hete_thresh=0.2, k_size=10, min_accept=50, max_hash_len=1000)
    for k, v in over.items():
        setattr(o, k, v)
    return o
