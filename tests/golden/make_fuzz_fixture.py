#!/usr/bin/env python3
"""Generates tests/golden/fuzz_small.expected.json: randomised unit cases pushed through the
REFERENCE's functions (this container only):
  * analyze_gap      /root/reference/src/collection/analyze_reads.py:155   (all branches, with helpers)
  * analyze_inside_align + cigar_to_list                                   :804, collect_signatures.py:27
  * refine_type      /root/reference/src/network/output.py:352
  * get_region_potential_svtypes  /root/reference/src/network/predict.py:29
  * linearOrNot / cal_non_linear  /root/reference/src/collection/output_clusters.py:11,218
"""
import copy
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refdriver  # noqa: E402

refdriver.install_stubs()
from src.collection import analyze_reads as ref_ar  # noqa: E402
from src.collection.collect_signatures import cigar_to_list  # noqa: E402
from src.collection import output_clusters as ref_oc  # noqa: E402
from src.network.output import refine_type as ref_refine  # noqa: E402
from src.network.predict import Predict as RefPredict  # noqa: E402
from src.segmentplot.classes import Segment as RefSegment  # noqa: E402

rnd = random.Random(1234)
REF = "".join(rnd.choice("ACGT") for _ in range(3000)) + "ACGT" * 500 + "".join(rnd.choice("ACGT") for _ in range(3000)) + "A" * 400 + "".join(rnd.choice("AC") for _ in range(2600))
REF_START = 10_000


class FakeBam:
    def getrname(self, tid):
        return "chr%d" % tid


def seg(q0, q1, r0, r1, rev=False, typ="main", ref_id=0):
    return {"q_start": q0, "q_end": q1, "qual": 60, "ref_id": ref_id, "ref_chr": "chr%d" % ref_id, "ref_start": r0, "ref_end": r1,
            "read_name": "r", "cigarstring": "", "read_seq": "", "is_supplementary": False, "is_reverse": rev, "type": typ}


def rand_gap_case():
    base_r = REF_START + rnd.randint(200, 6000)
    l1 = rnd.randint(60, 3000)
    cur = seg(rnd.randint(0, 500), 0, base_r, base_r + l1, rev=rnd.random() < 0.2)
    cur["q_end"] = cur["q_start"] + l1 + rnd.randint(-5, 5)
    d_read = rnd.choice([0, 1, -10, -30, 50, 300, 2000, rnd.randint(-60, 4000)])
    d_ref = rnd.choice([0, 1, -10, -49, -50, -51, -500, 60, 300, 2000, rnd.randint(-2500, 2500)])
    l2 = rnd.randint(60, 3000)
    nxt = seg(cur["q_end"] + d_read, 0, cur["ref_end"] + d_ref, 0, rev=cur["is_reverse"] if rnd.random() < 0.8 else not cur["is_reverse"],
              ref_id=0 if rnd.random() < 0.95 else 1)
    nxt["q_end"] = nxt["q_start"] + l2
    nxt["ref_end"] = nxt["ref_start"] + l2 + rnd.randint(-5, 5)
    helps = []
    for _ in range(rnd.choice([0, 0, 0, 1, 1, 2])):
        hl = rnd.randint(30, 1500)
        hq = cur["q_end"] + rnd.randint(0, max(1, abs(d_read)))
        hr = REF_START + rnd.randint(100, 11000)
        helps.append(seg(hq, hq + hl, hr, hr + hl, rev=rnd.random() < 0.4, typ="other"))
    helps.sort(key=lambda s: (s["q_start"], s["q_end"]))
    return cur, nxt, helps


def main():
    out = {"ref": REF, "ref_start": REF_START, "gap": [], "inside": [], "refine": [], "vote": [], "linear": []}
    opts = refdriver.default_options()
    ref_ar.fetch_ref_seq = lambda path, chrom, start, end: REF[max(0, start - REF_START):max(0, end - REF_START)]
    n_sig = 0
    for _ in range(600):
        cur, nxt, helps = rand_gap_case()
        inp = [copy.deepcopy(cur), copy.deepcopy(nxt), copy.deepcopy(helps)]
        c, n, h = copy.deepcopy(cur), copy.deepcopy(nxt), copy.deepcopy(helps)
        try:
            sig = ref_ar.analyze_gap(c, n, FakeBam(), opts, h) if h else ref_ar.analyze_gap(c, n, FakeBam(), opts)
            err = None
        except Exception as e:      # noqa: BLE001
            sig, err = None, type(e).__name__
        exp = None
        if sig is not None:
            n_sig += 1
            exp = [sig.type, sig.tstart, sig.tend, sig.bkps, sig.mechanism,
                   [[a["q_start"], a["q_end"], a["ref_start"], a["ref_end"], bool(a["is_reverse"])] for a in sig.sorted_aligns]]
        keep = lambda s: [s["q_start"], s["q_end"], s["ref_start"], s["ref_end"], bool(s["is_reverse"]), s["ref_id"]]
        out["gap"].append({"cur": keep(inp[0]), "nxt": keep(inp[1]), "help": [keep(x) for x in inp[2]], "sig": exp, "err": err,
                           "help_after": [[x["ref_start"], x["ref_end"]] for x in h]})
    print("analyze_gap cases", len(out["gap"]), "signatures", n_sig)
    for _ in range(150):
        ops = []
        for _k in range(rnd.randint(1, 14)):
            ops.append("%d%s" % (rnd.choice([1, 5, 49, 50, 51, 200, 1500]), rnd.choice("MMM=XIDIDNSHP")))
        cigar = ("%dS" % rnd.randint(1, 500) if rnd.random() < 0.5 else "") + "".join(ops)
        o, l = cigar_to_list(cigar.replace("H", "S"))
        span = sum(n for c, n in zip(o, l) if c in "MD=XN")
        sd = seg(rnd.randint(0, 900), 0, 5000, 5000 + span)
        sd["q_end"] = sd["q_start"] + sum(n for c, n in zip(o, l) if c in "MI=X")
        major, minor = ref_ar.analyze_inside_align(sd, o, l, opts)
        exp = None if major is None else [[m["q_start"], m["q_end"], m["ref_start"], m["ref_end"]] for m in major]
        out["inside"].append({"cigar": cigar, "q_start": sd["q_start"], "ref_start": 5000, "ref_end": 5000 + span, "segs": exp})
    types_pool = ["DEL", "INS", "INV", "DUP", "tDUP"]
    for _ in range(200):
        k = rnd.randint(1, 4)
        types = rnd.sample(types_pool, k)
        bkps = [[rnd.randint(1000, 1100), rnd.randint(1000, 1200), rnd.randint(40, 400)] for _ in range(k)]
        t_in, b_in = copy.deepcopy(types), copy.deepcopy(bkps)
        t, b = ref_refine(types, bkps, opts)
        out["refine"].append({"types": t_in, "bkps": b_in, "out_types": list(t), "out_bkps": [list(x) for x in b]})
    pred = RefPredict("chr", "none")
    for _ in range(100):
        reads = {}
        for r in range(rnd.randint(1, 12)):
            infos = {}
            for c in rnd.sample(range(5), rnd.randint(1, 3)):
                infos[c] = [rnd.randint(1000, 1100), rnd.randint(1100, 1200), rnd.randint(50, 500)]
            reads[str(r + 1)] = infos
        got = pred.get_region_potential_svtypes(copy.deepcopy(reads))
        out["vote"].append({"reads": {k: {str(c): v for c, v in d.items()} for k, d in reads.items()},
                            "out": [[t, list(ids), [list(x) for x in bk]] for t, ids, bk in got]})
    for _ in range(300):
        a = RefSegment(rnd.randint(0, 3000), rnd.randint(0, 3000), rnd.randint(1, 2000), rnd.random() < 0.7, 0)
        b = RefSegment(rnd.randint(0, 6000), rnd.randint(0, 6000), rnd.randint(1, 2000), rnd.random() < 0.7, 0)
        score = ref_oc.cal_non_linear([a, b])
        out["linear"].append({"a": [a.xStart(), a.yStart(), a._length, a.forward()], "b": [b.xStart(), b.yStart(), b._length, b.forward()],
                              "linear": ref_oc.linearOrNot(a, b), "score": score})
    with open(os.path.join(HERE, "fuzz_small.expected.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
