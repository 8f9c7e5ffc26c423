#!/usr/bin/env python3
"""Generates tests/golden/hash_small.expected.json by calling the REFERENCE's hashplot_unmapped
(/root/reference/src/segmentplot/run_hash_lineplot.py:52; pure Python, imported unmodified) on
random windows / pieces: novel insertions, forward and reverse-complemented copies of window
pieces (with and without a mismatch), tandem repeats, N runs, lower-case bases."""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
from src.segmentplot.run_hash_lineplot import hashplot_unmapped  # noqa: E402  (reference)

rnd = random.Random(99)
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def rs(n):
    return "".join(rnd.choice("ACGT") for _ in range(n))


def rc(s):
    return "".join(COMP.get(c, "N") for c in reversed(s))


def mutate(s, n):
    s = list(s)
    for _ in range(n):
        p = rnd.randrange(len(s))
        s[p] = rnd.choice([c for c in "ACGT" if c != s[p]])
    return "".join(s)


def main():
    cases = []
    for _ in range(220):
        ref = rs(rnd.randint(300, 2500))
        kind = rnd.randrange(8)
        if kind == 0:
            seq = rs(rnd.randint(30, 900))
        elif kind in (1, 2):
            a = rnd.randrange(0, len(ref) - 120); n = rnd.randint(60, min(800, len(ref) - a))
            seq = ref[a:a + n] if kind == 1 else rc(ref[a:a + n])
        elif kind == 3:
            a = rnd.randrange(0, len(ref) - 200); n = rnd.randint(120, min(700, len(ref) - a))
            seq = mutate(ref[a:a + n], rnd.randint(1, 3))
        elif kind == 4:
            a = rnd.randrange(0, len(ref) - 150); n = rnd.randint(80, min(400, len(ref) - a))
            seq = rs(rnd.randint(0, 80)) + ref[a:a + n] + rs(rnd.randint(0, 80)) + rc(ref[a:a + n // 2])
        elif kind == 5:
            unit = rs(rnd.randint(12, 60)); ref = rs(200) + unit * rnd.randint(3, 8) + rs(300)
            seq = unit * rnd.randint(2, 6)
        elif kind == 6:
            a = rnd.randrange(0, len(ref) - 200); n = rnd.randint(100, min(500, len(ref) - a))
            piece = list(ref[a:a + n]); piece[rnd.randrange(n)] = "N"; seq = "".join(piece)
        else:
            a = rnd.randrange(0, len(ref) - 200); n = rnd.randint(100, min(500, len(ref) - a))
            seq = ref[a:a + n].lower() if rnd.random() < 0.5 else ref[a:a + n // 2] + ref[a + n // 2:a + n].lower()
        _m, segs = hashplot_unmapped(ref, seq, 10, 50)
        cases.append({"ref": ref, "seq": seq, "segs": [[s.xStart(), s.xEnd(), s.yStart(), s.yEnd(), bool(s.forward())] for s in segs]})
    with open(os.path.join(HERE, "hash_small.expected.json"), "w") as f:
        json.dump(cases, f)
    print("cases", len(cases), "with hits", sum(1 for c in cases if c["segs"]), "multi", sum(1 for c in cases if len(c["segs"]) > 1))


if __name__ == "__main__":
    main()
