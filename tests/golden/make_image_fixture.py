#!/usr/bin/env python3
"""Generates tests/golden/image_small.expected.json.gz by running the REFERENCE's
``BatchGenerator.next_batch`` (/root/reference/src/network/create_batch.py:88-155), which drives
``Segment.__init__`` (src/segmentplot/classes.py:42-54) and ``PlotSingleImg.plot``
(src/segmentplot/plot_segment.py:8-73), on

  (i)  every distinct segment-TSV line of the golden collection fixtures
       (collect_small / modes / hash_collect: HiFi-like, ONT-like, --contig, --hash), and
  (ii) hostile hand-made and seeded random lines: end points far outside [0,226]^2, reverse
       segments running off the image, max(read_len, ref_len) < 227 (ratio clamps to 1), zero /
       one / negative lengths, a strand column that is neither 'True' nor 'False', the pad record.

The fixture stores the TSV lines (inputs) and, per image and channel, the flat indices
row * 227 + col of the pixels that are 255 (expected output); the script asserts that the
reference returns nothing but {-mean, 255-mean}.  ``cv2.line`` is third-party arithmetic the
reference does not carry: the stand-in is oracle.encode_ref.cv_line (OpenCV clipLine +
LineIterator restated), so this fixture pins everything the reference itself does around it —
segment rebuilding, ratio and truncation, argument order, channel 1's column rule, mean
subtraction, padding — and leaves cv2.line "parity unpinned" (DESIGN.md section 3).
Run in this container only; the output is byte-reproducible (gzip mtime 0, sorted keys)."""
import gzip
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refdriver  # noqa: E402

refdriver.install_stubs()
from tests import datagen  # noqa: E402

from src.network.create_batch import BatchGenerator as RefBatch  # noqa: E402  (reference)

MEAN = np.array([104., 117., 124.])
BATCH = 64


def _line(data, k, strand=("True", "False")):
    """A 23-column TSV line around 12 data values (the other columns never reach the image)."""
    d = list(data)
    f = ["chrH+%d+%d+9" % (k, k + 1)] + [str(int(v)) for v in d[:4]] + [strand[0] if d[4] == 1 else strand[1] if d[4] == 0 else str(d[4])]
    f += [str(int(v)) for v in d[5:9]] + [strand[0] if d[9] == 1 else strand[1] if d[9] == 0 else str(d[9])]
    f += [str(int(d[10])), str(int(d[11])), "%dm" % k, "1", "read%d" % k, "sigGap", "1", "2", "3", "True", "None", "1"]
    return "\t".join(f) + "\n"


def hand_made():
    """Hostile records written by hand; each names the branch it is after."""
    r = []
    r.append((0, 1, 0, 1, 1, 1, 1, 1, 1, 1, 2, 2))                     # the pad record itself
    r.append((0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 1))                     # length 0: x_end = x_start -/+ (-1)
    r.append((5, 5, 7, 7, 0, 9, 9, 3, 3, 1, 100, 100))                 # length 0, reverse first, ratio clamps to 1
    r.append((0, 226, 0, 227, 1, 226, 0, 0, 227, 0, 227, 227))         # both diagonals, ratio exactly 1
    r.append((0, 227, 0, 228, 1, 227, 0, 0, 228, 0, 228, 228))         # ratio just above 1
    r.append((0, 500, 0, 501, 1, 500, 0, 0, 501, 0, 226, 226))         # ratio < 1 clamps; both lines clip at the border
    r.append((-300, 0, -300, 1000, 1, 900, 0, -50, 1200, 0, 800, 640))  # start points left/above the image
    r.append((100000, 0, 0, 50, 1, 0, 0, 100000, 100050, 1, 4000, 4000))  # entirely outside: nothing drawn
    r.append((0, 0, 0, 100000, 1, 99999, 0, 0, 100000, 0, 100000, 90000))  # full-length diagonals, big ratio
    r.append((10, 0, 0, 4000, 0, 20, 0, 0, 4000, 0, 4000, 4000))       # reverse segments leave through the top edge
    r.append((0, 0, 3000, 3001, 1, 10, 0, 3000, 3001, 1, 4000, 4000))  # two hits in one column from two segments -> channel 1
    r.append((0, 0, 1000, 2000, 1, 600, 0, 1500, 2500, 1, 3000, 3000))  # overlapping ref spans (tandem duplication picture)
    r.append((0, 0, 1000, 2000, 1, 1600, 0, 1200, 1700, 0, 3000, 3000))  # forward + reverse crossing
    r.append((0, 0, 0, 1, 1, 226, 0, 226, 227, 1, 227, 1))             # single pixels in opposite corners
    r.append((0, 0, 0, 2, "None", 50, 0, 50, 120, "maybe", 200, 200))  # strand neither True nor False -> None -> drawn as reverse
    r.append((1 << 30, 0, 1 << 30, (1 << 30) + 5000, 1, 0, 0, 0, 5000, 0, 1 << 30, 5000))  # huge coordinates, int(… / ratio)
    r.append((0, 0, 0, 2147483647, 1, 0, 0, 0, 227, 1, 2147483647, 2147483647))           # INT32_MAX lengths
    for k in range(12):                                                 # steep / shallow / tie slopes around the clamp
        n = 200 + 5 * k
        r.append((k, 0, 0, n, 1, n, 0, k, k + n // 2, 0, 227 + k - 6, 100 + 20 * k))
    return r


def main():
    lines = []
    seen = set()
    for name in ("collect_small", "modes", "hash_collect"):
        with open(os.path.join(HERE, name + ".expected.json")) as f:
            doc = json.load(f)

        def walk(o):
            if isinstance(o, dict):
                for k, v in o.items():
                    if k == "tsv" and isinstance(v, str):
                        for ln in v.splitlines(True):
                            key = "\t".join(ln.split("\t")[1:13])
                            if key not in seen:
                                seen.add(key)
                                lines.append(ln)
                    else:
                        walk(v)
            elif isinstance(o, list):
                for v in o:
                    walk(v)
        walk(doc)
    n_golden = len(lines)
    k = 0
    for rec in hand_made():
        lines.append(_line(rec, k))
        k += 1
    for seed in (21, 22, 23, 24):
        for rec in datagen.random_records(160, seed=seed, hostile=True):
            lines.append(_line([int(v) for v in rec], k))
            k += 1
    tmp = tempfile.mkdtemp()
    bed = os.path.join(tmp, "hostile.segments.all.bed")
    with open(bed, "w") as f:
        f.writelines(lines)
    gen = RefBatch(bed, shuffle=False, nb_classes=5, batch_size=BATCH)
    pixels, data = [], []
    lo = 0
    while lo < gen.data_size:
        images, labels = gen.next_batch(BATCH)
        assert images.shape == (BATCH, 227, 227, 3)
        for i in range(len(labels)):
            mask = images[i] + MEAN
            on = mask == 255.0
            assert np.all(on | (mask == 0.0)), "the reference produced a value outside {-mean, 255-mean}"
            pixels.append([np.flatnonzero(on[:, :, c]).tolist() for c in range(3)])
            data.append(gen.images[lo + i])
        lo += BATCH
    os.remove(bed)
    os.rmdir(tmp)
    doc = {"batch_size": BATCH, "mean": MEAN.tolist(), "n_golden": n_golden, "n_lines": len(lines), "lines": lines,
           "data": data, "pixels": pixels}
    out = os.path.join(HERE, "image_small.expected.json.gz")
    with open(out, "wb") as raw:
        with gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0, compresslevel=9) as gz:
            gz.write(json.dumps(doc, sort_keys=True, separators=(",", ":")).encode())
    per = [sum(len(c) for c in p) for p in pixels]
    print("lines", len(lines), "golden", n_golden, "images", len(pixels), "empty", sum(1 for p in per if p == 0),
          "with ch1", sum(1 for p in pixels if p[1]), "with ch2", sum(1 for p in pixels if p[2]), "bytes", os.path.getsize(out))


if __name__ == "__main__":
    main()
