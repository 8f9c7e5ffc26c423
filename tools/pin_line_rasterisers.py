#!/opt/conda/bin/python3.9
"""A partial, OFFLINE pin of the one third-party line rule the image leg rests on.

The reference draws every segment with ``cv2.line(img, p1, p2, 255, 1)`` (/root/reference/src/segmentplot/plot_segment.py:46-52).
OpenCV is in no image this repository was built or tested in, so ``oracle.encode_ref.cv_line_pixels`` is a restatement of
OpenCV's published LineIterator ("parity unpinned").  This script compares that restatement with the two independent
line rasterisers the build container does hold -- ``skimage.draw.line`` (scikit-image 0.18.3) and Pillow's ``ImageDraw.line``
(8.4.0), both under /opt/conda/lib/python3.9 -- on

  * every in-bounds end-point pair the image fixture's 1,347 TSV lines produce (tests/golden/image_small.expected.json.gz), and
  * 100,000 random in-bounds end-point pairs (seed 5),

and stores the outcome in tests/golden/line_pin.json.  Lines on which Bresenham's error term hits zero ("ties": the three
implementations are free to round either way, and do) are counted separately.  What it shows: on every line WITHOUT a tie the
three agree pixel for pixel; on lines with ties the oracle keeps OpenCV's documented rule (a tie does not step).  What it does not
show: OpenCV itself, its clipLine for end points outside the image, or its tie rule -- tools/pin_thirdparty.py does, on a
machine that has cv2.

Run:  /opt/conda/bin/python3.9 tools/pin_line_rasterisers.py     (numpy, scikit-image, Pillow; no GPU, no libsvx.so)
"""
import gzip
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
IMG = 227


def fixture_pairs():
    from oracle import encode_ref
    with gzip.open(os.path.join(ROOT, "tests", "golden", "image_small.expected.json.gz"), "rt") as f:
        fx = json.load(f)
    lines = []

    def walk(o):
        if isinstance(o, str):
            if o.count("\t") >= 22:
                lines.append(o)
        elif isinstance(o, dict):
            for v in o.values():
                walk(v)
        elif isinstance(o, list):
            for v in o:
                walk(v)
    walk(fx)
    pairs = set()
    for line in lines:
        try:
            rec, _label = encode_ref.parse_tsv_line(line)
        except (ValueError, IndexError):
            continue
        ratio = float(max(rec[10], rec[11]) / 227.0)
        if ratio < 1:
            ratio = 1
        try:
            segs = encode_ref.record_segments(rec)
        except Exception:                                      # noqa: BLE001
            continue
        for xs, xe, ys, ye, fwd in segs:
            p1, p2 = (int(ys / ratio), int(xs / ratio)), (int(ye / ratio), int(xe / ratio))
            if not fwd:
                p1, p2 = p2, p1
            pairs.add((p1, p2))
    return len(lines), sorted(pairs)


def has_tie(p1, p2):
    """Does OpenCV's error term (err = dx - 2 dy, then += -2 dy (+ 2 dx on a minor step)) ever test exactly 0?"""
    dx, dy = abs(p2[0] - p1[0]), abs(p2[1] - p1[1])
    if dy > dx:
        dx, dy = dy, dx
    err = dx - 2 * dy
    for _ in range(dx + 1):
        if err == 0:
            return True
        err += -2 * dy + (2 * dx if err < 0 else 0)
    return False


def main():
    from PIL import Image, ImageDraw
    import PIL
    import skimage
    from skimage.draw import line as sk_line
    from oracle import encode_ref

    def oracle_px(p1, p2):
        return set(encode_ref.cv_line_pixels(IMG, IMG, p1, p2))

    def skimage_px(p1, p2):
        rr, cc = sk_line(p1[1], p1[0], p2[1], p2[0])
        return set(zip(cc.tolist(), rr.tolist()))

    def pillow_px(p1, p2):
        im = Image.new("L", (IMG, IMG), 0)
        ImageDraw.Draw(im).line([p1, p2], fill=255)
        ys, xs = np.nonzero(np.asarray(im))
        return set(zip(xs.tolist(), ys.tolist()))

    n_lines, fixture = fixture_pairs()
    inb = [(a, b) for a, b in fixture if all(0 <= v < IMG for v in a + b)]
    rng = np.random.default_rng(5)
    rnd = [((int(a), int(b)), (int(c), int(d))) for a, b, c, d in rng.integers(0, IMG, size=(100_000, 4))]
    report = {"oracle": "oracle/encode_ref.py cv_line_pixels (OpenCV LineIterator restated, 8-connected, thickness 1)",
              "against": {"skimage.draw.line": skimage.__version__, "PIL.ImageDraw.line": PIL.__version__},
              "python": sys.version.split()[0], "fixture_tsv_lines": n_lines, "fixture_endpoint_pairs": len(fixture),
              "fixture_pairs_out_of_bounds_not_compared": len(fixture) - len(inb), "sets": {}}
    ok = True
    for name, pairs in (("fixture_in_bounds", inb), ("random_100000_seed5", rnd)):
        st = {"pairs": len(pairs), "no_tie": {"n": 0, "skimage_equal": 0, "pillow_equal": 0},
              "tie": {"n": 0, "skimage_equal": 0, "pillow_equal": 0, "skimage_max_pixels_different": 0, "pillow_max_pixels_different": 0},
              "same_pixel_count_always": True, "endpoints_always_drawn": True}
        for p1, p2 in pairs:
            o, s, p = oracle_px(p1, p2), skimage_px(p1, p2), pillow_px(p1, p2)
            k = "tie" if has_tie(p1, p2) else "no_tie"
            st[k]["n"] += 1
            st[k]["skimage_equal"] += o == s
            st[k]["pillow_equal"] += o == p
            if k == "tie":
                st[k]["skimage_max_pixels_different"] = max(st[k]["skimage_max_pixels_different"], len(o ^ s) // 2)
                st[k]["pillow_max_pixels_different"] = max(st[k]["pillow_max_pixels_different"], len(o ^ p) // 2)
            st["same_pixel_count_always"] &= len(o) == len(s) == len(p) == max(abs(p2[0] - p1[0]), abs(p2[1] - p1[1])) + 1
            st["endpoints_always_drawn"] &= p1 in o and p2 in o
        report["sets"][name] = st
        ok &= st["no_tie"]["skimage_equal"] == st["no_tie"]["n"] and st["same_pixel_count_always"] and st["endpoints_always_drawn"]
    report["verdict"] = ("on every line without a Bresenham tie the oracle's walk equals skimage.draw.line pixel for pixel; "
                         "ties are where implementations differ by design (OpenCV: a tie does not step) and remain pinned to OpenCV's "
                         "published rule only") if ok else "MISMATCH on a line without a tie"
    out = os.path.join(ROOT, "tests", "golden", "line_pin.json")
    with open(out, "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(report, indent=1, sort_keys=True))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
