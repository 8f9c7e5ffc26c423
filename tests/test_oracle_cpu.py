"""CPU tests: the oracle's own consistency (Python restatement == C restatement)."""
import numpy as np
import pytest

from oracle import cigar_ref, encode_ref
from tests import datagen


def test_closed_form_matches_line_iterator():
    # the parallel closed form used by the HIP kernel == the serial iterator, all small lines
    for x1 in range(0, 12, 3):
        for y1 in range(0, 12, 5):
            for x2 in range(0, 14):
                for y2 in range(0, 14):
                    px = encode_ref.cv_line_pixels(16, 16, (x1, y1), (x2, y2))
                    dx, dy = x2 - x1, y2 - y1
                    sx0, sy0 = x1, y1
                    if dx < 0:
                        dx, dy, sx0, sy0 = -dx, -dy, x2, y2
                    sy = -1 if dy < 0 else 1
                    dy = abs(dy)
                    steep = dy > dx
                    if steep:
                        dx, dy = dy, dx
                    got = []
                    for k in range(dx + 1):
                        m = (2 * dy * k + dx - 1) // (2 * dx) if dx else 0
                        got.append((sx0 + m, sy0 + sy * k) if steep else (sx0 + k, sy0 + sy * m))
                    assert got == px


def test_python_and_c_raster_agree(oracle_lib):
    from oracle import cbind
    rec = datagen.random_records(300, seed=11)
    a = encode_ref.encode_records(rec)
    b = cbind.rasterize(rec, "NHWC")
    assert a.dtype == b.dtype == np.float32
    assert np.array_equal(a, b)
    c = cbind.rasterize(rec, "NCHW")
    assert np.array_equal(c, np.transpose(b, (0, 3, 1, 2)))


def test_pad_record_image():
    img = encode_ref.plot_pair_mask(encode_ref.PAD_RECORD)
    # both pad segments are single points at (row 0, col 0) and (row 1, col 1): no column has two hits
    assert img[0, 0, 0] == 255 and img[1, 1, 0] == 255 and img[..., 0].sum() == 2 * 255
    assert img[..., 1].sum() == 0 and img[..., 2].sum() == 0


def test_python_and_c_cigar_agree(oracle_lib):
    from oracle import cbind
    cigar, off, ref_start = datagen.random_cigars(500, seed=3, mean_ops=150, long_gap_rate=0.02)
    gaps, gap_off, stats = cbind.cigar_scan(cigar, off, ref_start, 50)
    for a in range(len(ref_start)):
        ops = [(int(w) & 15, int(w) >> 4) for w in cigar[int(off[a]):int(off[a + 1])]]
        want = cigar_ref.scan_long_gaps(ops, int(ref_start[a]), 50)
        got = gaps[int(gap_off[a]):int(gap_off[a + 1])]
        assert [(int(g["op"]), int(g["kind"]), int(g["read_pos"]), int(g["ref_pos"]), int(g["len"])) for g in got] == want
        assert tuple(int(v) for v in stats[a]) == cigar_ref.alignment_stats(ops)


def test_cigar_text_roundtrip():
    ops = cigar_ref.parse_cigar("100S2000M300I1500M200D1500M50S")
    assert cigar_ref.inside_align(ops, 100, 10000, 15200, 50) == [
        [100, 2100, 10000, 11999], [2401, 3901, 12000, 13500], [3901, 5401, 13700, 15200]]
    assert cigar_ref.inside_align(cigar_ref.parse_cigar("5000M"), 0, 0, 5000, 50) is None


def test_line_walk_was_compared_with_two_independent_rasterisers():
    """tools/pin_line_rasterisers.py (run under /opt/conda python3.9 in the build container; its result is committed): on every
    line without a Bresenham tie -- fixture end points and 100,000 random ones -- the oracle's OpenCV LineIterator restatement
    equals skimage.draw.line and Pillow's ImageDraw.line pixel for pixel.  (OpenCV's tie rule itself stays pinned to its
    published text only: DESIGN.md section 3.)"""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "line_pin.json")) as f:
        pin = json.load(f)
    for name, st in pin["sets"].items():
        assert st["no_tie"]["n"] > 500 and st["no_tie"]["skimage_equal"] == st["no_tie"]["pillow_equal"] == st["no_tie"]["n"], name
        assert st["same_pixel_count_always"] and st["endpoints_always_drawn"], name
        assert st["tie"]["n"] > 0
    assert pin["sets"]["random_100000_seed5"]["pairs"] == 100000
