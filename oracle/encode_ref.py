"""ORACLE (test infrastructure, not product): similarity-image encoding.

CPU restatement, in plain Python/NumPy, of

* ``PlotSingleImg.__init__/plot``   /root/reference/src/segmentplot/plot_segment.py:8-73
* ``Segment.__init__``              /root/reference/src/segmentplot/classes.py:42-54
* ``BatchGenerator.next_batch``     /root/reference/src/network/create_batch.py:88-155

and of the third-party arithmetic those call: OpenCV ``cv2.line`` with
thickness 1 / LINE_8 / shift 0 on a single-channel image, i.e. OpenCV's
``clipLine`` + 8-connected ``LineIterator`` (modules/imgproc/src/drawing.cpp).
OpenCV is not vendored by the reference and is not installed in this image:
the line arithmetic is restated from the published algorithm and is therefore
**parity unpinned** (no reference test or fixture pins it).  Everything above
``cv2.line`` -- segment rebuilding, ratio / truncation, argument order, channel 1's column
rule, mean subtraction, padding -- is pinned by the reference itself:
tests/golden/make_image_fixture.py runs its ``BatchGenerator.next_batch`` on 1347 TSV lines
(golden + hostile) and tests/test_image_golden.py compares this file, the C oracle and the HIP
rasteriser with the stored images bit for bit.
"""
import numpy as np

IMG = 227
MEAN = (104.0, 117.0, 124.0)          # create_batch.py:13
PAD_RECORD = (0, 1, 0, 1, 1, 1, 1, 1, 1, 1, 2, 2)   # '0_1_0_1_True_1_1_1_1_True_2_2' create_batch.py:55


def cv_clip_line(width, height, x1, y1, x2, y2):
    """OpenCV clipLine(Size, Point&, Point&): returns (visible, x1, y1, x2, y2).

    Outcode clipping, y first then x, with the intersection computed in double
    and truncated toward zero; the second endpoint's clip uses the already
    updated first endpoint (they are C++ references upstream).
    """
    right, bottom = width - 1, height - 1
    if width <= 0 or height <= 0:
        return False, x1, y1, x2, y2
    c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8
    c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(float(a - y1) * float(x2 - x1) / float(y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(float(a - y2) * float(x2 - x1) / float(y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(float(a - x1) * float(y2 - y1) / float(x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(float(a - x2) * float(y2 - y1) / float(x2 - x1))
                x2 = a
                c2 = 0
    return (c1 | c2) == 0, x1, y1, x2, y2


def cv_line_pixels(width, height, pt1, pt2):
    """Pixels (x=col, y=row) touched by cv2.line(img, pt1, pt2, color, 1) -- the
    serial LineIterator(img, pt1, pt2, 8, leftToRight=True) walk."""
    x1, y1 = int(pt1[0]), int(pt1[1])
    x2, y2 = int(pt2[0]), int(pt2[1])
    if not (0 <= x1 < width and 0 <= x2 < width and 0 <= y1 < height and 0 <= y2 < height):
        ok, x1, y1, x2, y2 = cv_clip_line(width, height, x1, y1, x2, y2)
        if not ok:
            return []
    dx, dy = x2 - x1, y2 - y1
    if dx < 0:                      # left-to-right: start from the left endpoint
        dx, dy = -dx, -dy
        x1, y1 = x2, y2
    sy = -1 if dy < 0 else 1
    dy = abs(dy)
    steep = dy > dx
    if steep:
        dx, dy = dy, dx
    err = dx - 2 * dy
    plus_delta, minus_delta = 2 * dx, -2 * dy
    x, y = x1, y1
    out = []
    for _ in range(dx + 1):
        out.append((x, y))
        minor = err < 0             # ties (err == 0) do not step
        err += minus_delta + (plus_delta if minor else 0)
        if steep:
            y += sy
            if minor:
                x += 1
        else:
            x += 1
            if minor:
                y += sy
    return out


def cv_line(img, pt1, pt2, value):
    """Stand-in for cv2.line(img2d, pt1, pt2, value, 1) used when driving the reference."""
    h, w = img.shape[:2]
    for (x, y) in cv_line_pixels(w, h, pt1, pt2):
        img[y, x] = value
    return img


def record_segments(rec):
    """12-int record -> two (xStart, xEnd, yStart, yEnd, forward) tuples, as
    BatchGenerator rebuilds them (create_batch.py:118,132: length = y_end - y_start,
    stored x_end ignored) through Segment.__init__ (classes.py:50-54)."""
    segs = []
    for k in (0, 5):
        x0, _x1, y0, y1, fwd = (int(v) for v in rec[k:k + 5])
        length = y1 - y0
        x_end = x0 + (length - 1) if fwd else x0 - (length - 1)
        y_end = y0 + (length - 1)
        segs.append((x0, x_end, y0, y_end, bool(fwd)))
    return segs


def plot_pair_mask(rec):
    """uint8 [227,227,3] 0/255 planes for one 12-int record (plot_segment.py:33-68)."""
    read_len, ref_len = int(rec[10]), int(rec[11])
    ratio = float(max(read_len, ref_len) / 227.0)          # plot_segment.py:12
    if ratio < 1:
        ratio = 1
    ch0 = np.zeros((IMG, IMG), np.uint8)
    ch2 = np.zeros((IMG, IMG), np.uint8)
    for (xs, xe, ys, ye, fwd) in record_segments(rec):
        p_start = (int(ys / ratio), int(xs / ratio))       # (col=ref, row=read)
        p_end = (int(ye / ratio), int(xe / ratio))
        if fwd:
            cv_line(ch0, p_start, p_end, 255)
        else:
            cv_line(ch0, p_end, p_start, 255)
            cv_line(ch2, p_end, p_start, 255)
    ch1 = np.zeros((IMG, IMG), np.uint8)
    for c in range(IMG):
        rows = np.nonzero(ch0[:, c])[0]
        if rows.size >= 2:
            ch1[rows, c] = 255
    return np.stack([ch0, ch1, ch2], axis=-1)


def encode_records(records, mean=MEAN):
    """[n,12] ints -> float32 [n,227,227,3] (NHWC), mean-subtracted (create_batch.py:146-150)."""
    records = np.asarray(records).reshape(-1, 12)
    out = np.empty((records.shape[0], IMG, IMG, 3), np.float32)
    m = np.asarray(mean, np.float32)
    for i, rec in enumerate(records):
        out[i] = plot_pair_mask(rec).astype(np.float32) - m
    return out


def parse_tsv_line(line):
    """One segment-TSV line -> (12-int record, label string) per create_batch.py:42-49."""
    f = line.rstrip('\n').split('\t')
    d = f[1:13]
    rec = [int(d[0]), int(d[1]), int(d[2]), int(d[3]), 1 if d[4] == 'True' else 0,
           int(d[5]), int(d[6]), int(d[7]), int(d[8]), 1 if d[9] == 'True' else 0,
           int(d[10]), int(d[11])]
    label = 'svision'.join([f[13], f[0], f[15], f[16], f[17], f[18], f[19], f[20], f[21], f[22]])
    return rec, label
