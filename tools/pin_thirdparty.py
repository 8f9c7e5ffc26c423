#!/usr/bin/env python3
"""Closes the three INHERENT parity pins of this repository on a machine that has the real libraries.

The build image has no OpenCV, no TensorFlow and no pysam/htslib, so three legs of the oracle are restatements of
published third-party behaviour (SURVEY 8(a')):

  cv2     cv2.line (thickness 1, LINE_8) == oracle.encode_ref.cv_line              /root/reference/src/segmentplot/plot_segment.py:46-52
  tf      TF1 graph ops of the reference's AlexNet == oracle.alexnet_ref            /root/reference/src/network/alexnet.py:109-166
          tf.compat.v1.train.Saver checkpoint <-> svision_amd.network.tf_checkpoint  /root/reference/src/network/predict.py:181-184
  pysam   AlignedSegment / AlignmentFile fields == svision_amd.io.bam               /root/reference/src/collection/collect_signatures.py:131-155

Run it from the repository root wherever any of those libraries is installed:

    python tools/pin_thirdparty.py            # every leg whose library imports; the others are reported as skipped
    python tools/pin_thirdparty.py cv2 tf     # selected legs

It needs NumPy and this repository only (oracle/, svision_amd/io, svision_amd/network/tf_checkpoint.py, tests/golden);
no GPU, no libsvx.so.  Exit status 0 = every leg that ran agrees; 1 = a mismatch (printed with the offending input).
It never runs in this repository's own test suite: neither the build container nor the GPU box has the libraries.
"""
import gzip
import json
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


class Mismatch(Exception):
    pass


def _hostile_records():
    """The 12-int records of the image fixture (678 golden + 669 hostile TSV lines, tests/golden/make_image_fixture.py)."""
    from oracle import encode_ref
    with gzip.open(os.path.join(GOLDEN, "image_small.expected.json.gz"), "rt") as f:
        fx = json.load(f)
    lines = fx["lines"] if "lines" in fx else [l for k in fx for l in (fx[k] if isinstance(fx[k], list) else []) if isinstance(l, str) and "\t" in l]
    recs = []
    for line in lines:
        try:
            recs.append(encode_ref.parse_tsv_line(line)[0])
        except (ValueError, IndexError):
            continue
    return recs


# ---------------------------------------------------------------------------------------------------------------- cv2
def pin_cv2():
    import cv2
    from oracle import encode_ref
    n = 0

    def same(pt1, pt2):
        a = np.zeros((227, 227), np.uint8)
        b = np.zeros((227, 227), np.uint8)
        cv2.line(a, pt1, pt2, 255, 1)
        encode_ref.cv_line(b, pt1, pt2, 255)
        if not np.array_equal(a, b):
            raise Mismatch("cv2.line %s -> %s: %d pixels differ (cv2 %d set, restatement %d set)"
                           % (pt1, pt2, int((a != b).sum()), int((a > 0).sum()), int((b > 0).sum())))

    # 1. every line the reference would draw for the fixture's records (plot_segment.py:33-52), both argument orders
    for rec in _hostile_records():
        read_len, ref_len = int(rec[10]), int(rec[11])
        ratio = max(float(max(read_len, ref_len) / 227.0), 1)
        for (xs, xe, ys, ye, fwd) in encode_ref.record_segments(rec):
            p0, p1 = (int(ys / ratio), int(xs / ratio)), (int(ye / ratio), int(xe / ratio))
            if max(abs(v) for v in p0 + p1) >= 2 ** 31:       # cv2 takes C ints
                continue
            same(p0, p1) if fwd else same(p1, p0)
            n += 1
    # 2. random lines: inside, crossing every edge and corner, far outside (clipLine's double arithmetic), ties, degenerate
    rng = np.random.default_rng(20260928)
    for scale in (226, 400, 5000, 10 ** 6, 10 ** 8):
        for _ in range(4000):
            x1, y1, x2, y2 = (int(v) for v in rng.integers(-scale, scale + 227, 4))
            same((x1, y1), (x2, y2))
            n += 1
    for k in range(-3, 231):                                   # horizontals, verticals, exact diagonals through the borders
        for pts in (((k, -5), (k, 240)), ((-5, k), (240, k)), ((k, k), (k + 300, k + 300)), ((k, 226 - k), (k + 7, 226 - k - 14)), ((k, k), (k, k))):
            same(*pts)
            n += 1
    # 3. the whole image of a record: reference procedure with the real cv2 vs oracle.encode_ref.plot_pair_mask
    for rec in _hostile_records()[::5]:
        read_len, ref_len = int(rec[10]), int(rec[11])
        ratio = max(float(max(read_len, ref_len) / 227.0), 1)
        ch0, ch2 = np.zeros((227, 227), np.uint8), np.zeros((227, 227), np.uint8)
        ok = True
        for (xs, xe, ys, ye, fwd) in encode_ref.record_segments(rec):
            p0, p1 = (int(ys / ratio), int(xs / ratio)), (int(ye / ratio), int(xe / ratio))
            if max(abs(v) for v in p0 + p1) >= 2 ** 31:
                ok = False
                break
            if fwd:
                cv2.line(ch0, p0, p1, 255, 1)
            else:
                cv2.line(ch0, p1, p0, 255, 1)
                cv2.line(ch2, p1, p0, 255, 1)
        if ok:
            want = encode_ref.plot_pair_mask(rec)
            if not (np.array_equal(want[..., 0], ch0) and np.array_equal(want[..., 2], ch2)):
                raise Mismatch("image of record %s differs" % (list(rec),))
    return "cv2 %s: %d lines identical to oracle.encode_ref.cv_line" % (cv2.__version__, n)


# ----------------------------------------------------------------------------------------------------------------- tf
def pin_tf():
    import tensorflow as tf
    from oracle import alexnet_ref, encode_ref
    from svision_amd.network import tf_checkpoint as ck
    tf1 = tf.compat.v1
    tf1.disable_eager_execution()
    params = alexnet_ref.random_params(seed=3)
    rec = np.asarray(_hostile_records()[:6], np.int64)
    images = encode_ref.encode_records(rec)
    report = []

    # 1. op by op on the layer shapes of alexnet.py:26-58
    rng = np.random.default_rng(5)
    with tf1.Session(graph=tf.Graph()) as sess:
        x = rng.standard_normal((2, 27, 27, 96)).astype(np.float32)
        got = sess.run(tf.nn.local_response_normalization(tf.constant(x), depth_radius=2, alpha=2e-05, beta=0.75, bias=1.0))
        want = alexnet_ref._lrn(x)
        if np.abs(got - want).max() > 1e-5:
            raise Mismatch("tf.nn.local_response_normalization differs from alexnet_ref._lrn by %g" % np.abs(got - want).max())
        got = sess.run(tf.nn.max_pool2d(tf.constant(x), ksize=[1, 3, 3, 1], strides=[1, 2, 2, 1], padding="VALID"))
        if not np.array_equal(got, alexnet_ref._max_pool_3x3s2_valid(x)):
            raise Mismatch("max_pool 3x3/2 VALID differs")
        for name, kh, kw, cin, cout, stride, padding, groups in alexnet_ref.LAYERS:
            hw = 227 if name == "conv1" else 27 if name == "conv2" else 13
            xin = rng.standard_normal((2, hw, hw, cin * groups)).astype(np.float32)
            w, b = params[name + "/weights"], params[name + "/biases"]
            if groups == 1:
                y = tf.nn.conv2d(tf.constant(xin), tf.constant(w), strides=[1, stride, stride, 1], padding=padding)
            else:                                              # alexnet.py:118-127: split, convolve, concat
                xs = tf.split(tf.constant(xin), groups, axis=3)
                ws = tf.split(tf.constant(w), groups, axis=3)
                y = tf.concat([tf.nn.conv2d(a, k, strides=[1, stride, stride, 1], padding=padding) for a, k in zip(xs, ws)], axis=3)
            got = sess.run(tf.nn.relu(tf.nn.bias_add(y, tf.constant(b))))
            want = alexnet_ref._conv_layer(xin, w, b, stride, padding, groups)
            err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-6)
            if got.shape != want.shape or err > 1e-4:
                raise Mismatch("%s: shape %s vs %s, relative error %g" % (name, got.shape, want.shape, err))
            report.append("%s %.1e" % (name, err))

    # 2. the whole graph, built the way alexnet.py builds it, on real similarity images
    with tf1.Session(graph=tf.Graph()) as sess:
        x = tf1.placeholder(tf.float32, [None, 227, 227, 3])
        var = {}
        for key, value in params.items():
            scope, leaf = key.split("/")
            with tf1.variable_scope(scope, reuse=tf1.AUTO_REUSE):
                var[key] = tf1.get_variable(leaf, initializer=tf.constant(value))
        h = x
        for name, _kh, _kw, _cin, _cout, stride, padding, groups in alexnet_ref.LAYERS:
            w, b = var[name + "/weights"], var[name + "/biases"]
            if groups == 1:
                h = tf.nn.conv2d(h, w, strides=[1, stride, stride, 1], padding=padding)
            else:
                h = tf.concat([tf.nn.conv2d(a, k, strides=[1, stride, stride, 1], padding=padding)
                               for a, k in zip(tf.split(h, groups, axis=3), tf.split(w, groups, axis=3))], axis=3)
            h = tf.nn.relu(tf.nn.bias_add(h, b))
            if name in ("conv1", "conv2"):
                h = tf.nn.max_pool2d(h, ksize=[1, 3, 3, 1], strides=[1, 2, 2, 1], padding="VALID")
                h = tf.nn.local_response_normalization(h, depth_radius=2, alpha=2e-05, beta=0.75, bias=1.0)
            elif name == "conv5":
                h = tf.nn.max_pool2d(h, ksize=[1, 3, 3, 1], strides=[1, 2, 2, 1], padding="VALID")
        h = tf.reshape(h, [-1, 6 * 6 * 256])
        h = tf.nn.relu(tf1.nn.xw_plus_b(h, var["fc6/weights"], var["fc6/biases"]))
        h = tf.nn.relu(tf1.nn.xw_plus_b(h, var["fc7/weights"], var["fc7/biases"]))
        logits = tf1.nn.xw_plus_b(h, var["fc8/weights"], var["fc8/biases"])
        sess.run(tf1.global_variables_initializer())
        got_logits, got_prob = sess.run([logits, tf.nn.softmax(logits)], feed_dict={x: images})
        _wl, _wc, want_prob = alexnet_ref.predict(params, images)
        err = float(np.abs(got_prob - want_prob).max())
        if err > 1e-3:
            raise Mismatch("softmax of the whole graph differs from alexnet_ref by %g (north_star tolerance 1e-3)" % err)
        report.append("softmax %.1e" % err)

        # 3. checkpoint bytes: TF's Saver -> own reader, own writer -> TF's Saver
        d = tempfile.mkdtemp(prefix="svx_pin_ckpt_")
        try:
            prefix = os.path.join(d, "tf-written.ckpt")
            tf1.train.Saver().save(sess, prefix, write_meta_graph=False)
            back = ck.read_checkpoint(prefix)
            for key, value in params.items():
                if key not in back or not np.array_equal(np.asarray(back[key]), value):
                    raise Mismatch("own reader returns a different %s from the TF-written checkpoint" % key)
            own = os.path.join(d, "own-written.ckpt")
            ck.write_checkpoint(own, {k: v + np.float32(1) for k, v in params.items()})
            tf1.train.Saver().restore(sess, own)
            for key, value in params.items():
                if not np.array_equal(sess.run(var[key]), value + np.float32(1)):
                    raise Mismatch("TF restores a different %s from the own-written checkpoint" % key)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return "tensorflow %s: %s; Saver round trips both ways" % (tf.__version__, ", ".join(report))


# -------------------------------------------------------------------------------------------------------------- pysam
def pin_pysam():
    import pysam
    from oracle import cigar_ref
    from svision_amd.io import bam
    n_rec = n_fetch = 0
    d = tempfile.mkdtemp(prefix="svx_pin_bam_")
    try:
        for name in ("collect_small.bam", "ont_small.bam", "dup_small.bam"):
            path = os.path.join(d, name)
            shutil.copy(os.path.join(GOLDEN, name), path)
            pysam.index(path)
            table = bam.read_bam_python(path)
            # the derived fields svx_cigar_scan computes on the device, restated on the host by the oracle
            stats = np.asarray([cigar_ref.alignment_stats([(int(w) & 15, int(w) >> 4) for w in table.cigar[table.cig_off[i]:table.cig_off[i + 1]]])
                                for i in range(len(table))], np.int64).reshape(-1, 4)
            table.attach_scan(stats.astype(np.int32))
            with pysam.AlignmentFile(path) as f:
                if list(f.references) != table.references or list(f.lengths) != table.lengths:
                    raise Mismatch("%s: reference dictionary differs" % name)
                for i, a in enumerate(f.fetch(until_eof=True)):
                    words = table.cigar[table.cig_off[i]:table.cig_off[i + 1]]
                    cigar = "".join("%d%s" % (int(w) >> 4, cigar_ref.OPS[int(w) & 15]) for w in words) or None
                    mine = (int(table.tid[i]), int(table.pos[i]), int(table.flag[i]), int(table.mapq[i]), table.names[int(table.name_id[i])], cigar)
                    theirs = (a.reference_id, a.reference_start, a.flag, a.mapping_quality, a.query_name, a.cigarstring)
                    if mine != theirs:
                        raise Mismatch("%s record %d: %s vs pysam %s" % (name, i, mine, theirs))
                    if cigar is not None and not a.is_unmapped:
                        span, lead, trail, qlen = (int(v) for v in stats[i])
                        hard_lead = int(words[0] >> 4) if len(words) and int(words[0]) & 15 == 5 else 0
                        hard_trail = int(words[-1] >> 4) if len(words) > 1 and int(words[-1]) & 15 == 5 else 0
                        if a.reference_end != a.reference_start + span:
                            raise Mismatch("%s record %d: reference_end %s vs %s" % (name, i, a.reference_end, a.reference_start + span))
                        # the reference rewrites H to S before reading these (collect_signatures.py:91): clips count either way here
                        if a.query_alignment_start != lead - hard_lead or a.query_alignment_end != qlen - hard_lead - trail:
                            raise Mismatch("%s record %d: query_alignment_start/end %s/%s vs %s/%s" % (
                                name, i, a.query_alignment_start, a.query_alignment_end, lead - hard_lead, qlen - hard_lead - trail))
                        if a.infer_read_length() != qlen:
                            raise Mismatch("%s record %d: read length %s vs %s" % (name, i, a.infer_read_length(), qlen))
                    n_rec += 1
                rng = np.random.default_rng(7)
                for _ in range(300):                           # fetch(contig, start, end): the same records in the same order; coverage
                    t = int(rng.integers(0, len(table.references)))
                    s = int(rng.integers(0, max(1, table.lengths[t] - 1)))
                    e = int(min(table.lengths[t], s + rng.integers(1, 60_000)))
                    want = [(a.reference_start, a.query_name, a.flag) for a in f.fetch(table.references[t], s, e)]
                    rows = table.fetch(t, s, e)
                    got = [(int(table.pos[r]), table.names[int(table.name_id[r])], int(table.flag[r])) for r in rows]
                    if got != want:
                        raise Mismatch("%s fetch(%s, %d, %d): %d rows vs pysam %d" % (name, table.references[t], s, e, len(got), len(want)))
                    if int(table.count_overlaps(t, [s], [e])[0]) != f.count(table.references[t], s, e, read_callback="nofilter"):
                        raise Mismatch("%s count(%s, %d, %d) differs" % (name, table.references[t], s, e))
                    n_fetch += 1
            if os.path.exists(path + ".bai"):                  # an htslib-written index through the own .bai reader
                spans = bam.read_bai(path + ".bai")
                for t, span in enumerate(spans):
                    part = bam.read_bam(path, tids=[t]) if span is not None and os.path.exists(os.path.join(ROOT, "svision_amd", "libsvx.so")) else None
                    if part is not None and len(part) != int((table.tid == t).sum()):
                        raise Mismatch("%s: shard of reference %d through the htslib index has %d records, expected %d" % (name, t, len(part), int((table.tid == t).sum())))
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return "pysam %s: %d records, %d fetch / count windows identical to svision_amd.io.bam" % (pysam.__version__, n_rec, n_fetch)


LEGS = {"cv2": pin_cv2, "tf": pin_tf, "pysam": pin_pysam}


def main(argv):
    wanted = argv or list(LEGS)
    bad = 0
    for leg in wanted:
        if leg not in LEGS:
            print("unknown leg %r (choose from %s)" % (leg, ", ".join(LEGS)))
            return 2
        try:
            print("[ok]      " + LEGS[leg]())
        except ImportError as exc:
            print("[skipped] %s: %s" % (leg, exc))
        except Mismatch as exc:
            bad += 1
            print("[DIFFERS] %s: %s" % (leg, exc))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
