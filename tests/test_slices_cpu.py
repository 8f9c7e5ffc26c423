"""Slices of a chromosome (svision_amd/ingest_gpu.py: Unit, plan_units; ingest.ChromosomeFeed._slice_complete) without a GPU.

The device engine hands a chromosome over in slices of whole collection windows: the records between the linear-index entries
of `first window's start - margin` and `last window's end + margin`.  Here the same file ranges are decoded by the HOST
decoder (svx_bam_open_range takes virtual offsets), scanned by the oracle, and every window is collected, voted and stitched
on its slice: the results must be those of the whole chromosome, byte for byte -- the reference fetches window by window from
the whole file (run_collection.py:23-26), so a slice must never change what a window sees."""
import io
import os
import zlib

import numpy as np
import pytest

from svision_amd import _lib, synth
from svision_amd.collection.output_clusters import collect_pair_lines
from svision_amd.collection.run_collection import detect_window
from svision_amd.ingest import ChromosomeFeed
from svision_amd.ingest_gpu import DeviceDecoder, Unit, voff_at
from svision_amd.io import bam
from svision_amd.pipeline import WindowResult, _vote, stitch_windows
from svision_amd.sample import Sample
from tests import helpers

LENGTH, WINDOW = 1_200_000, 100_000


@pytest.fixture(scope="module")
def sliced(tmp_path_factory):
    d = tmp_path_factory.mktemp("slices")
    cfg = synth.SimConfig(contigs=[("chrA", LENGTH), ("chrB", 300_000)], coverage=14, read_len_mean=8000, read_len_sd=1500, err_rate=0.004,
                          sv_spacing=9_000, sv_min_gap=6_000, sv_max=3000, inline_max=1200, seed=77)
    table, genome, _svs = synth.simulate(cfg)
    path = str(d / "s.bam")
    bam.write_bam(path, table, level=1, index=True)
    head = bam.read_bam_header(path)
    dec = DeviceDecoder(path, path + ".bai", head.references, head.lengths, head.header_text, "cpu")
    return path, table, bam.Fasta(sequences=genome), dec


def _windows(length):
    return [(a, min(length, a + WINDOW)) for a in range(0, length, WINDOW)]


def _range_table(dec, unit):
    lib = _lib.load()
    h = lib.svx_bam_open_range(dec.path.encode(), 2, 0, unit.vlo, unit.vhi)
    assert h, lib.svx_bam_error().decode()
    return bam._table_from_handle(lib, h, False)


def test_plan_cuts_whole_windows_in_file_order(sliced):
    _path, _table, _fasta, dec = sliced
    units = dec.plan_units([0, 1], lambda t: _windows(dec.lengths[t]), margin=64 << 10, slice_bytes=1, min_span_margins=0)
    a = [u for u in units if u.tid == 0]
    assert [(u.lo, u.hi) for u in a] == _windows(LENGTH)                        # slice_bytes below one window: a slice per window
    assert a[0].left_edge is None and a[0].vlo == dec.spans[0][0] and a[-1].to_end and a[-1].vhi == dec.spans[0][1]
    assert all(u.vlo <= v.vlo and u.vhi <= v.vhi and u.vlo < u.vhi for u, v in zip(a, a[1:]))
    assert all(v.vlo < u.vhi for u, v in zip(a, a[1:]))                         # neighbours overlap by their margins
    assert [u.tid for u in units] == [0] * len(a) + [1] * (len(units) - len(a))
    # larger slices: runs of windows; every window in exactly one slice
    big = dec.plan_units([0], lambda t: _windows(LENGTH), margin=64 << 10, slice_bytes=((dec.spans[0][1] >> 16) - (dec.spans[0][0] >> 16)) // 3, min_span_margins=0)
    assert 3 <= len(big) < 12 and big[0].lo == 0 and big[-1].hi == LENGTH
    assert all(u.hi == v.lo for u, v in zip(big, big[1:]))
    # no windows known / one window: the whole chromosome
    whole = dec.plan_units([0, 1], None, margin=0)
    assert [(u.tid, u.lo, u.hi, u.vlo, u.vhi) for u in whole] == [(t, 0, dec.lengths[t], dec.spans[t][0], dec.spans[t][1]) for t in (0, 1)]
    one = dec.plan_units([1], lambda t: [(0, dec.lengths[1])], margin=64 << 10)
    assert len(one) == 1 and one[0].left_edge is None and one[0].to_end
    # long reads: behind the first slice a slice spans at least 13 margins, so that the margins stay a small part of what is read
    wide = dec.plan_units([0], lambda t: _windows(LENGTH), margin=32 << 10, slice_bytes=1)
    assert (wide[0].lo, wide[0].hi) == (0, WINDOW) and all(u.hi - u.lo >= 13 * (32 << 10) or u.last for u in wide[1:]) and len(wide) == 4
    assert [w for u in wide for w in u.windows] == _windows(LENGTH)
    # resumed behind a rejected slice
    rest = dec.plan_units([0, 1], lambda t: _windows(dec.lengths[t]), margin=128 << 10, slice_bytes=1, resume=(0, 500_000), min_span_margins=0)
    assert rest[0].tid == 0 and rest[0].lo == 500_000 and rest[0].left_edge == (500_000 - (128 << 10)) >> 14 << 14
    assert {u.tid for u in rest} == {0, 1}


def test_linear_index_lookup(sliced):
    _path, table, _fasta, dec = sliced
    span = dec.spans[0]
    assert voff_at(span, -5) == span[0] and voff_at(span, 0) == span[0] and voff_at(span, 10 * LENGTH) == span[1]
    # the entry of a bin = the first record overlapping it: everything in front of it ends at or before the bin's start
    rows = np.flatnonzero(table.tid == 0)
    scan = helpers.oracle_scan(table, 50)
    ref_end = table.pos[rows].astype(np.int64) + np.maximum(scan[2][rows, 0], 1)
    for coord in (16384 * 7, 300_000, 777_777):
        u = Unit(0, 0, LENGTH, voff_at(span, coord), span[1])
        t = _range_table(dec, u)
        n_front = rows.size - len(t)
        assert n_front > 0 and (ref_end[:n_front] <= coord >> 14 << 14).all()
        assert np.array_equal(t.pos, table.pos[rows][n_front:])


def test_guessed_reach_covers_the_reads(sliced):
    _path, table, _fasta, dec = sliced
    scan = helpers.oracle_scan(table, 50)
    smp = Sample.with_scan(table, None, 50, scan)
    assert dec.estimate_reach([0, 1]) >= smp.reach() and dec.estimate_reach([0, 1]) % 16384 == 0


def _votes(sample_of, opts, chrom, windows):
    results, texts = [], []
    for part, (start, end) in enumerate(windows):
        smp = sample_of(start)
        _s, clusters = detect_window(opts, smp, chrom, start, end, part)
        lines = collect_pair_lines(clusters, opts)
        texts.append("".join(ln.text() for ln in lines))
        h = np.array([zlib.crc32(ln.text().encode()) for ln in lines], np.int64)
        cls = h % 5
        prob = np.full((len(lines), 5), 0.05, np.float32)
        prob[np.arange(len(lines)), cls] = (0.5 + (h % 50) / 100.0).astype(np.float32)
        res = WindowResult()
        res.chrom, res.start, res.end = chrom, start, end
        res.vcf, res.scores, res.n_sites, res.head, res.tail = _vote(smp, opts, chrom, lines, cls, prob, start, end)
        results.append(res)
    return results, texts


@pytest.mark.parametrize("slice_windows", [1, 3])
def test_windows_on_their_slices_equal_windows_on_the_whole_chromosome(sliced, slice_windows):
    _path, table, fasta, dec = sliced
    opts = helpers.default_options(min_support=3, batch_size=64, window_size=WINDOW, qname=True)
    windows = _windows(LENGTH)
    whole = Sample.with_scan(table, fasta, 50, helpers.oracle_scan(table, 50))
    want_results, want_tsv = _votes(lambda _s: whole, opts, "chrA", windows)
    want = stitch_windows(want_results, opts, whole)["chrA"]
    assert want[0].count("\n") > 40 and sum(bool(r.head) + bool(r.tail) for r in want_results) > 4

    per_window = ((dec.spans[0][1] >> 16) - (dec.spans[0][0] >> 16)) // len(windows)
    units = dec.plan_units([0], lambda t: windows, margin=dec.estimate_reach([0]), slice_bytes=per_window * slice_windows + per_window // 2, min_span_margins=0)
    assert len(units) >= len(windows) // slice_windows - 1 and len(units) > 2
    samples = []
    for u in units:
        t = _range_table(dec, u)
        assert len(t) < 0.6 * int((table.tid == 0).sum())                      # a slice, not the chromosome
        smp = Sample.with_scan(t, fasta, 50, helpers.oracle_scan(t, 50))
        assert ChromosomeFeed._slice_complete(u, smp)
        samples.append((u, smp))

    def sample_of(start):
        return next(s for u, s in samples if u.lo <= start < u.hi)
    got_results, got_tsv = _votes(sample_of, opts, "chrA", windows)
    assert got_tsv == want_tsv                                                  # region strings carry the coverage counts
    got = stitch_windows(got_results, opts, lambda _c, start: sample_of(start))["chrA"]
    assert got == want                                                          # GT:DR:DV from the genotyper's +-1000 bp, edge sites once


def test_a_margin_that_is_too_small_is_noticed(sliced):
    _path, table, fasta, dec = sliced
    windows = _windows(LENGTH)
    units = dec.plan_units([0], lambda t: windows, margin=1, slice_bytes=1, min_span_margins=0)
    bad = 0
    for u in units[1:-1]:
        t = _range_table(dec, u)
        smp = Sample.with_scan(t, fasta, 50, helpers.oracle_scan(t, 50))
        bad += not ChromosomeFeed._slice_complete(u, smp)
    assert bad == len(units) - 2


def test_a_record_spanning_a_whole_slice_is_not_an_empty_slice(tmp_path):
    """with a small window under long records both linear-index look-ups of a slice (start - margin, end + 2 x margin)
    can return the SAME record -- the one that spans all of it.  Such a slice used to come out as the empty file range and was
    handed over as "no records": the spanning record and everything starting inside the window behind it were dropped.  Now the
    range reaches to the next index entry, so the slice holds that record and the completeness check judges it like any other."""
    cfg = synth.SimConfig(contigs=[("ctg", 1_500_000)], coverage=2, read_len_mean=500_000, read_len_sd=100_000, err_rate=0.001, seed=5)
    table, genome, _svs = synth.simulate(cfg)
    path = str(tmp_path / "long.bam")
    bam.write_bam(path, table, level=1, index=True)
    head = bam.read_bam_header(path)
    dec = DeviceDecoder(path, path + ".bai", head.references, head.lengths, head.header_text, "cpu")
    span = dec.spans[0]
    windows = [(a, a + 50_000) for a in range(0, 1_500_000, 50_000)]
    table.attach_scan(helpers.oracle_scan(table, 50)[2])
    ref_end = table.ref_end()
    fasta = bam.Fasta(sequences=genome)
    margin = 16 << 10
    units = dec.plan_units([0], lambda t: windows, margin=margin, slice_bytes=1, min_span_margins=0)
    assert sum(len(u.windows) for u in units) == len(windows)
    spanned = sum(1 for u in units if voff_at(span, u.lo - margin) == voff_at(span, u.hi + 2 * margin) < span[1]
                  and np.any((table.pos < u.hi) & (ref_end > u.lo)))
    assert spanned >= 1                                       # the case really occurs in this file with this margin
    # the feed's loop (ingest.ChromosomeFeed._run / _decode): a rejected slice -> everything from it on is cut again with twice
    # what it needed; an accepted slice must hold every record the whole file has for its windows
    served, replans = [], 0
    while units:
        u = units.pop(0)
        overlapping = np.flatnonzero((table.pos < u.hi) & (ref_end > u.lo))
        if u.vhi == u.vlo:                                    # only a range nothing of the reference lies behind may be empty
            assert u.vlo == span[1] and overlapping.size == 0, u
            served.append(u)
            continue
        t = _range_table(dec, u)
        assert len(t) > 0
        smp = Sample.with_scan(t, fasta, 50, helpers.oracle_scan(t, 50))
        if not ChromosomeFeed._slice_complete(u, smp):
            replans += 1
            assert replans < 12
            margin = (max(2 * ChromosomeFeed._slice_needs(u, smp), 2 * margin) + 16383) >> 14 << 14
            units = dec.plan_units([0], lambda t: windows, margin=margin, slice_bytes=1, min_span_margins=0, resume=(0, u.lo))
            continue
        for a, b in u.windows:
            want = np.flatnonzero((table.pos < b) & (ref_end > a))
            got = t.fetch(0, a, b)
            assert sorted(t.pos[got].tolist()) == sorted(table.pos[want].tolist()), (u, a, b)
        served.append(u)
    assert sum(len(u.windows) for u in served) == len(windows) and replans >= 1


def test_lazy_cigar_in_the_owner_process_waits_for_the_spill():
    """Round 6: with the hand-over a few ms earlier, the first window's collection in a `-t 1` run could read a table's CIGAR
    words (by-value comparison of duplicated records) before the spill thread had attached them -- RuntimeError, and the window
    was skipped as failed.  A reader in the owner process waits for the spill's event now."""
    import threading
    import time
    from svision_amd.ingest_gpu import LazyCigar
    lazy = LazyCigar(5)
    with pytest.raises(RuntimeError):
        lazy[0]                                               # no event: the words are on the device only, said at once
    lazy = LazyCigar(5)
    lazy.event = threading.Event()

    def spill():
        time.sleep(0.05)
        lazy.attach(np.arange(5, dtype=np.uint32))
        lazy.event.set()
    threading.Thread(target=spill).start()
    assert lazy[3] == 3 and np.asarray(lazy).tolist() == [0, 1, 2, 3, 4]
    failed = LazyCigar(5)
    failed.event = threading.Event()
    failed.event.set()                                        # a spill that failed sets the event without attaching anything
    with pytest.raises(RuntimeError):
        failed[0]
