"""Slices of a chromosome through the device ingest engine (-m gpu): svision_amd/ingest_gpu.py units_pipelined / plan_units,
ingest.ChromosomeFeed with ``tasks``.  The semantics -- a window collected on its slice == the window collected on the whole
chromosome -- are tested without a GPU in tests/test_slices_cpu.py; here: the device decodes exactly the file ranges the plan
names, the feed serves every window from a complete slice (also after a margin guess that was too small), and the command
line writes the same files with and without slices."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from svision_amd import _lib, ingest, synth
from svision_amd.io import bam
from tests import helpers
from tests.test_e2e_golden import device_model, expected  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WINDOW = 100_000


@pytest.fixture(scope="module")
def sample_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("slices_gpu")
    cfg = synth.SimConfig(contigs=[("chrA", 1_500_000), ("chrB", 250_000), ("chrC", 700_000)], coverage=14, read_len_mean=8000, read_len_sd=1500,
                          err_rate=0.004, sv_spacing=9_000, sv_min_gap=6_000, sv_max=3000, inline_max=1200, seed=78)
    table, genome, _svs = synth.simulate(cfg)
    path, fa = str(d / "s.bam"), str(d / "s.fa")
    segs = [bam.encode_reference_segment(table.subset(np.flatnonzero(table.tid == t)), seq="random", seed=t) for t in range(3)]
    bam.write_bam_segments(path, table.references, table.lengths, segs)      # the bench's writer: 64 KB blocks, records straddle them
    bam.write_fasta(fa, genome)
    return path, fa, table, genome


def _windows(length):
    return [(a, min(length, a + WINDOW)) for a in range(0, length, WINDOW)]


def _range_table(path, vlo, vhi):
    lib = _lib.load()
    h = lib.svx_bam_open_range(path.encode(), 2, 0, vlo, vhi)
    assert h, lib.svx_bam_error().decode()
    return bam._table_from_handle(lib, h, False)


def _same(a, b):
    for f in ("tid", "pos", "flag", "mapq", "l_seq", "cig_off"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert np.array_equal(np.asarray(a.cigar), np.asarray(b.cigar))
    assert [a.names[i] for i in a.name_id] == [b.names[i] for i in b.name_id]


def test_device_decodes_the_ranges_of_the_plan(sample_files):
    import svision_amd.ingest_gpu as ig
    path, _fa, _table, _genome = sample_files
    head = bam.read_bam_header(path)
    saved = ig.FIRST_GROUP_BYTES, ig.PIPE_GROUP_BYTES
    try:
        for first, later, slice_bytes in ((1 << 10, 1 << 10, 1), (1 << 20, 2 << 20, 400_000), (1 << 40, 1 << 40, 1)):
            ig.FIRST_GROUP_BYTES, ig.PIPE_GROUP_BYTES = first, later
            dec = ig.DeviceDecoder(path, path + ".bai", head.references, head.lengths, head.header_text, "cuda:0", threads=3)
            units = dec.plan_units([0, 1, 2], lambda t: _windows(head.lengths[t]), slice_bytes=slice_bytes, min_span_margins=0)
            assert len(units) > (20 if slice_bytes == 1 else 5)
            got = []
            for unit, finish, (d_cigar, d_off, d_pos) in dec.units_pipelined(units):
                tb = finish()
                ig.spill_cigar(tb)
                got.append(unit)
                _same(tb, _range_table(path, unit.vlo, unit.vhi))
                assert np.array_equal(d_off.cpu().numpy(), tb.cig_off) and np.array_equal(d_pos.cpu().numpy(), tb.pos)
            assert got == units
    finally:
        ig.FIRST_GROUP_BYTES, ig.PIPE_GROUP_BYTES = saved


def _serve(path, genome, tasks, env):
    """Every window's Sample through ChromosomeFeed -> ({(chrom, start): (table fields, scan)}, stats)."""
    head = bam.read_bam_header(path)
    opts = helpers.default_options(min_support=3, batch_size=64, bam_path=path, window_size=WINDOW)
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        feed = ingest.ChromosomeFeed(path, bam.Fasta(sequences=genome), opts, head.references, head.references, head.lengths,
                                     device=torch.device("cuda:0"), index=bam.find_index(path), threads=4, engine="gpu", tasks=tasks)
        out = {}
        try:
            for chrom, wins in tasks.items():
                for start, end in wins:
                    _key, smp = feed.get(chrom, block=True, start=start)
                    out[(chrom, start)] = smp
            stats = dict(feed.stats)
        finally:
            feed.close()
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    return out, stats


def _complete(whole, scan_span, smp, chrom, start, end):
    """Does the window's Sample hold every record of the whole table that overlaps [start - reach, end + reach)?"""
    tid = whole.get_tid(chrom)
    rows = np.flatnonzero(whole.tid == tid)
    pos = whole.pos[rows].astype(np.int64)
    ref_end = pos + np.maximum(scan_span[rows], 1)
    reach = smp.reach()
    need = rows[(pos < end + reach) & (ref_end > start - reach)]
    t = smp.table
    if need.size == 0:
        return True
    # the slice is a contiguous run of the chromosome's records: find it by its first record
    first = int(np.flatnonzero((whole.pos[rows] == t.pos[0]) & (whole.flag[rows] == t.flag[0]) & (whole.l_seq[rows] == t.l_seq[0]))[0])
    assert np.array_equal(whole.pos[rows][first:first + len(t)], t.pos)
    return rows[first] <= need[0] and need[-1] < rows[first] + len(t)


def test_feed_serves_every_window_from_a_complete_slice(sample_files):
    path, _fa, table, genome = sample_files
    head = bam.read_bam_header(path)
    tasks = {c: _windows(n) for c, n in zip(head.references, head.lengths)}
    span = helpers.oracle_scan(table, 50)[2][:, 0]
    got, stats = _serve(path, genome, tasks, {"SVX_SLICE_BYTES": "200000", "SVX_SLICE_MIN_MARGINS": "0"})
    assert stats["engine"] == "gpu" and stats["slices"] > 8 and stats["replans"] == 0
    assert len({id(s) for s in got.values()}) == stats["slices"]
    for (chrom, start), smp in got.items():
        assert len(smp.table) < 0.7 * int((table.tid == table.get_tid(chrom)).sum()) or chrom == "chrB"
        assert _complete(table, span, smp, chrom, start, min(start + WINDOW, head.lengths[head.references.index(chrom)]))
    # a guess that is too small: noticed on the first slice, everything behind it cut again
    got2, stats2 = _serve(path, genome, tasks, {"SVX_SLICE_BYTES": "200000", "SVX_SLICE_MIN_MARGINS": "0", "SVX_SLICE_MARGIN": "1"})
    assert stats2["replans"] >= 1
    for (chrom, start), smp in got2.items():
        assert _complete(table, span, smp, chrom, start, min(start + WINDOW, head.lengths[head.references.index(chrom)]))
    # whole chromosomes (no tasks): one part each
    got3, stats3 = _serve(path, genome, {c: [(0, n)] for c, n in zip(head.references, head.lengths)}, {})
    assert stats3["slices"] == 3


def _cli(args, env=None, timeout=900):
    return subprocess.run([sys.executable, os.path.join(ROOT, "SVision")] + args, capture_output=True, text=True, timeout=timeout,
                          env=dict(os.environ, PYTHONPATH=ROOT, **(env or {})))


def test_command_line_with_slices_equals_whole_chromosomes(sample_files, device_model, tmp_path):
    path, fa, _table, _genome = sample_files
    outs = {}
    for name, t, env in (("whole", "1", {"SVX_SLICES": "0"}), ("sliced", "1", {"SVX_SLICE_BYTES": "200000", "SVX_SLICE_MIN_MARGINS": "0"}),
                         ("sliced pooled", "3", {"SVX_SLICE_BYTES": "200000", "SVX_SLICE_MIN_MARGINS": "0"}), ("cut again", "3", {"SVX_SLICE_BYTES": "200000", "SVX_SLICE_MIN_MARGINS": "0", "SVX_SLICE_MARGIN": "1"}),
                         ("host engine", "1", {"SVX_INGEST": "cpu"})):
        out = str(tmp_path / name.replace(" ", "_"))
        r = _cli(["-o", out, "-b", path, "-m", device_model, "-g", fa, "-n", "HGs", "-s", "3", "--window_size", str(WINDOW), "--batch_size", "64",
                  "--qname", "--debug", "-t", t], env=dict(env, SVX_TIMING="1"))
        assert r.returncode == 0, r.stdout + r.stderr
        if name.startswith("sliced"):
            assert "'slices': " in r.stdout and "'slices': 3," not in r.stdout, r.stdout
        outs[name] = {rel: open(os.path.join(out, rel)).read() for rel in ["HGs.svision.s3.vcf"] +
                      ["segments/" + f for f in sorted(os.listdir(os.path.join(out, "segments")))] +
                      ["predict_results/" + f for f in sorted(os.listdir(os.path.join(out, "predict_results")))]}
    assert outs["whole"]["HGs.svision.s3.vcf"].count("\n") > 60
    for name in outs:
        assert outs[name] == outs["whole"], name


def test_a_slice_the_device_refuses_is_served_by_the_host_engine(sample_files, device_model, tmp_path, caplog):
    """An entry of chrA's LINEAR index that points into the middle of a record, half-way through the chromosome: the slices in
    front of it come from the device engine, the slice that walks into it is refused, and the host reader's table of the whole
    chromosome serves every window that had not been handed over; chrB and chrC come from the device engine again.  Every window
    still sees every record it can touch, and the command line writes what it writes from an intact index."""
    import logging
    import shutil
    import struct as st
    path, fa, table, genome = sample_files
    head = bam.read_bam_header(path)
    raw = bytearray(open(path + ".bai", "rb").read())
    at = 8
    n_bin, = st.unpack_from("<i", raw, at); at += 4
    for _ in range(n_bin):
        _bin, n_chunk = st.unpack_from("<Ii", raw, at); at += 8 + 16 * n_chunk
    n_intv, = st.unpack_from("<i", raw, at); at += 4
    vals = list(st.unpack_from("<%dQ" % n_intv, raw, at))
    k = n_intv // 2
    while vals[k] == vals[k - 1] or vals[k] == 0:
        k += 1
    st.pack_into("<Q", raw, at + 8 * k, vals[k] + 5)               # five bytes into the record it pointed at
    bad = str(tmp_path / "bad.bam")
    shutil.copy(path, bad)
    with open(bad + ".bai", "wb") as f:
        f.write(raw)
    tasks = {c: _windows(n) for c, n in zip(head.references, head.lengths)}
    span = helpers.oracle_scan(table, 50)[2][:, 0]
    import svision_amd.ingest_gpu as ig
    saved = ig.FIRST_GROUP_BYTES, ig.PIPE_GROUP_BYTES, ig.LARGE_GROUP_BYTES
    ig.FIRST_GROUP_BYTES = ig.PIPE_GROUP_BYTES = ig.LARGE_GROUP_BYTES = 1 << 20      # several inflate launches per chromosome: the refusal comes late
    try:
        with caplog.at_level(logging.WARNING):
            got, stats = _serve(bad, genome, tasks, {"SVX_SLICE_BYTES": "200000", "SVX_SLICE_MIN_MARGINS": "0"})
    finally:
        ig.FIRST_GROUP_BYTES, ig.PIPE_GROUP_BYTES, ig.LARGE_GROUP_BYTES = saved
    assert stats["engine"] == "gpu" and any("on the host" in r.getMessage() for r in caplog.records)
    kinds = {}
    for (chrom, start), smp in got.items():
        assert _complete(table, span, smp, chrom, start, min(start + WINDOW, head.lengths[head.references.index(chrom)])), (chrom, start)
        kinds.setdefault(chrom, []).append(type(smp.table.cigar).__name__)
    assert kinds["chrA"][0] == "LazyCigar" and kinds["chrA"][-1] != "LazyCigar"      # device slices first, the host's table behind them
    assert set(kinds["chrC"]) == {"LazyCigar"}
    outs = {}
    for name, b in (("intact", path), ("damaged", bad)):
        out = str(tmp_path / name)
        r = _cli(["-o", out, "-b", b, "-m", device_model, "-g", fa, "-n", "HGs", "-s", "3", "--window_size", str(WINDOW), "--batch_size", "64",
                  "--qname", "-t", "3"], env={"SVX_SLICE_BYTES": "200000", "SVX_SLICE_MIN_MARGINS": "0", "SVX_FIRST_GROUP_MB": "1", "SVX_PIPE_GROUP_MB": "1",
                                              "SVX_LARGE_GROUP_MB": "1"})
        assert r.returncode == 0, r.stdout + r.stderr
        outs[name] = open(os.path.join(out, "HGs.svision.s3.vcf")).read()
    assert outs["damaged"] == outs["intact"] and outs["intact"].count("\n") > 60
