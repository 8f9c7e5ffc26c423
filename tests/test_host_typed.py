"""The statically typed forms of the host collection (augmenting .pxd files, svision_amd/build_host.py) against their plain
NumPy / Python forms.  The golden and fuzz fixtures pin the results of the whole step against the reference; these pin the
two implementations of one function against each other on inputs the fixtures do not hold (zero spans, ties, huge values)."""
import numpy as np
import pytest

from svision_amd.collection import cluster_signatures as cs


def _numpy_form(starts, ends):
    saved = cs._COMPILED
    cs._COMPILED = False
    try:
        return cs.span_position_distance_condensed(starts, ends)
    finally:
        cs._COMPILED = saved


@pytest.mark.skipif(not cs._COMPILED, reason="host modules not compiled: only the NumPy form exists")
def test_condensed_distance_loops_equal_the_numpy_form_bit_for_bit():
    rng = np.random.default_rng(5)
    cases = [([10, 10, 10], [10, 10, 20]),                    # zero spans: 0/0 -> nan, x/0 never (max > 0) ...
             ([5, 7], [5, 7]),                                 # ... both zero: nan
             ([0, 3], [0, 9]), ([1], [2]), ([], []),
             ([2_000_000_000, 2_147_483_000, 17], [2_147_483_647, 2_147_483_647, 4_000_000_000])]
    for _ in range(200):
        n = int(rng.integers(2, 80))
        s = rng.integers(0, 250_000_000, n)
        e = s + rng.integers(0, 5, n) * rng.integers(0, 100_000, n)          # many zero spans and equal values
        cases.append((s.tolist(), e.tolist()))
    two_d = np.sort(rng.integers(0, 1000, (40, 2)), axis=1).astype(np.float64)
    cases.append((two_d[:, 0], two_d[:, 1]))                                 # strided views
    for starts, ends in cases:
        got = cs.span_position_distance_condensed(starts, ends)
        want = _numpy_form(starts, ends)
        assert got.dtype == want.dtype == np.float64 and got.shape == want.shape
        assert got.tobytes() == want.tobytes(), (starts, ends)               # bit for bit, NaNs included


def test_segments_are_extension_types_when_compiled_and_compare_by_value():
    from svision_amd.collection.classes import Seg
    a = Seg(1, 5, 100, 104, 0, False, qual=60, aln=-1)
    b = a.copy()
    assert a is not b and a.same_value(b) and b.same_value(a)
    b.ref_end += 1
    assert not a.same_value(b)
    with pytest.raises((AttributeError, TypeError)):
        a.no_such_field = 1                                                  # __slots__ interpreted, C struct compiled
    big = Seg(0, 2**40, 2**40, 2**41, 3, True)
    assert big.ref_end - big.q_end == 2**40 and big.is_reverse is True


def test_collect_parts_hands_whole_clusters_on_and_drops_a_failing_window(monkeypatch, oracle_lib):
    """pipeline._collect_parts: the parts are the window's lines in order; an exception anywhere makes the window empty."""
    from svision_amd import pipeline
    from tests import helpers
    sample = helpers.golden_sample(50)
    opts = helpers.default_options(min_support=3, batch_size=64, bam_path="<resident>")
    want = pipeline._collect_lines(sample, opts, "chrA", 0, 150_000)
    assert len(want) > 64
    parts = []
    lines, ok = pipeline._collect_parts(sample, opts, "chrA", 0, 150_000, parts.append, granule=32)
    assert ok and len(parts) > 1 and all(len(p) >= 32 for p in parts[:-1])
    flat = [ln for p in parts for ln in p]
    assert [ln.text() for ln in flat] == [ln.text() for ln in want] == [ln.text() for ln in lines]
    real = pipeline.iter_pair_lines

    def failing(clusters, options):
        for i, got in enumerate(real(clusters, options)):
            if i == 3:
                raise ValueError("start out of range (-1)")
            yield got
    monkeypatch.setattr(pipeline, "iter_pair_lines", failing)
    parts = []
    lines, ok = pipeline._collect_parts(sample, opts, "chrA", 0, 150_000, parts.append, granule=8)
    assert (lines, ok) == ([], False) and parts                   # parts left before the failure: the owner drops them


def test_the_same_sources_interpreted_reproduce_the_golden_fixtures():
    """The host modules are compiled with static types from sources that must stay plain Python: with SVX_HOST_INTERPRETED=1
    every one of them is imported from its .py (what a machine without the build runs) and the reference-generated collection,
    fuzz, duplicate-record and vote fixtures must come out the same."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import svision_amd.collection.classes as c, svision_amd.network.genotype as g, sys; "
            "assert c.__file__.endswith('.py') and g.__file__.endswith('.py'), (c.__file__, g.__file__); "
            "import pytest; sys.exit(pytest.main(['-x', '-q', '-m', 'not gpu', '-p', 'no:cacheprovider', "
            "'tests/test_collection_golden.py', 'tests/test_fuzz_golden.py', 'tests/test_dup_golden.py', 'tests/test_boundary_golden.py', "
            "'tests/test_predict_golden.py']))")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, SVX_HOST_INTERPRETED="1", PYTHONPATH=root))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_window_vote_fed_in_chunks_equals_the_vote_in_one_go(oracle_lib):
    """pipeline.WindowVote: the predictions of a window arrive launch by launch; any chunking gives the vote of the whole."""
    from svision_amd import pipeline
    from tests import helpers
    sample = helpers.golden_sample(50)
    opts = helpers.default_options(min_support=3, batch_size=64, bam_path="<resident>")
    rng = np.random.default_rng(11)
    for chrom, start, end in (("chrA", 0, 150_000), ("chrA", 150_000, 300_000), ("chrB", 0, 200_000)):
        lines = pipeline._collect_lines(sample, opts, chrom, start, end)
        assert len(lines) > 40
        probs = rng.random((len(lines), 5)).astype(np.float32)
        probs /= probs.sum(1, keepdims=True)
        classes = probs.argmax(1).astype(np.int64)
        want = pipeline._vote(sample, opts, chrom, lines, classes, probs, start, end)
        for sizes in ([1] * len(lines), [7, 64, 1, 256], [len(lines)]):
            vote = pipeline.WindowVote(sample, opts, chrom, lines, start, end)
            at, k = 0, 0
            while at < len(lines):
                n = min(sizes[k % len(sizes)], len(lines) - at)
                vote.feed(classes[at:at + n], probs[at:at + n])
                vote.feed(classes[:0], probs[:0])                 # an empty chunk changes nothing
                at += n
                k += 1
            assert vote.finish() == want
    empty = pipeline.WindowVote(sample, opts, "chrA", [], 0, 10)
    assert empty.finish() == ("", "", 0, None, None)
    short = pipeline.WindowVote(sample, opts, "chrA", lines[:10], 0, 150_000)
    short.feed(classes[:4], probs[:4])
    with pytest.raises(RuntimeError):
        short.finish()                                            # closed before all predictions arrived


def test_image_queue_cuts_launch_groups_across_windows_and_maps_them_back():
    """pipeline.ImageQueue: parts of several windows in arrival order -> groups of any size; the mapping puts every image's
    prediction back at its place in its window."""
    from svision_amd.pipeline import ImageQueue
    rng = np.random.default_rng(3)
    q = ImageQueue()
    windows = {}                                                  # wid -> all its records, in order
    arrivals = [(0, 300), (1, 5), (0, 17), (2, 700), (1, 256), (0, 1)]
    for t, (wid, n) in enumerate(arrivals):
        recs = rng.integers(0, 1000, (n, 12)).astype(np.int32)
        q.add(wid, recs, now=float(t))
        windows[wid] = recs if wid not in windows else np.concatenate([windows[wid], recs])
    q.add(3, np.empty((0, 12), np.int32), now=9.0)                # an empty part is nothing
    total = sum(n for _w, n in arrivals)
    assert q.images == total and q.oldest() == 0.0
    back = {wid: np.zeros_like(r) for wid, r in windows.items()}
    seen = 0
    for n in (256, 256, 64, 512, 1, total - 1089):
        records, mapping = q.take(n)
        assert records.shape == (n, 12) and sum(k for *_x, k in mapping) == n
        for wid, w_off, g_off, k in mapping:
            back[wid][w_off:w_off + k] = records[g_off:g_off + k]
        seen += n
        assert q.images == total - seen
    assert q.images == 0 and q.oldest() is None
    for wid in windows:
        assert np.array_equal(back[wid], windows[wid])
    with pytest.raises(ValueError):
        q.take(1)
    # a window whose collection failed: what is queued of it goes, the others stay in order
    q = ImageQueue()
    a, b, c = (rng.integers(0, 9, (n, 12)).astype(np.int32) for n in (10, 20, 30))
    q.add(7, a); q.add(8, b); q.add(7, c)
    q.drop(7)
    assert q.images == 20
    records, mapping = q.take(20)
    assert np.array_equal(records, b) and mapping == [(8, 0, 0, 20)]
