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


def test_rank_launcher_line_and_the_first_contact_block():
    """`python bench.py --gpus 8 --workload wg` without a launcher: the line it starts its ranks with, their environment, and
    the layout block every N-GPU line carries (rccl_world, per-rank loads and bytes, the imbalance and what
    it caps strong scaling at, ranks that share a device)."""
    b = _bench()
    from svision_amd import dist as sdist
    cmd = b.rank_command(8, 29517, ["--gpus", "8", "--workload", "wg", "--steps", "40"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29517" and cmd[-6:] == ["--gpus", "8", "--workload", "wg", "--steps", "40"]
    assert cmd[cmd.index("29517") + 1].endswith("bench.py")
    assert b.rank_environment({})["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and b.rank_environment({"HSA_ENABLE_IPC_MODE_LEGACY": "1"})["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"
    # the whole genome on 8 ranks: LPT loads, max / mean = 1.036 -> at most 7.72 x of 8
    shards = sdist.shard_chromosomes([n for n, _l in b.GRCH38], [l for _n, l in b.GRCH38], 8)
    rank_mb = [sum(dict(b.GRCH38)[c] for c in sh) / 1e6 for sh in shards]
    ids = [("node0", "GPU-%02d" % r) for r in range(8)]
    fc = b.first_contact(8, True, "nccl", rank_mb, [10 ** 9] * 8, ids)
    assert fc["rccl_world"] == 8 and fc["ranks_sharing_a_device"] == [] and len(fc["devices"]) == 8
    assert abs(fc["imbalance"] - 1.036) < 5e-4 and abs(fc["strong_scaling_cap"] - 7.72) < 6e-3 and len(fc["rank_mb"]) == 8
    assert fc["rank_bam_bytes"] == [10 ** 9] * 8
    # two ranks that resolved to one device are named (under nccl dist.init_from_env refuses them outright)
    ids[5] = ids[2]
    assert b.first_contact(8, True, "gloo", rank_mb, None, ids)["ranks_sharing_a_device"] == [2, 5]
    assert b.first_contact(8, True, "gloo", rank_mb, None, ids)["rccl_world"] == 0
    assert sdist.duplicate_devices(ids) == {("node0", "GPU-02"): [2, 5]}


def test_more_ranks_than_devices_is_refused_under_nccl(monkeypatch):
    import pytest
    import torch
    from svision_amd import dist as sdist
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("LOCAL_RANK", "3")
    monkeypatch.setenv("SVX_DIST_BACKEND", "nccl")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    with pytest.raises(RuntimeError, match="8 ranks on this node but 2 visible GPU"):
        sdist.init_from_env()


def test_stdout_line_is_compact_strict_json_whatever_the_record_holds():
    """a 29 KB line (decoder traces inside) was not parsed by the driver.  The stdout line is built from the full
    record by bench.compact_line: < 4 KB, strict JSON, the contract's keys + roofline + cpu_baseline + parity verdict."""
    import json
    b = _bench()
    with open(os.path.join(ROOT, "profiles", "r05_bench_steps20_a.json")) as f:
        full = json.load(f)                                   # a real full record of round 5 (29 KB)
    assert len(json.dumps(full)) > 20000
    full["value_repeats"] = {"n": 3, "min": 3500.0, "median": 3600.0, "max": 3700.0, "seconds": [0.39, 0.38, 0.40]}
    full["parity_check"] = {"ok": True, "path": "file-inclusive", "windows": 16, "tsv_equal": True, "sites_equal": True, "tsv_lines": 29000,
                            "sites": 1100, "images_compared": 24576, "max_softmax_delta": 3.1e-6, "softmax_tol": 1e-3, "argmax_differs": 0,
                            "per_window": [{"window": ["chr1", 0, 10000000], "lines": 1800}] * 16}
    full["roofline"]["traffic"] = float("nan")                # a NaN must not reach the line (json.dumps would print a bare NaN)
    full["config"]["workload"] = full["config"]["workload"] * 20
    text = b.compact_line(full, "bench_detail.json")
    assert len(text) < 4096 and "\n" not in text
    line = json.loads(text, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))      # strict: no NaN / Infinity
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "parity_check", "value_repeats"):
        assert key in line, key
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"] and line["steps"] == full["steps"]
    for key in ("workload", "timed_region", "batch", "windows", "sites_per_step", "images_per_site", "resident_sites_per_s"):
        assert key in line["config"], key
    for key in ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_stage_alone", "frac_algorithmic", "traffic"):
        assert key in line["roofline"], key
    assert line["roofline"]["traffic"] is None and line["roofline"]["bound"] == "mfma"
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-9
    assert set(line["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"}
    assert "per_window" not in line["parity_check"] and line["parity_check"]["ok"] is True
    assert not any(k in line for k in ("e2e", "e2e_cold_cache", "roofline_kernels"))       # those live in the detail file only


def test_median_leg_and_repeats_summary():
    b = _bench()
    legs = [{"seconds": 0.40, "sites": 1400}, {"seconds": 0.36, "sites": 1400}, {"seconds": 0.38, "sites": 1400}]
    assert b.median_leg(legs) is legs[2]
    rep = b.repeats_summary(legs, lambda l: l["sites"] / l["seconds"])
    assert rep["n"] == 3 and rep["min"] < rep["median"] < rep["max"] and rep["median"] == 1400 / 0.38
    assert b.median_leg(legs[:1]) is legs[0] and b.median_leg(legs[:2]) is legs[0]         # upper median of an even count = the slower one


def test_parity_check_compares_digests_and_softmax():
    import hashlib
    b = _bench()
    tsv = "chr1+10+20+5\tA\n" "chr1+10+20+5\tB\n" "chr1+90+99+7\tC\n"
    regions = ["chr1+10+20+5", "chr1+90+99+7"]
    probs = np.full((3, 5), 0.2, np.float32)
    gpu = {("chr1", 0, 100): (tsv, np.zeros(3, np.int64), probs)}
    cpu = [{"window": ("chr1", 0, 100), "tsv_sha": hashlib.sha256(tsv.encode()).hexdigest(),
            "sites_sha": hashlib.sha256("\n".join(regions).encode()).hexdigest(), "n_lines": 3, "index": np.array([0, 2]),
            "classes": np.zeros(2, np.int64), "probs": probs[[0, 2]] + np.float32(5e-4)}]
    out = b.parity_check(gpu, cpu, "resident")
    assert out["ok"] and out["tsv_equal"] and out["sites_equal"] and out["windows"] == 1 and out["images_compared"] == 2
    assert 4e-4 < out["max_softmax_delta"] < 6e-4 and out["sites"] == 2 and out["tsv_lines"] == 3
    cpu[0]["probs"] = probs[[0, 2]] + np.float32(2e-3)
    assert not b.parity_check(gpu, cpu, "resident")["ok"]                                  # beyond the 1e-3 the north star allows
    cpu[0]["probs"] = probs[[0, 2]]
    cpu[0]["tsv_sha"] = "0" * 64
    bad = b.parity_check(gpu, cpu, "resident")
    assert not bad["ok"] and not bad["tsv_equal"] and bad["sites_equal"]
    assert not b.parity_check({}, cpu, "resident")["ok"]                                   # a window the device leg never ran


def test_wg_steps_are_windows_per_rank():
    """Round 6: `--steps K` of the wg workload = K windows per rank (weak scaling: the path partitions by chromosome with no
    data-path collective) -- the job is exactly K x N windows, LPT-sharded; at K x N >= 322 it is the genome whatever N (strong)."""
    from svision_amd import dist as sdist
    b = _bench()
    for k, world in ((20, 1), (20, 2), (20, 8), (5, 8), (40, 8)):
        contigs, strong = b.wg_job(k, world)
        n = sum(len(b.windows_of(name, length)) for name, length in contigs)
        assert n == k * world and not strong, (k, world, n)
        shards = sdist.shard_chromosomes([c for c, _l in contigs], [l for _c, l in contigs], world)
        loads = [sum(len(b.windows_of(c, dict(contigs)[c])) for c in sh) for sh in shards]
        assert sum(loads) == k * world and min(loads) >= 1
        assert max(loads) <= 1.5 * k + 1                       # chromosomes are not divisible: LPT keeps the ranks within half a job of K
    for k, world in ((None, 1), (None, 8), (322, 1), (100, 4), (50, 8)):
        contigs, strong = b.wg_job(k, world)
        assert strong and contigs == list(b.GRCH38)
