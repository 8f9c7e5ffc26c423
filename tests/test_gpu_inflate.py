"""svx_bgzf_inflate (one lane per BGZF block on the device) == zlib, byte for byte: stored, fixed-code and dynamic-code
DEFLATE blocks, long matches at every distance, incompressible and highly compressible data, empty blocks, the golden
BAMs and a synthetic HiFi-like one; damaged blocks are reported, never decoded silently."""
import os
import struct
import zlib

import numpy as np
import pytest
import torch

from svision_amd import kernels
from svision_amd.io import bam
from tests import helpers


@pytest.fixture(autouse=True)
def _restore_group_sizes():
    """Some tests shrink / blow up the decoder's group sizes: put the module's constants back behind every test."""
    import svision_amd.ingest_gpu as ig
    saved = {k: getattr(ig, k) for k in ("FIRST_GROUP_BYTES", "PIPE_GROUP_BYTES", "LARGE_GROUP_BYTES", "GROUP_BYTES")}
    yield
    for k, v in saved.items():
        setattr(ig, k, v)

pytestmark = pytest.mark.gpu


def _block(payload, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    cdata = co.compress(payload) + co.flush()
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cdata) + 25) + cdata
            + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))


def _inflate_on_device(raw, wave=False):
    raw = np.frombuffer(raw, np.uint8)
    src_off, src_len, isize, _blk = kernels.bgzf_block_table(raw)
    padded = np.zeros((raw.size + 31) // 16 * 16, np.uint8)
    padded[:raw.size] = raw
    out, status = kernels.bgzf_inflate(torch.from_numpy(padded).cuda(), src_off, src_len, isize, wave=wave)
    return out.cpu().numpy().tobytes(), status.cpu().numpy()


@pytest.mark.parametrize("wave", ["lds", "private", "wave", "fast-lane", "fast-wave"])
def test_every_block_type_and_match_shape(wave):
    rng = np.random.default_rng(3)
    text = (b"ACGTTGCA" * 40 + bytes(rng.integers(0, 256, 300, dtype=np.uint8))) * 20
    far = bytes(rng.integers(0, 256, 32768, dtype=np.uint8))
    payloads = [
        b"", b"A", b"hello, hello, hello, hello", bytes(65280), bytes(rng.integers(0, 256, 65280, dtype=np.uint8)),
        text[:65280], far + far[:32000],                          # matches at the maximum distance
        bytes(rng.integers(65, 69, 60000, dtype=np.uint8)), b"\xff" * 30000 + bytes(rng.integers(0, 4, 30000, dtype=np.uint8)),
        b"ab" * 30000, b"x" * 258 + b"y" + b"x" * 600,           # run-length style overlaps (distance 1, 2), length 258
    ]
    blocks, want = [], []
    for p in payloads:
        for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
            blocks.append(_block(p, level, strategy))
            want.append(p)
    got, status = _inflate_on_device(b"".join(blocks), wave)
    assert not status.any(), status.tolist()
    assert got == b"".join(want)


@pytest.mark.parametrize("wave", ["lds", "private", "wave", "fast-lane", "fast-wave"])
def test_golden_and_synthetic_bams_inflate_like_zlib(tmp_path, wave):
    from svision_amd import synth
    paths = [os.path.join(helpers.GOLDEN, n) for n in ("collect_small.bam", "ont_small.bam", "hash_collect.bam")]
    table, _g, _ = synth.simulate(synth.SimConfig(contigs=[("c1", 400_000)], coverage=20, seed=4), with_genome=False)
    seg = bam.encode_reference_segment(table, seq="random", seed=1)          # libdeflate level 1 blocks, realistic SEQ / QUAL
    p = str(tmp_path / "hifi.bam")
    bam.write_bam_segments(p, table.references, table.lengths, [seg])
    for path in paths + [p]:
        raw = open(path, "rb").read()
        got, status = _inflate_on_device(raw, wave)
        assert not status.any()
        assert got == bam.bgzf_decompress(raw), path


@pytest.mark.parametrize("wave", ["lds", "private", "wave", "fast-lane", "fast-wave"])
def test_damaged_blocks_are_flagged(wave):
    rng = np.random.default_rng(5)
    good = _block(bytes(rng.integers(65, 70, 50000, dtype=np.uint8)))
    bad = bytearray(good)
    for i in range(40, 60):
        bad[i] ^= 0x5a                                            # garbage inside the DEFLATE payload
    short = bytearray(good)
    short[-4:] = struct.pack("<I", 50001)                        # ISIZE says one byte more than the stream holds
    got, status = _inflate_on_device(bytes(good) + bytes(bad) + bytes(short) + good, wave)
    assert status[0] == 0 and status[3] == 0 and status[1] != 0 and status[2] != 0


@pytest.mark.parametrize("wave", ["lds", "private", "wave", "fast-lane", "fast-wave"])
def test_a_damaged_payload_that_keeps_isize_is_caught_by_the_crc(wave):
    """a flipped bit that leaves the DEFLATE stream decodable and ISIZE right (here: inside a stored block)
    went through both engines silently; htslib checks the footer's CRC32 on every block, and so does svx_bgzf_crc32."""
    rng = np.random.default_rng(6)
    payloads = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in (50000, 3, 1, 65280, 4, 4099)]
    blocks = [bytearray(_block(p, level=0)) for p in payloads]            # level 0: stored blocks -- any payload byte may change
    flipped = bytearray(blocks[0])
    flipped[18 + 5 + 777] ^= 0x04                                          # one bit of one data byte (18-byte header, 5-byte stored-block header)
    raw = bytes(blocks[0]) + bytes(flipped) + b"".join(bytes(b) for b in blocks[1:]) + bam._BGZF_EOF
    got, status = _inflate_on_device(raw, wave)
    want = [0, kernels.INFLATE_BAD_CRC] + [0] * (len(blocks) - 1) + [0]
    assert status.tolist() == want
    # and with the check switched off the damaged block passes (what rounds 1-3 did)
    r = np.frombuffer(raw, np.uint8)
    src_off, src_len, isize, _blk = kernels.bgzf_block_table(r)
    padded = np.zeros((r.size + 31) // 16 * 16, np.uint8)
    padded[:r.size] = r
    _out, status = kernels.bgzf_inflate(torch.from_numpy(padded).cuda(), src_off, src_len, isize, wave=wave, crc=False)
    assert not status.cpu().numpy().any()


def test_fast_inflate_leaves_streams_of_tiny_deflate_blocks_to_the_wave_kernel():
    """svx_bgzf_inflate_fast transcodes a block into an LZ sequence stream of at most 1.5 x ISIZE + 1 KB; a BGZF block made of
    hundreds of one-byte DEFLATE blocks (a full flush behind every byte) needs more: the entry point hands exactly those
    blocks to the wave-per-block kernel, and the result is still zlib's."""
    rng = np.random.default_rng(8)
    data = bytes(rng.integers(65, 91, 3000, dtype=np.uint8))
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = b"".join(co.compress(data[i:i + 1]) + co.flush(zlib.Z_FULL_FLUSH) for i in range(len(data))) + co.flush()
    assert zlib.decompress(cdata, -15) == data and len(cdata) > 5 * len(data)
    tiny = (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cdata) + 25) + cdata
            + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
    normal = _block(bytes(rng.integers(65, 70, 40000, dtype=np.uint8)))
    got, status = _inflate_on_device(normal + tiny + normal, "fast")
    assert not status.any()
    assert got == bam.bgzf_decompress(normal + tiny + normal + bam._BGZF_EOF)


def test_fast_inflate_with_its_tokens_kernel_on_a_stream_of_its_own():
    """svx_bgzf_inflate_fast_on: kernel A on one stream, kernel B (and what the caller enqueues behind it) on another -- three
    launches in flight at once, the tokens kernels in a row on the process's "tokens" stream, each launch's tables uploaded on
    its own stream right in front of it: every output is zlib's, and the streams of svision_amd.streams are what they say."""
    from svision_amd import _lib, streams
    rng = np.random.default_rng(11)
    dev = torch.device("cuda:0")
    tok = streams.get("tokens", dev)
    assert streams.get("tokens", dev) is tok and streams.get("scan", dev).priority <= streams.get("copy", dev).priority
    lib = _lib.load()
    jobs = []
    for k, s in enumerate((streams.get("ingest0", dev), streams.get("ingest1", dev), torch.cuda.Stream())):
        payloads = [bytes(rng.integers(65, 69 + k, int(rng.integers(1, 65280)), dtype=np.uint8)) for _ in range(200)]
        raw = np.frombuffer(b"".join(_block(p, level=(1, 6, 9)[k]) for p in payloads), np.uint8)
        src_off, src_len, isize, _blk = kernels.bgzf_block_table(raw)
        padded = np.zeros((raw.size + 31) // 16 * 16, np.uint8)
        padded[:raw.size] = raw
        dst = np.zeros(len(isize) + 1, np.uint64)
        dst[1:] = np.cumsum(isize.astype(np.uint64))
        with torch.cuda.stream(s):
            d_comp = torch.from_numpy(padded).to(dev, non_blocking=True)
            d_src, d_len = torch.from_numpy(src_off.view(np.int64)).to(dev), torch.from_numpy(src_len.view(np.int32)).to(dev)
            d_dst = torch.from_numpy(dst.view(np.int64)).to(dev)
            d_out = torch.empty(int(dst[-1]), dtype=torch.uint8, device=dev)
            d_status = torch.zeros(len(isize), dtype=torch.int32, device=dev)
            ws = kernels.inflate_workspace(lib, "fast", int(dst[-1]), len(isize), dev)
            kernels.launch_inflate(lib, "fast", d_comp.data_ptr(), d_src.data_ptr(), d_len.data_ptr(), d_dst.data_ptr(), len(isize), d_out.data_ptr(),
                                   d_status.data_ptr(), int(dst[-1]), dev, ws=ws, tokens_stream=tok)
        jobs.append((s, d_out, d_status, b"".join(payloads), (d_comp, d_src, d_len, d_dst, ws)))
    for s, d_out, d_status, want, _keep in jobs:
        s.synchronize()
        assert not d_status.cpu().numpy().any()
        assert d_out.cpu().numpy().tobytes() == want


def _same_table(a, b):
    for f in ("tid", "pos", "flag", "mapq", "l_seq", "name_id", "cigar", "cig_off"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert a.names == b.names and a.references == b.references and a.lengths == b.lengths


def test_device_decoder_equals_the_host_decoder(tmp_path):
    """ingest_gpu.DeviceDecoder (read -> upload -> svx_bgzf_inflate -> svx_bam_walk_* through the .bai linear index) gives,
    chromosome by chromosome, the tables of the host decoder, and the packed CIGARs it leaves in HBM are the same words."""
    from svision_amd import synth
    from svision_amd.ingest_gpu import DeviceDecoder, DeviceIngestError
    cfg = synth.SimConfig(contigs=[("c1", 900_000), ("c2", 50_000), ("c3", 600_000), ("c4", 300_000)], coverage=12, read_len_mean=9000,
                          read_len_sd=1500, sv_spacing=20_000, sv_min_gap=9_000, sv_max=3000, seed=9)
    table, _g, _ = synth.simulate(cfg, with_genome=False)
    table = table.subset(np.flatnonzero(table.tid != 1))                # a reference without records
    segs = [bam.encode_reference_segment(table.subset(np.flatnonzero(table.tid == t)), seq="random", seed=t) for t in (0, 2, 3)]
    path = str(tmp_path / "dev.bam")
    bam.write_bam_segments(path, table.references, table.lengths, segs)
    head = bam.read_bam_header(path)
    for first_group in (1 << 10, 1 << 40):                               # every chromosome its own launch / all in one
        import svision_amd.ingest_gpu as ig
        ig.FIRST_GROUP_BYTES = ig.GROUP_BYTES = first_group
        dec = DeviceDecoder(path, path + ".bai", head.references, head.lengths, head.header_text, "cuda:0", threads=3)
        assert dec.usable([0, 1, 2, 3])
        groups = dec.groups([0, 1, 2, 3])
        assert sorted(t for g in groups for t in g) == [0, 2, 3]
        got = {}
        for g in groups:
            for finish, (d_cigar, d_off, d_pos) in dec.decode_group(g):
                tb = finish()
                assert tb.cigar.size == int(d_off[-1].item())
                with pytest.raises(RuntimeError):
                    tb.cigar[0]                                # the words are on the device only ...
                ig.spill_cigar(tb)                             # ... until the feed spills them to the table's slot
                t = int(tb.tid[0])
                got[t] = tb
                assert np.array_equal(d_cigar.cpu().numpy().view(np.uint32)[:tb.cigar.size], tb.cigar)
                assert np.array_equal(d_off.cpu().numpy(), tb.cig_off) and np.array_equal(d_pos.cpu().numpy(), tb.pos)
        for t in (0, 2, 3):
            _same_table(got[t], bam.read_bam(path, tids=[t]))
    # the golden BAM written by the plain writer (own .bai), and an index that does not fit the file
    plain = str(tmp_path / "plain.bam")
    bam.write_bam(plain, bam.read_bam(os.path.join(helpers.GOLDEN, "collect_small.bam")), index=True)
    head = bam.read_bam_header(plain)
    dec = DeviceDecoder(plain, plain + ".bai", head.references, head.lengths, head.header_text, "cuda:0")
    for g in dec.groups([0, 1]):
        for finish, _arrays in dec.decode_group(g):
            tb = finish()
            ig.spill_cigar(tb)
            _same_table(tb, bam.read_bam(plain, tids=[int(tb.tid[0])]))
    wrong = DeviceDecoder(path, plain + ".bai", head.references, head.lengths, head.header_text, "cuda:0")
    with pytest.raises(DeviceIngestError):
        for g in wrong.groups([0, 1]):
            list(wrong.decode_group(g))


def test_command_line_with_device_ingest_equals_host_ingest(tmp_path):
    """SVX_INGEST=gpu: the whole command line (-t 1 and -t 3) with the BGZF blocks inflated and the records packed on the
    device writes the VCF of the default run (host inflate); duplicated records included (their by-value comparison reads the
    CIGAR words the device engine spills to shared memory after the hand-over)."""
    import subprocess
    import sys
    from oracle import alexnet_ref
    from svision_amd.network import tf_checkpoint as ck
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prefix = str(tmp_path / "m.ckpt")
    ck.write_checkpoint(prefix, alexnet_ref.random_params(seed=7))
    for name in ("collect_small", "dup_small"):
        fasta = helpers.load_golden_fasta(name + ".fa.gz")
        fa = str(tmp_path / (name + ".fa"))
        bam.write_fasta(fa, {n: fasta._seq[n] for n in fasta.references})
        path = str(tmp_path / (name + ".bam"))
        bam.write_bam(path, bam.read_bam(os.path.join(helpers.GOLDEN, name + ".bam")), index=True)
        outs = {}
        for engine, t in (("cpu", "1"), ("gpu", "1"), ("gpu", "3")):
            out = str(tmp_path / ("%s_%s_%s" % (name, engine, t)))
            r = subprocess.run([sys.executable, os.path.join(root, "SVision"), "-o", out, "-b", path, "-m", prefix, "-g", fa, "-n", "HGi", "-s", "3",
                                "--window_size", "150000", "--batch_size", "64", "-t", t, "--debug"], capture_output=True, text=True, timeout=600,
                               env=dict(os.environ, PYTHONPATH=root, SVX_INGEST=engine, SVX_TIMING="1"))
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
            assert ("'engine': '%s'" % engine) in r.stdout
            if os.environ.get("SVX_TEST_DUMP"):
                with open(os.path.join(os.environ["SVX_TEST_DUMP"], "%s_%s_%s.log" % (name, engine, t)), "w") as f_:
                    f_.write(r.stdout[-20000:] + "\n---- stderr\n" + r.stderr[-20000:])
                    for lf in os.listdir(out):
                        if lf.endswith(".log"):
                            f_.write("\n---- %s\n" % lf + open(os.path.join(out, lf)).read()[-20000:])
            outs[(engine, t)] = {rel: open(os.path.join(out, rel)).read() for rel in ["HGi.svision.s3.vcf"] +
                                 ["segments/" + f for f in sorted(os.listdir(os.path.join(out, "segments")))]}
        assert outs[("cpu", "1")] == outs[("gpu", "1")] == outs[("gpu", "3")]
        assert outs[("cpu", "1")]["HGi.svision.s3.vcf"].count("\n") > 20


def test_pipelined_device_decoder_equals_the_host_decoder(tmp_path):
    """DeviceDecoder.parts_pipelined (reader thread, several groups in flight on their own streams, one read-back per
    group and per chromosome) == the host decoder, chromosome by chromosome, for several group sizes."""
    from svision_amd import synth
    import svision_amd.ingest_gpu as ig
    cfg = synth.SimConfig(contigs=[("c%d" % i, 260_000 + 70_000 * (i % 3)) for i in range(7)], coverage=10, read_len_mean=9000, read_len_sd=1500,
                          sv_spacing=20_000, sv_min_gap=9_000, sv_max=3000, seed=21)
    table, _g, _ = synth.simulate(cfg, with_genome=False)
    table = table.subset(np.flatnonzero(table.tid != 4))
    tids = [0, 1, 2, 3, 5, 6]
    segs = [bam.encode_reference_segment(table.subset(np.flatnonzero(table.tid == t)), seq="random", seed=t) for t in tids]
    path = str(tmp_path / "pipe.bam")
    bam.write_bam_segments(path, table.references, table.lengths, segs)
    head = bam.read_bam_header(path)
    for first, later in ((1 << 10, 1 << 10), (1 << 20, 3 << 20), (1 << 40, 1 << 40)):
        ig.FIRST_GROUP_BYTES, ig.PIPE_GROUP_BYTES = first, later
        dec = ig.DeviceDecoder(path, path + ".bai", head.references, head.lengths, head.header_text, "cuda:0", threads=3)
        got = []
        for finish, (d_cigar, d_off, d_pos) in dec.parts_pipelined(list(range(7))):
            tb = finish()
            ig.spill_cigar(tb)
            got.append(int(tb.tid[0]))
            _same_table(tb, bam.read_bam(path, tids=[got[-1]]))
            assert np.array_equal(d_off.cpu().numpy(), tb.cig_off) and np.array_equal(d_pos.cpu().numpy(), tb.pos)
        assert got == tids


def _bam_with_a_cg_tag_record(tmp_path):
    """Three references; on the second one a record of 66,000 CIGAR operations (CG:B,I tag + placeholder, SAMv1 4.2.2)."""
    from svision_amd import synth
    cfg = synth.SimConfig(contigs=[("c1", 300_000), ("c2", 200_000), ("c3", 250_000)], coverage=6, read_len_mean=4000, read_len_sd=600,
                          sv_spacing=9000, sv_min_gap=5000, sv_max=1000, seed=21)
    table, genome, _ = synth.simulate(cfg)
    n_ops = 66_000
    ops = np.tile(np.array([7, 8], np.uint32), n_ops // 2)
    words = (np.full(n_ops, 1, np.uint32) << 4) | ops                       # 1=1X1=1X...: 66,000 reference bases
    k = int(np.flatnonzero(table.tid == 1)[0])
    at = int(table.cig_off[k])

    def ins(col, value):
        return np.concatenate([col[:k], np.asarray([value], col.dtype), col[k:]])
    n_cig = np.diff(np.asarray(table.cig_off))
    cig_off = np.zeros(len(table) + 2, np.int64)
    cig_off[1:] = np.cumsum(ins(n_cig, n_ops))
    merged = bam.AlignmentTable(table.references, table.lengths, ins(table.tid, 1), ins(table.pos, int(table.pos[k])), ins(table.flag, 0),
                                ins(table.mapq, 60), ins(table.l_seq, n_ops), ins(table.name_id, len(table.names)),
                                list(table.names) + ["long_cigar_read"], np.concatenate([table.cigar[:at], words, table.cigar[at:]]), cig_off)
    path = str(tmp_path / "cg.bam")
    bam.write_bam(path, merged, index=True)
    return path, genome, n_ops


def _feed_tables(path, genome, engine):
    """Every reference's table (+ scan) through ChromosomeFeed with one ingest engine -> ({reference: fields}, feed.stats)."""
    import time
    from svision_amd import ingest
    head = bam.read_bam_header(path)
    fasta = bam.Fasta(sequences=genome)
    opts = helpers.default_options(min_support=3, batch_size=64, bam_path=path)
    feed = ingest.ChromosomeFeed(path, fasta, opts, head.references, head.references, head.lengths, device=torch.device("cuda:0"),
                                 index=bam.find_index(path), threads=4, engine=engine)
    out = {}
    try:
        for chrom in head.references:
            _key, smp = feed.get(chrom, block=True)
            t = smp.table
            t_end = time.time() + 30                              # a device-decoded table's CIGAR words reach the host a little later (the spill)
            while getattr(t.cigar, "_arr", 0) is None and time.time() < t_end:
                time.sleep(0.005)
            out[chrom] = (t.pos.copy(), t.flag.copy(), t.mapq.copy(), t.l_seq.copy(), np.asarray(t.cig_off).copy(), np.asarray(t.cigar).copy(),
                          [t.names[i] for i in t.name_id], smp.stats.copy(), np.asarray(smp.gap_off).copy(), type(t.cigar).__name__)
            feed.release(chrom)
    finally:
        feed.close()
    return out, dict(feed.stats)


def _same_feed_tables(got, want):
    assert list(got) == list(want)
    for chrom in want:
        for a, b in zip(got[chrom][:-1], want[chrom][:-1]):
            assert (a == b) if isinstance(a, list) else np.array_equal(a, b), chrom


def test_device_engine_follows_a_long_cigar_into_its_cg_tag(tmp_path):
    """A CIGAR of more than 65,535 operations sits in the record's CG:B,I tag: svx_bam_walk_* read it there, like the host
    reader (svx_bam.cpp find_long_cigar); every reference is decoded on the device and equals the host engine's."""
    path, genome, n_ops = _bam_with_a_cg_tag_record(tmp_path)
    got, stats = _feed_tables(path, genome, "gpu")
    want, _ = _feed_tables(path, genome, "cpu")
    assert stats["engine"] == "gpu" and {v[-1] for v in got.values()} == {"LazyCigar"} and {v[-1] for v in want.values()} == {"ndarray"}
    _same_feed_tables(got, want)
    assert "long_cigar_read" in got["c2"][6] and int(np.diff(got["c2"][4]).max()) == n_ops


def test_feed_falls_back_to_the_host_engine_where_the_device_engine_refuses(tmp_path, caplog):
    """An entry of the second reference's LINEAR index that points into the middle of a record: the host engine never reads
    the linear index (it takes a reference's byte range from the bins), the device engine starts a record walk there, finds
    that it does not end on the next entry and refuses the reference -- which then comes from the host reader, the one
    behind it from the device engine again, and all tables are what the host engine alone produces."""
    import logging
    import shutil
    import struct as st
    path, genome, _n_ops = _bam_with_a_cg_tag_record(tmp_path)
    want, _ = _feed_tables(path, genome, "cpu")
    bai = bam.find_index(path)
    raw = bytearray(open(bai, "rb").read())
    at = 8                                                        # magic, n_ref
    for ref in range(2):                                          # walk to reference 1's linear index
        n_bin, = st.unpack_from("<i", raw, at); at += 4
        for _ in range(n_bin):
            _bin, n_chunk = st.unpack_from("<Ii", raw, at); at += 8 + 16 * n_chunk
        n_intv, = st.unpack_from("<i", raw, at); at += 4
        if ref == 0:
            at += 8 * n_intv
    vals = list(st.unpack_from("<%dQ" % n_intv, raw, at))
    k = max(i for i in range(n_intv) if vals[i] and vals[i] != vals[-1])    # an entry in the middle of the reference's records
    st.pack_into("<Q", raw, at + 8 * k, vals[k] + 5)               # five bytes into the record it pointed at
    shutil.copy(path, str(tmp_path / "bad.bam"))
    with open(str(tmp_path / "bad.bam.bai"), "wb") as f:
        f.write(raw)
    with caplog.at_level(logging.WARNING):
        got, stats = _feed_tables(str(tmp_path / "bad.bam"), genome, "gpu")
    assert stats["engine"] == "gpu" and any("on the host" in r.getMessage() for r in caplog.records)
    _same_feed_tables(got, want)
    # who decoded what: the refused reference (and, in its group, the one in front of it) by the host reader, the one behind it
    # by the device engine again
    assert got["c2"][-1] != "LazyCigar" and got["c3"][-1] == "LazyCigar"


def test_device_decoder_on_a_file_with_secondary_records_and_an_unmapped_tail(tmp_path):
    """A coordinate-sorted file ends with its unmapped reads (tid -1, no CIGAR, behind the last reference's records) and holds
    secondary records and records without a CIGAR: the device decoder's tables == the host decoder's, chromosome by chromosome
    (the last chromosome's walk has to end where the unmapped reads begin)."""
    from svision_amd import synth
    import svision_amd.ingest_gpu as ig
    cfg = synth.SimConfig(contigs=[("c1", 200_000), ("c2", 100_000), ("c3", 150_000)], coverage=10, read_len_mean=5000, read_len_sd=800,
                          sv_spacing=6000, sv_min_gap=4000, sv_max=2000, seed=3)
    t, _genome, _ = synth.simulate(cfg, with_genome=False)
    extra = 7
    flag = np.concatenate([t.flag, np.full(extra, 4, np.uint16)])
    flag[[10, 20, 30]] |= 0x100
    t2 = bam.AlignmentTable(t.references, t.lengths, np.concatenate([t.tid, np.full(extra, -1, np.int32)]),
                            np.concatenate([t.pos, np.full(extra, -1, np.int32)]), flag, np.concatenate([t.mapq, np.zeros(extra, np.uint8)]),
                            np.concatenate([t.l_seq, np.full(extra, 100, np.int32)]),
                            np.concatenate([t.name_id, np.arange(len(t.names), len(t.names) + extra, dtype=np.int32)]),
                            list(t.names) + ["unmapped%d" % i for i in range(extra)], t.cigar,
                            np.concatenate([t.cig_off, np.full(extra, t.cig_off[-1], np.int64)]), "")
    path = str(tmp_path / "u.bam")
    bam.write_bam(path, t2, index=True)
    head = bam.read_bam_header(path)
    for first, later in ((1 << 10, 1 << 10), (1 << 40, 1 << 40)):
        ig.FIRST_GROUP_BYTES, ig.PIPE_GROUP_BYTES = first, later
        dec = ig.DeviceDecoder(path, bam.find_index(path), head.references, head.lengths, head.header_text, "cuda:0", threads=3)
        assert dec.usable([0, 1, 2])
        got = []
        for finish, (_d_cigar, d_off, d_pos) in dec.parts_pipelined([0, 1, 2]):
            tb = finish()
            ig.spill_cigar(tb)
            got.append(int(tb.tid[0]))
            _same_table(tb, bam.read_bam(path, tids=[got[-1]]))
            assert np.array_equal(d_off.cpu().numpy(), tb.cig_off) and np.array_equal(d_pos.cpu().numpy(), tb.pos)
        assert got == [0, 1, 2]


def test_contig_mode_command_line_with_cg_tag_cigars_device_ingest_equals_host_ingest(tmp_path):
    """--contig on assembly-like alignments whose CIGARs have more than 65,535 operations (CG:B,I tags): the command line
    with the device ingest engine writes what it writes with the host engine."""
    import subprocess
    import sys
    from oracle import alexnet_ref
    from svision_amd import synth
    from svision_amd.network import tf_checkpoint as ck
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prefix = str(tmp_path / "m.ckpt")
    ck.write_checkpoint(prefix, alexnet_ref.random_params(seed=7))
    cfg = synth.SimConfig(contigs=[("ctgA", 1_500_000), ("ctgB", 900_000)], coverage=2, read_len_mean=600_000, read_len_sd=100_000,
                          err_rate=0.08, sv_spacing=40_000, sv_min_gap=20_000, sv_max=3000, seed=5)
    table, genome, _ = synth.simulate(cfg)
    assert int(np.diff(np.asarray(table.cig_off)).max()) > 65535
    fa = str(tmp_path / "asm.fa")
    bam.write_fasta(fa, genome)
    path = str(tmp_path / "asm.bam")
    bam.write_bam(path, table, index=True)
    outs = {}
    for engine, t in (("cpu", "1"), ("gpu", "1"), ("gpu", "3")):
        out = str(tmp_path / ("out_%s_%s" % (engine, t)))
        r = subprocess.run([sys.executable, os.path.join(root, "SVision"), "-o", out, "-b", path, "-m", prefix, "-g", fa, "-n", "ASM", "--contig",
                            "--batch_size", "64", "-t", t, "--debug"], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, PYTHONPATH=root, SVX_INGEST=engine, SVX_TIMING="1"))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        assert ("'engine': '%s'" % engine) in r.stdout and "on the host" not in r.stderr
        vcf = [f for f in os.listdir(out) if f.endswith(".vcf")]
        outs[(engine, t)] = {rel: open(os.path.join(out, rel)).read() for rel in vcf + ["segments/" + f for f in sorted(os.listdir(os.path.join(out, "segments")))]}
    assert outs[("cpu", "1")] == outs[("gpu", "1")] == outs[("gpu", "3")]
    assert sum(v.count("\n") for k, v in outs[("cpu", "1")].items() if k.startswith("segments/")) > 20
