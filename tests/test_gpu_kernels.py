"""GPU parity tests (-m gpu): HIP kernels through the C ABI vs the oracle, bit-exact."""
import numpy as np
import pytest
import torch

from oracle import cigar_ref, encode_ref
from svision_amd import _lib, kernels
from tests import datagen

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(DEV) if dtype is None else t.to(DEV, dtype)


@pytest.mark.parametrize("layout", ["NHWC", "NCHW"])
@pytest.mark.parametrize("n", [1, 3, 64, 130, 257, 600])       # (130: two slices per image, more items than resident workgroups; 600: several images per workgroup)
def test_rasterize_matches_oracle(oracle_lib, layout, n):
    from oracle import cbind
    rec = datagen.random_records(n, seed=100 + n)
    got = kernels.rasterize(_dev(rec), layout=layout).cpu().numpy()
    want = cbind.rasterize(rec, layout)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_rasterize_python_oracle_and_pad_record():
    rec = np.concatenate([datagen.random_records(40, seed=7), np.asarray([encode_ref.PAD_RECORD], np.int32)])
    got = kernels.rasterize(_dev(rec), layout="NHWC").cpu().numpy()
    assert np.array_equal(got, encode_ref.encode_records(rec))


def test_rasterize_large_batch_properties(oracle_lib):
    """Full-size batch (4096 images = 2.5 GB): spot-check images against the oracle and
    check size-independent properties (value set, ch1 subset of ch0, ch2 subset of ch0)."""
    from oracle import cbind
    n = 4096
    rec = datagen.random_records(n, seed=5)
    out = kernels.rasterize(_dev(rec), layout="NCHW")
    m = torch.tensor(kernels.MEAN, device=DEV).view(1, 3, 1, 1)
    mask = out + m
    assert torch.all((mask == 0) | (mask == 255))
    on = mask > 0
    assert torch.all(on[:, 0] | ~on[:, 1]) and torch.all(on[:, 0] | ~on[:, 2])
    colcnt = on[:, 0].sum(dim=1, keepdim=True)            # [n,1,227]
    assert torch.equal(on[:, 1], on[:, 0] & (colcnt >= 2))
    idx = np.arange(0, n, 97)
    assert np.array_equal(out[idx].cpu().numpy(), cbind.rasterize(rec[idx], "NCHW"))


def test_rasterize_empty():
    out = kernels.rasterize(torch.empty((0, 12), dtype=torch.int32, device=DEV))
    assert out.shape == (0, 3, 227, 227)


SCAN_MODES = ["groups4", "groups8", "groups4s", "groups8s", "flat"]      # svx_cigar_scan (four / eight lanes per alignment in the count pass, three launches) / svx_cigar_scan_flat (one pass over chunks of words)


@pytest.mark.parametrize("mode", SCAN_MODES)
@pytest.mark.parametrize("n_aln,mean_ops,rate", [(1, 5, 0.5), (64, 30, 0.2), (5000, 300, 0.01), (300, 5000, 0.002)])
def test_cigar_scan_matches_oracle(oracle_lib, n_aln, mean_ops, rate, mode):
    from oracle import cbind
    cigar, off, ref_start = datagen.random_cigars(n_aln, seed=n_aln, mean_ops=mean_ops, long_gap_rate=rate)
    res = kernels.cigar_scan(_dev(cigar.view(np.int32)), _dev(off.astype(np.int64)), _dev(ref_start), 50, mode=mode)
    gaps, gap_off, stats = res.to_host()
    o_gaps, o_off, o_stats = cbind.cigar_scan(cigar, off, ref_start, 50)
    assert np.array_equal(gap_off, o_off)
    assert np.array_equal(stats, o_stats)
    assert gaps.tobytes() == o_gaps.tobytes()


def test_sample_from_device_reads_the_scan_back_through_pinned_memory_and_grows_its_capacity(oracle_lib):
    """Sample.from_device (device ingest): the scan's result comes back through a pinned scratch buffer on the caller's
    stream; more long gaps than the first guess of the capacity (n / 2, at least 1024) -> one more scan with the exact one.
    Same arrays as the C oracle's, both ways."""
    from oracle import cbind
    from svision_amd.io.bam import AlignmentTable
    from svision_amd.sample import Sample
    for n_aln, mean_ops, rate in ((40, 4000, 0.05), (3000, 100, 0.001)):       # ~8,000 gaps for a capacity of 1,024; and a few
        cigar, off, ref_start = datagen.random_cigars(n_aln, seed=7 + n_aln, mean_ops=mean_ops, long_gap_rate=rate)
        o_gaps, o_off, o_stats = cbind.cigar_scan(cigar, off, ref_start, 50)
        table = AlignmentTable(["c"], [10 ** 9], np.zeros(n_aln, np.int32), ref_start, np.zeros(n_aln, np.uint16), np.full(n_aln, 60, np.uint8),
                               np.zeros(n_aln, np.int32), np.arange(n_aln, dtype=np.int32), ["r%d" % i for i in range(n_aln)], cigar, off.astype(np.int64), "")
        with torch.cuda.stream(torch.cuda.Stream()):
            sample = Sample.from_device(table, None, 50, _dev(cigar.view(np.int32)), _dev(off.astype(np.int64)), _dev(ref_start))
        assert (n_aln == 40) == (int(o_off[-1]) > 1024)
        assert np.array_equal(sample.gap_off, o_off.astype(np.int64))
        assert np.array_equal(sample.stats, o_stats)
        assert sample.gaps.tobytes() == o_gaps.tobytes()


@pytest.mark.parametrize("mode", SCAN_MODES)
def test_cigar_scan_on_a_window_of_a_larger_array(oracle_lib, mode):
    """The offsets of a window's rows point into the chromosome's whole word array (Sample.rescan_window_async): d_cig_off[0] > 0,
    not a multiple of four, the words in front of and behind the window belong to other alignments."""
    from oracle import cbind
    cigar, off, ref_start = datagen.random_cigars(900, seed=31, mean_ops=700, long_gap_rate=0.01, lognormal_sigma=1.0)
    d_cigar, d_off, d_pos = _dev(cigar.view(np.int32)), _dev(off.astype(np.int64)), _dev(ref_start)
    for lo, hi in ((0, 900), (1, 2), (137, 612), (899, 900), (450, 450 + 3)):
        res = kernels.cigar_scan(d_cigar, d_off[lo:hi + 1], d_pos[lo:hi], 50, mode=mode)
        gaps, gap_off, stats = res.to_host()
        sub = cigar[int(off[lo]):int(off[hi])]
        o_gaps, o_off, o_stats = cbind.cigar_scan(sub, (off[lo:hi + 1] - off[lo]).astype(np.uint64), ref_start[lo:hi], 50)
        assert np.array_equal(gap_off, o_off) and np.array_equal(stats, o_stats) and gaps.tobytes() == o_gaps.tobytes(), (lo, hi)


@pytest.mark.parametrize("mode", SCAN_MODES)
def test_cigar_scan_full_size(oracle_lib, mode):
    """A whole-chromosome-sized batch (1.5 M alignments: several scan steps of 1024 tiles, a last partial tile, a work list
    of tens of thousands of alignments) against the C oracle, plus the size-independent properties of the output."""
    from oracle import cbind
    n_aln = 1_500_003
    cigar, off, ref_start = datagen.random_cigars(n_aln, seed=11, mean_ops=24, long_gap_rate=0.004)
    res = kernels.cigar_scan(_dev(cigar.view(np.int32)), _dev(off.astype(np.int64)), _dev(ref_start), 50, mode=mode)
    gaps, gap_off, stats = res.to_host()
    o_gaps, o_off, o_stats = cbind.cigar_scan(cigar, off, ref_start, 50)
    assert np.array_equal(gap_off, o_off)
    assert np.array_equal(stats, o_stats)
    assert gaps.tobytes() == o_gaps.tobytes()
    g = np.frombuffer(gaps.tobytes(), np.int32).reshape(-1, 6)
    assert g.shape[0] == int(gap_off[-1]) > 20_000
    key = g[:, 0].astype(np.int64) * (1 << 32) + g[:, 1].astype(np.uint32)
    assert np.all(np.diff(key) > 0)                                      # sorted by (alignment, op), no duplicates
    assert np.array_equal(np.bincount(g[:, 0], minlength=n_aln), np.diff(gap_off.astype(np.int64)))   # CSR consistent
    words = cigar[off[g[:, 0]].astype(np.int64) + g[:, 1]]
    assert np.array_equal(words >> 4, g[:, 4].astype(np.uint32)) and np.array_equal(words & 15, g[:, 5].astype(np.uint32))
    again = kernels.cigar_scan(_dev(cigar.view(np.int32)), _dev(off.astype(np.int64)), _dev(ref_start), 50, mode=mode)
    assert again.to_host()[0].tobytes() == gaps.tobytes()                # deterministic


@pytest.mark.parametrize("mode", SCAN_MODES)
def test_cigar_scan_ultra_long_reads(oracle_lib, mode):
    """ONT-like op counts (log-normal around 4,000, tail beyond 10^5): most alignments exceed the 512 words the eight-lane count
    pass keeps for itself and are finished by count_long_kernel; mixed with short ones so that count workgroups hold both kinds,
    gaps on either side of the hand-over, lengths around the threshold (511..516 words, 2047..2049)."""
    from oracle import cbind
    cigar, off, ref_start = datagen.random_cigars(6000, seed=77, mean_ops=4000, long_gap_rate=0.002, lognormal_sigma=1.0)
    n_ops = np.diff(off.astype(np.int64))
    assert n_ops.max() > 100_000 and (n_ops > 512).sum() > 4000 and (n_ops <= 512).sum() > 100
    parts = [(cigar, off, ref_start)]
    for i, n in enumerate((511, 512, 513, 514, 515, 516, 2047, 2048, 2049, 1, 3)):
        parts.append(datagen.random_cigars(1, seed=900 + i, mean_ops=n + 200, long_gap_rate=0.05, max_ops=n))
        parts[-1][1][1] = len(parts[-1][0])
    cig = np.concatenate([p[0] for p in parts])
    lens = np.concatenate([np.diff(p[1].astype(np.int64)) for p in parts])
    offs = np.zeros(lens.size + 1, np.uint64)
    offs[1:] = np.cumsum(lens)
    rs = np.concatenate([p[2] for p in parts])
    res = kernels.cigar_scan(_dev(cig.view(np.int32)), _dev(offs.astype(np.int64)), _dev(rs), 50, mode=mode)
    gaps, gap_off, stats = res.to_host()
    o_gaps, o_off, o_stats = cbind.cigar_scan(cig, offs, rs, 50)
    assert np.array_equal(gap_off, o_off)
    assert np.array_equal(stats, o_stats)
    assert gaps.tobytes() == o_gaps.tobytes() and int(gap_off[-1]) > 15_000
    again = kernels.cigar_scan(_dev(cig.view(np.int32)), _dev(offs.astype(np.int64)), _dev(rs), 50, mode=mode)
    assert again.to_host()[0].tobytes() == gaps.tobytes()                # the long list's order varies, the output does not


@pytest.mark.parametrize("mode", SCAN_MODES)
def test_cigar_scan_frame_edges(oracle_lib, mode):
    """Round 5: an alignment of more than 512 words is cut into frames of 512 words behind its first 128 quads, four frames per
    step, and the emit pass walks only frames that hold a long gap, from positions it derives from the frames' sums.  Directed
    cases: long gaps and N ops (read advance without reference advance) on either side of every edge -- head | first frame,
    frame | frame, step | step, the last whole quad, the alignment's last word --, for all four positions of the alignment's
    first word inside its quad, with a neighbour whose first words are long gaps too, and an array that ends inside a quad."""
    from oracle import cbind
    rng = np.random.default_rng(5)
    aligns = []
    for lead in range(4):                                  # 1-op fillers shift the next alignment's first word through the quad
        for n in (512, 513, 516, 517, 1023, 1024, 1025, 1028, 2559, 2560, 2561, 2564, 2565, 4608 + 3, 70_001):
            kinds = rng.choice([0, 7, 8, 1, 2], size=n, p=[0.3, 0.4, 0.1, 0.1, 0.1]).astype(np.uint32)
            lens = rng.integers(1, 30, n).astype(np.uint32)
            edges = [e for e in (0, 1, 2, 3, 508, 509, 510, 511, 512, 513, 515, 516, 1020, 1023, 1024, 1027, 1028, 1535, 1536, 2047, 2048, 2049,
                                 2556, 2559, 2560, 2563, 2564, 4607, 4608, 4609, n - 5, n - 4, n - 3, n - 2, n - 1) if 0 <= e < n]
            for j, e in enumerate(edges):
                if j % 3 == 0: kinds[e], lens[e] = 1 + (j // 3) % 2, 50 + j          # a long I / D
                elif j % 3 == 1: kinds[e], lens[e] = 3, 1000 + j                      # N
            aligns.append((lens << 4 | kinds).astype(np.uint32))
            aligns.append(np.asarray([(60 << 4) | 1, (3 << 4) | 7][:1 + (len(aligns) % 2)], np.uint32))   # a short neighbour that starts with a long gap
        aligns.append(np.asarray([(5 << 4) | 7] * (lead + 1), np.uint32)[:1])
        aligns.append(np.asarray([(7 << 4) | 0], np.uint32))
    aligns.append(np.tile(aligns[0], 3)[:1024 + (2 - sum(len(a) for a in aligns)) % 4].copy())      # the array's last alignment is a long one and ends inside a quad
    off = np.zeros(len(aligns) + 1, np.uint64)
    off[1:] = np.cumsum([len(a) for a in aligns])
    cigar = np.concatenate(aligns)
    assert int(off[-1]) % 4 != 0 and len({int(o) % 4 for o in off[:-1][[len(a) > 512 for a in aligns]]}) == 4
    ref_start = rng.integers(0, 1 << 28, len(aligns)).astype(np.int32)
    padded = np.concatenate([cigar, np.zeros((-cigar.size) % 4, np.uint32)])
    res = kernels.cigar_scan(_dev(padded.view(np.int32)), _dev(off.astype(np.int64)), _dev(ref_start), 50, mode=mode, n_words=int(off[-1]))
    gaps, gap_off, stats = res.to_host()
    o_gaps, o_off, o_stats = cbind.cigar_scan(cigar, off, ref_start, 50)
    assert np.array_equal(gap_off, o_off) and np.array_equal(stats, o_stats)
    assert gaps.tobytes() == o_gaps.tobytes() and int(gap_off[-1]) > 500


def test_cigar_scan_refuses_an_array_longer_than_the_caller_said(oracle_lib):
    """svx_cigar_scan sizes its frame records by n_words: offsets that reach beyond it are answered with SVX_SCAN_FAILED in
    d_gap_off[n_aln] (kernels: SvxError on read-back), never with records written outside the workspace."""
    cigar, off, ref_start = datagen.random_cigars(200, seed=3, mean_ops=3000, long_gap_rate=0.01)
    for mode in ("groups8s", "groups4", "flat"):              # (flat: svx_experimental.h; its chunk map stays inside the workspace too)
        with pytest.raises(_lib.SvxError):
            kernels.cigar_scan(_dev(cigar.view(np.int32)), _dev(off.astype(np.int64)), _dev(ref_start), 50, mode=mode, n_words=int(off[-1]) // 2).total()


@pytest.mark.parametrize("mode", SCAN_MODES)
def test_cigar_scan_ragged_and_empty(oracle_lib, mode):
    from oracle import cbind
    # empty batch
    res = kernels.cigar_scan(torch.empty(0, dtype=torch.int32, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV),
                             torch.empty(0, dtype=torch.int32, device=DEV), 50, mode=mode)
    assert res.total() == 0
    # a batch of empty CIGARs only; empty ones at the very start and the very end
    for texts in (["", "", ""], ["", "", "30M60I", "", "10S", "", ""]):
        ops = [cigar_ref.parse_cigar(t) for t in texts]
        words = [cigar_ref.pack_cigar(o) for o in ops]
        off = np.zeros(len(ops) + 1, np.uint64)
        off[1:] = np.cumsum([len(w) for w in words])
        cigar = np.asarray([w for ws in words for w in ws] + [0, 0, 0, 0], np.uint32)
        ref_start = np.arange(len(ops), dtype=np.int32) * 10 + 3
        res = kernels.cigar_scan(_dev(cigar.view(np.int32)), _dev(off.astype(np.int64)), _dev(ref_start), 50, mode=mode, n_words=int(off[-1]))
        gaps, gap_off, stats = res.to_host()
        o_gaps, o_off, o_stats = cbind.cigar_scan(cigar[:int(off[-1])], off, ref_start, 50)
        assert np.array_equal(gap_off, o_off) and np.array_equal(stats, o_stats) and gaps.tobytes() == o_gaps.tobytes(), texts
    # alignments with empty CIGARs in the middle, all-clip CIGARs, one very long CIGAR
    texts = ["", "10S", "5H10S", "100S2000M300I1500M200D1500M50S", "", "60I", "60D", "49I49D50I50D", "3H7S100M2N5P60I8S2H"]
    ops = [cigar_ref.parse_cigar(t) for t in texts]
    big = [(7, 10), (1, 55), (8, 1), (2, 70)] * 20000
    ops.append(big)
    words = [cigar_ref.pack_cigar(o) for o in ops]
    off = np.zeros(len(ops) + 1, np.uint64)
    off[1:] = np.cumsum([len(w) for w in words])
    cigar = np.asarray([w for ws in words for w in ws], np.uint32)
    ref_start = np.arange(len(ops), dtype=np.int32) * 1000 + 7
    res = kernels.cigar_scan(_dev(cigar.view(np.int32)), _dev(off.astype(np.int64)), _dev(ref_start), 50, mode=mode)
    gaps, gap_off, stats = res.to_host()
    o_gaps, o_off, o_stats = cbind.cigar_scan(cigar, off, ref_start, 50)
    assert np.array_equal(gap_off, o_off) and np.array_equal(stats, o_stats) and gaps.tobytes() == o_gaps.tobytes()
    assert int(gap_off[-1]) == 2 + 1 + 1 + 2 + 1 + 40000
    # capacity overflow is reported, not silently truncated
    small = kernels.cigar_scan(_dev(cigar.view(np.int32)), _dev(off.astype(np.int64)), _dev(ref_start), 50, gaps_cap=16, mode=mode)
    assert small.total() == int(gap_off[-1])
    with pytest.raises(Exception):
        small.to_host()


def test_alexnet_matches_numpy_oracle():
    """Product path (records -> packed softmax) vs the NumPy restatement on the oracle's images; the plain-PyTorch
    restatement on the HIP rasteriser's images agrees with both (north_star tolerance: softmax within 1e-3, fp32)."""
    from oracle import alexnet_ref
    from oracle.alexnet_torch import TorchAlexNet
    from svision_amd.network.alexnet import AlexNet
    params = alexnet_ref.random_params(seed=3)
    rec = datagen.random_records(6, seed=21, hostile=False)
    x = encode_ref.encode_records(rec)
    o_logits, o_cls, o_prob = alexnet_ref.predict(params, x)
    logits, cls, prob = AlexNet(params, device=DEV).predict_records(_dev(rec))
    assert np.abs(prob.cpu().numpy() - o_prob).max() < 1e-3
    assert np.allclose(logits.cpu().numpy(), o_logits, rtol=1e-3, atol=1e-3 * np.abs(o_logits).max())
    img = kernels.rasterize(_dev(rec), layout="NCHW")
    t_logits, _c, t_prob = TorchAlexNet(params, device=DEV).predict(img)
    assert np.abs(t_prob.cpu().numpy() - o_prob).max() < 1e-3
    assert np.abs(t_prob.cpu().numpy() - prob.cpu().numpy()).max() < 1e-3


@pytest.mark.parametrize("shape,lrn", [((3, 96, 55, 55), True), ((2, 256, 27, 27), True), ((4, 256, 13, 13), False), ((1, 8, 9, 11), True)])
def test_bias_relu_pool_lrn_matches_numpy_oracle(shape, lrn):
    """fp32 op: tolerance 1e-5 relative vs the NumPy restatement of relu(x+b) -> max_pool -> tf LRN."""
    from oracle import alexnet_ref
    rng = np.random.default_rng(shape[1])
    x = (rng.standard_normal(shape) * 30).astype(np.float32)
    b = rng.standard_normal(shape[1]).astype(np.float32)
    got = kernels.from_c8(kernels.bias_relu_pool_lrn(kernels.to_c8(_dev(x)), _dev(b), lrn=lrn)).cpu().numpy()
    nhwc = np.maximum(x.transpose(0, 2, 3, 1) + b, 0)
    want = alexnet_ref._max_pool_3x3s2_valid(np.ascontiguousarray(nhwc))
    if lrn:
        want = alexnet_ref._lrn(np.ascontiguousarray(want))
    want = want.transpose(0, 3, 1, 2)
    assert got.shape == want.shape
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6)


def test_encode_conv1_matches_dense_path():
    """Sparse rasterise+conv1+relu+pool+LRN vs the dense path (rasterise -> NumPy oracle layers):
    fp32, 1e-4 relative on the layer output; end to end softmax within 1e-3 of the NumPy AlexNet."""
    from oracle import alexnet_ref
    from svision_amd.network.alexnet import AlexNet
    params = alexnet_ref.random_params(seed=11)
    rec = np.concatenate([datagen.random_records(24, seed=33), np.asarray([encode_ref.PAD_RECORD], np.int32)])
    net = AlexNet(params, device=DEV)
    got = kernels.from_c8(kernels.encode_conv1(_dev(rec), net.conv1_hwio, net.conv1_base)).cpu().numpy()
    x = encode_ref.encode_records(rec)
    a = alexnet_ref._conv_layer(x, params["conv1/weights"], params["conv1/biases"], 4, "VALID", 1)
    want = alexnet_ref._lrn(np.ascontiguousarray(alexnet_ref._max_pool_3x3s2_valid(a))).transpose(0, 3, 1, 2)
    assert got.shape == want.shape == (25, 96, 27, 27)
    assert np.allclose(got, want, rtol=1e-4, atol=1e-4)
    _l, cls, prob = net.predict_records(_dev(rec))
    _ol, o_cls, o_prob = alexnet_ref.predict(params, x)
    assert np.abs(prob.cpu().numpy() - o_prob).max() < 1e-3


@pytest.mark.parametrize("n,cin,cout,hw,k,groups", [(3, 96, 256, 27, 5, 2), (5, 256, 384, 13, 3, 1), (2, 384, 384, 13, 3, 2),
                                                     (64, 384, 256, 13, 3, 2), (1, 16, 64, 5, 3, 1), (2, 48, 128, 6, 5, 1)])
def test_conv2d_same_matches_torch_fp32(n, cin, cout, hw, k, groups):
    """fp32 MFMA implicit GEMM vs a plain PyTorch fp32 conv of the same op (tolerance 2e-4 of the output scale)."""
    import torch.nn.functional as F
    rng = np.random.default_rng(cin + cout)
    x = rng.standard_normal((n, cin, hw, hw)).astype(np.float32)
    w = (rng.standard_normal((k, k, cin // groups, cout)) / np.sqrt(k * k * cin // groups)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    xt, wt, bt = _dev(x), _dev(w), _dev(b)
    want = F.conv2d(xt.double(), wt.permute(3, 2, 0, 1).contiguous().double(), bt.double(), 1, k // 2, 1, groups)
    xc, wp = kernels.to_c8(xt), kernels.pack_conv_weights(wt)
    got = kernels.from_c8(kernels.conv2d_same(xc, wp, bt, groups=groups, relu=False))
    scale = float(want.abs().max())
    assert float((got.double() - want).abs().max()) < 2e-4 * scale
    got_raw = kernels.from_c8(kernels.conv2d_same(xc, wp, None, groups=groups, relu=False))
    assert float((got_raw.double() + bt.double().view(1, -1, 1, 1) - want).abs().max()) < 2e-4 * scale
    got_relu = kernels.from_c8(kernels.conv2d_same(xc, wp, bt, groups=groups, relu=True))
    assert float((got_relu.double() - want.clamp_min(0)).abs().max()) < 2e-4 * scale


def test_conv2d_same_is_transpose_and_layout_sensitive():
    """Asymmetric one-hot probe: a single input element and a single weight element must land on exactly one output
    element (catches swapped MFMA operands, octet / half mix-ups in the C8 and packed layouts)."""
    n, cin, cout, hw, k, groups = 2, 32, 128, 7, 3, 2
    for (b, c, y, x0, ky, kx, o) in ((1, 5, 2, 3, 0, 2, 7), (0, 29, 6, 0, 2, 0, 100), (1, 16, 0, 6, 1, 1, 64)):
        x = torch.zeros(n, cin, hw, hw, device=DEV)
        w = torch.zeros(k, k, cin // groups, cout, device=DEV)
        x[b, c, y, x0] = 3.0
        g = c // (cin // groups)
        if o // (cout // groups) != g:
            o = g * (cout // groups) + o % (cout // groups)
        w[ky, kx, c % (cin // groups), o] = 2.0
        got = kernels.from_c8(kernels.conv2d_same(kernels.to_c8(x), kernels.pack_conv_weights(w), None, groups=groups))
        yy, xx = y - (ky - 1), x0 - (kx - 1)                 # output position whose tap (ky, kx) reads (y, x0)
        want = torch.zeros_like(got)
        if 0 <= yy < hw and 0 <= xx < hw:
            want[b, o, yy, xx] = 6.0
        assert torch.equal(got, want)


def test_fc8_softmax_and_packed_predict():
    from oracle import alexnet_ref
    from svision_amd.network.alexnet import AlexNet
    rng = np.random.default_rng(2)
    x = rng.standard_normal((7, 4096)).astype(np.float32)
    w = (rng.standard_normal((5, 4096)) * 0.03).astype(np.float32)
    b = rng.standard_normal(5).astype(np.float32)
    got = kernels.fc8_softmax(_dev(x), _dev(w), _dev(b)).cpu().numpy()
    logits = x.astype(np.float64) @ w.T.astype(np.float64) + b
    assert np.allclose(got[:, 6:11], logits, rtol=1e-5, atol=1e-5)
    assert np.array_equal(got[:, 5].astype(int), logits.argmax(1))
    assert np.allclose(got[:, :5], alexnet_ref.softmax(logits.astype(np.float32)), atol=1e-6)
    params = alexnet_ref.random_params(seed=4)
    net = AlexNet(params, device=DEV)
    rec = _dev(datagen.random_records(9, seed=5, hostile=False))
    packed = net.predict_records_packed(rec).cpu().numpy()
    _l, cls, prob = net.predict_records(rec)
    assert np.abs(packed[:, :5] - prob.cpu().numpy()).max() < 1e-5


def _dilate(m, r):
    out = np.zeros_like(m)
    h, w = m.shape[-2:]
    pad = np.pad(m, [(0, 0)] * (m.ndim - 2) + [(r, r), (r, r)])
    for dy in range(2 * r + 1):
        for dx in range(2 * r + 1):
            out |= pad[..., dy:dy + h, dx:dx + w]
    return out


def test_active_sets_match_a_numpy_restatement(oracle_lib):
    """Touched pixels of the first layer (vs the oracle's rasterised images) and the four pixel lists derived from them
    (vs dilation / pooling of boolean arrays)."""
    from oracle import cbind
    from bench import random_weights
    from svision_amd.network.alexnet import AlexNet
    n = 70
    rec_np = datagen.random_records(n, seed=21, hostile=True)
    net = AlexNet(random_weights(0), device=DEV)
    _y, touched = kernels.encode_conv1(_dev(rec_np), net.conv1_hwio, net.conv1_base, touched=True)
    img = cbind.rasterize(rec_np, "NHWC") + np.array([104, 117, 124], np.float32)
    on = (img > 0).any(3)
    want = np.zeros((n, 27, 27), bool)
    for y in range(27):
        for x in range(27):
            want[:, y, x] = on[:, 8 * y:8 * y + 19, 8 * x:8 * x + 19].any((1, 2))
    rows = touched.cpu().numpy().astype(np.int64) & ((1 << 27) - 1)
    got = ((rows[:, :, None] >> np.arange(27)[None, None, :]) & 1).astype(bool)
    assert np.array_equal(got, want)
    l2, l3, l4, l5, counts = kernels.alexnet_active_sets(touched)
    counts = counts.cpu().numpy()
    a2 = _dilate(want, 2)
    p2 = np.zeros((n, 13, 13), bool)
    for y in range(13):
        for x in range(13):
            p2[:, y, x] = a2[:, 2 * y:2 * y + 3, 2 * x:2 * x + 3].any((1, 2))
    a3 = _dilate(p2, 1)
    a4 = _dilate(a3, 1)
    a5 = _dilate(a4, 1)
    for lst, cnt, mask in ((l2, counts[0], a2), (l3, counts[1], a3), (l4, counts[2], a4), (l5, counts[3], a5)):
        assert np.array_equal(lst.cpu().numpy()[:cnt], np.flatnonzero(mask.reshape(-1)))          # active first, ascending
        assert np.array_equal(lst.cpu().numpy()[cnt:], np.flatnonzero(~mask.reshape(-1)))         # then everything else
    assert 0 < counts[0] < n * 729


def test_active_path_is_bit_identical_to_the_dense_path():
    """conv2..conv5 restricted to the active pixels + background elsewhere == the dense computation, bit for bit."""
    from bench import random_weights
    from svision_amd.network.alexnet import AlexNet
    params = random_weights(3)
    dense, sparse = AlexNet(params, device=DEV, active=False), AlexNet(params, device=DEV, active=True)
    for seed, hostile, n in ((1, False, 64), (2, True, 37), (3, False, 1), (4, True, 130)):
        rec = _dev(datagen.random_records(n, seed=seed, hostile=hostile))
        a = dense.predict_records_packed(rec)
        b = sparse.predict_records_packed(rec)
        assert torch.equal(a, b)
    pad = torch.tensor([[0, 1, 0, 1, 1, 1, 1, 1, 1, 1, 2, 2]] * 5, dtype=torch.int32, device=DEV)
    assert torch.equal(dense.predict_records_packed(pad), sparse.predict_records_packed(pad))


@pytest.mark.parametrize("fill", [0.0, 0.3, 0.99, 1.0])
def test_conv_with_pixel_list_and_background(fill):
    """List mode of svx_conv2d_same at the kernel level: active pixels = the dense result, all others = the given
    background, for empty / partial / nearly full (everything is computed above 97 %) / full touched masks."""
    rng = np.random.default_rng(int(fill * 100))
    n = 9
    touched = np.zeros((n, 27), np.int64)
    if fill >= 0.99:
        touched[:] = (1 << 27) - 1
        if fill < 1.0:
            touched[0, :3] = 0                          # a corner of one image stays inactive even after the 5x5 dilation
    elif fill > 0:
        for i in range(n):
            for _ in range(int(fill * 20)):
                y, x = rng.integers(0, 27, 2)
                touched[i, y] |= 1 << int(x)
    lists = kernels.alexnet_active_sets(_dev(touched.astype(np.int32)))
    counts = lists[4].cpu().numpy()
    for (cin, cout, hw, k, groups, li) in ((96, 256, 27, 5, 2, 0), (256, 384, 13, 3, 1, 1)):
        x = kernels.to_c8(torch.randn(n, cin, hw, hw, device=DEV))
        w = kernels.pack_conv_weights(torch.randn(k, k, cin // groups, cout, device=DEV) * 0.05)
        b = torch.randn(cout, device=DEV)
        bg = torch.randn(cout, hw, hw, device=DEV)
        bg8 = kernels.to_c8(bg.unsqueeze(0))[0]
        dense = kernels.from_c8(kernels.conv2d_same(x, w, b, groups=groups, relu=True))
        got = kernels.from_c8(kernels.conv2d_same(x, w, b, groups=groups, relu=True, pixels=lists[li], pixel_count=lists[4][li:li + 1],
                                                  background=bg8))
        active = torch.zeros(n * hw * hw, dtype=torch.bool, device=DEV)
        active[lists[li][:int(counts[li])].long()] = True
        active = active.view(n, 1, hw, hw)
        if int(counts[li]) * 100 >= n * hw * hw * 97:
            active[:] = True                            # nearly full: the kernel computes every pixel
        want = torch.where(active, dense, bg.unsqueeze(0).expand(n, -1, -1, -1))
        assert torch.equal(got, want)
        if fill == 0.0:
            assert int(counts[li]) == 0


def _callback_distance(a, b):
    """reference cluster_signatures.py:132-141 on double rows, as scipy's pdist hands them to the callback"""
    span1, span2 = a[1] - a[0], b[1] - b[0]
    c1, c2 = (a[0] + a[1]) // 2, (b[0] + b[1]) // 2
    with np.errstate(invalid="ignore", divide="ignore"):
        return min(abs(a[0] - b[0]), abs(a[1] - b[1]), abs(c1 - c2)) / a[2] + abs(span1 - span2) / max(span1, span2)


def test_span_position_distance_matches_the_reference_callback():
    """svx_span_position_distance (fp64) == the Python callback of the reference, bit for bit: golden partitions (the
    signatures of the golden sample, partitioned by the product), hostile rows (zero / negative / equal spans, odd sums,
    huge coordinates), several partitions per launch, and a 10^4-signature partition (5 x 10^7 pairs)."""
    from scipy.spatial.distance import pdist
    from svision_amd.collection.cluster_signatures import signature_partition, span_position_distance_condensed
    from svision_amd.collection.collect_signatures import analyze_alignments
    from tests import helpers
    rng = np.random.default_rng(5)
    parts = []
    sample = helpers.golden_sample(50, device=DEV)
    opts = helpers.default_options(min_support=3)
    for chrom, length in zip(sample.table.references, sample.table.lengths):
        sigs = analyze_alignments(sample.table.fetch(sample.table.get_tid(chrom), 0, length), sample, opts)
        parts += [np.array([[s.tstart, s.tend] for s in p], np.float64) for p in signature_partition(sigs, opts) if len(p) > 1]
    assert len(parts) > 10
    parts.append(np.array([[5, 5], [7, 7], [5, 5], [0, 11], [11, 0], [3, 10], [2, 9], [1e15, 1e15 + 3], [-7, 8], [-8, 7]], np.float64))
    parts.append(np.sort(rng.integers(0, 250_000_000, (300, 2)), axis=1).astype(np.float64))
    parts.append(np.array([[1, 2], [1, 2]], np.float64))
    big = np.cumsum(rng.integers(0, 40, 10_000))[:, None] + np.array([[0, 0]]) + np.stack([np.zeros(10_000), rng.integers(0, 3000, 10_000)], 1)
    parts.append(big.astype(np.float64))
    off = np.zeros(len(parts) + 1, np.int64)
    off[1:] = np.cumsum([len(p) for p in parts])
    flat = np.concatenate(parts)
    out, out_off = kernels.span_position_distance(_dev(flat[:, 0].copy()), _dev(flat[:, 1].copy()), off)
    out = out.cpu().numpy()
    for i, p in enumerate(parts):
        got = out[int(out_off[i]):int(out_off[i + 1])]
        assert got.size == len(p) * (len(p) - 1) // 2
        want_np = span_position_distance_condensed(p[:, 0], p[:, 1])
        assert np.array_equal(got, want_np, equal_nan=True)
        if len(p) <= 300:                                      # the callback itself, through scipy's pdist
            data = np.concatenate([p, np.full((len(p), 1), 1000.0)], axis=1)
            assert np.array_equal(got, pdist(data, metric=_callback_distance), equal_nan=True)
        else:                                                  # 5 x 10^7 pairs: sampled against the callback
            n = len(p)
            for k in rng.integers(0, got.size, 20_000):
                i0 = int(n - 2 - np.floor(np.sqrt(-8.0 * k + 4.0 * n * (n - 1) - 7) / 2.0 - 0.5))
                j0 = int(k + i0 + 1 - n * (n - 1) // 2 + (n - i0) * ((n - i0) - 1) // 2)
                want = _callback_distance(np.array([p[i0, 0], p[i0, 1], 1000.0]), np.array([p[j0, 0], p[j0, 1], 1000.0]))
                assert got[k] == want or (np.isnan(got[k]) and np.isnan(want))
    assert np.isnan(out[int(out_off[len(parts) - 4])])         # hostile partition: (5,5) vs (7,7): 0/0


@pytest.mark.parametrize("m,n,k", [(64, 4096, 9216), (64, 4096, 4096), (1, 64, 192), (37, 96, 1000), (130, 128, 2048), (256, 4096, 4096)])
def test_fc_bias_act_matches_fp64_reference(m, n, k):
    """fp32 MFMA split-K fc vs an fp64 matmul (tolerance 2e-5 of the output scale: f32 round-off over k <= 9216), ReLU and
    linear, ragged batch sizes (rows past m must not leak), and bit-reproducibility of the ordered reduction."""
    rng = np.random.default_rng(m + n + k)
    x = _dev(rng.standard_normal((m, k)).astype(np.float32))
    w = _dev((rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32))
    b = _dev(rng.standard_normal(n).astype(np.float32))
    want = x.double() @ w.double().t() + b.double()
    wp = kernels.pack_fc_weights(w)
    got = kernels.fc_bias_act(x, wp, b, relu=False)
    scale = float(want.abs().max())
    assert float((got.double() - want).abs().max()) < 2e-5 * scale
    got_relu = kernels.fc_bias_act(x, wp, b, relu=True)
    assert torch.equal(got_relu, got.clamp_min(0))
    assert torch.equal(kernels.fc_bias_act(x, wp, b, relu=False), got)
    probe = torch.zeros(m, k, device=DEV)                       # one-hot probe: transposition / packing mix-ups
    probe[m - 1, k - 3] = 2.0
    hot = kernels.fc_bias_act(probe, wp, torch.zeros(n, device=DEV), relu=False)
    ref = torch.zeros(m, n, device=DEV)
    ref[m - 1] = 2.0 * w[:, k - 3]
    assert torch.equal(hot, ref)


@pytest.mark.parametrize("c,hw,lrn", [(256, 27, True), (256, 13, False), (96, 27, True)])
def test_pool_reads_inactive_pixels_from_the_background(c, hw, lrn):
    """svx_bias_relu_pool_lrn with row masks: a pixel whose bit is clear is taken from the background tensor and never read
    from the input (NaN there must not leak) -- bit-identical to the plain kernel on the merged tensor."""
    torch.manual_seed(c + hw)
    n = 5
    x = torch.randn(n, c // 8, hw, hw, 8, device=DEV)
    bg = torch.randn(c // 8, hw, hw, 8, device=DEV)
    bias = torch.randn(c, device=DEV)
    active = torch.rand(n, hw, hw, device=DEV) < 0.4
    active[0] = True                                          # one image fully active, one fully inactive
    active[1] = False
    rows = (active.to(torch.int64) << torch.arange(hw, device=DEV)).sum(2).to(torch.int32).contiguous()
    merged = torch.where(active[:, None, :, :, None], x, bg[None])
    want = kernels.bias_relu_pool_lrn(merged, bias, lrn=lrn)
    holes = torch.where(active[:, None, :, :, None], x, torch.full_like(x, float("nan")))
    got = kernels.bias_relu_pool_lrn(holes, bias, lrn=lrn, active_rows=rows, background=bg)
    assert torch.equal(got, want) and not torch.isnan(got).any()
    with pytest.raises(Exception):
        kernels.bias_relu_pool_lrn(holes, bias, lrn=lrn, active_rows=rows)


def test_scan_lookback_makes_progress_on_a_busy_chip(oracle_lib):
    """The offsets pass's decoupled look-back assumes that the workgroups in front of a tile are dispatched (include/svx.h: it
    fails loudly -- SVX_SCAN_FAILED -- rather than hang or answer wrongly).  In the pipeline the scan's launches co-run with the
    inflate's tokens kernel, which owns every CU while it runs, and with the convolutions: 1,000 scans of a
    several-tile batch launched while a side stream keeps the chip full of tokens + LZ launches -- never a failure, always the
    same bytes."""
    import struct
    import zlib
    from oracle import cbind
    rng = np.random.default_rng(17)
    blocks = []
    for _ in range(48):                                       # 48 distinct blocks, repeated: ~3,000 blocks (190 MB inflated) per launch
        payload = bytes(rng.integers(0, 7, 65280, dtype=np.uint8) + 65)
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        cdata = co.compress(payload) + co.flush()
        blocks.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cdata) + 25) + cdata
                      + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))
    raw = np.frombuffer(b"".join(blocks * 64), np.uint8)
    src_off, src_len, isize, _blk = kernels.bgzf_block_table(raw)
    padded = np.zeros((raw.size + 31) // 16 * 16, np.uint8)
    padded[:raw.size] = raw
    d_comp = torch.from_numpy(padded).to(DEV)
    cigar, off, ref_start = datagen.random_cigars(40_000, seed=23, mean_ops=160, long_gap_rate=0.002)     # 40 tiles of 1,024 alignments
    d_c, d_o, d_r = _dev(cigar.view(np.int32)), _dev(off.astype(np.int64)), _dev(ref_start)
    o_gaps, o_off, o_stats = cbind.cigar_scan(cigar, off, ref_start, 50)
    side = torch.cuda.Stream(device=DEV)
    first = None
    for rep in range(1000):
        if rep % 10 == 0:                                     # keep inflate work (tokens + LZ of ~3,000 blocks per launch) queued beside the scans
            with torch.cuda.stream(side):
                _out, status = kernels.bgzf_inflate(d_comp, src_off, src_len, isize, wave="fast", crc=False)
        res = kernels.cigar_scan(d_c, d_o, d_r, 50, gaps_cap=1 << 16)
        total = res.total()                                   # raises SvxError on SVX_SCAN_FAILED
        if first is None:
            gaps, gap_off, stats = res.to_host()
            assert gaps.tobytes() == o_gaps.tobytes() and np.array_equal(gap_off, o_off) and np.array_equal(stats, o_stats) and total > 0
            first = (total, res.gaps[:total * 6].clone(), res.gap_off.clone())
        else:
            assert total == first[0]
            if rep % 50 == 0:
                assert torch.equal(res.gaps[:total * 6], first[1]) and torch.equal(res.gap_off, first[2])
    side.synchronize()
    assert not bool(status.any())
