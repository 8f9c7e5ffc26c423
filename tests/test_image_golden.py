"""The similarity-image leg pinned by the reference itself.

tests/golden/image_small.expected.json.gz holds what the REFERENCE's BatchGenerator.next_batch
(create_batch.py:88-155 -> classes.py:42-54 -> plot_segment.py:8-73) returned for 1347 TSV lines
(678 from the golden collection fixtures, 669 hostile) + the pad rows of the last batch
(tests/golden/make_image_fixture.py).  CPU: both oracles (Python and C) reproduce it bit for bit.
GPU: svx_rasterize (both layouts), the product's BatchGenerator and the touched-pixel masks of
svx_encode_conv1 reproduce it through the C ABI.
"""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import encode_ref
from svision_amd.network import create_batch
from tests.helpers import GOLDEN

IMG = 227


@pytest.fixture(scope="module")
def fixture():
    with gzip.open(os.path.join(GOLDEN, "image_small.expected.json.gz"), "rb") as f:
        doc = json.load(f)
    doc["records"] = np.asarray([create_batch.parse_data_fields(d.split("_")) for d in doc["data"]], np.int32)
    return doc


def dense(doc, idx=None, layout="NHWC"):
    """float32 images the reference returned (mask - mean), rebuilt from the sparse pixel lists."""
    idx = range(len(doc["pixels"])) if idx is None else idx
    mean = np.asarray(doc["mean"], np.float32)
    out = np.empty((len(idx), 3, IMG * IMG), np.float32)
    for j, i in enumerate(idx):
        for c in range(3):
            out[j, c] = -mean[c]
            out[j, c, doc["pixels"][i][c]] = np.float32(255.0) - mean[c]
    out = out.reshape(len(idx), 3, IMG, IMG)
    return out if layout == "NCHW" else np.ascontiguousarray(out.transpose(0, 2, 3, 1))


def touched_from_pixels(pix):
    """27 row masks of the pooled conv1 pixels with a set image pixel in their receptive field
    (pool 3x3/2 over conv 11x11/4 VALID: pooled (py,px) sees image rows 8py..8py+18, cols 8px..8px+18)."""
    on = np.zeros(IMG * IMG, bool)
    for c in range(3):
        on[pix[c]] = True
    on = on.reshape(IMG, IMG)
    words = np.zeros(27, np.uint32)
    for py in range(27):
        for px in range(27):
            if on[8 * py:8 * py + 19, 8 * px:8 * px + 19].any():
                words[py] |= np.uint32(1 << px)
    return words


def test_fixture_shape(fixture):
    doc = fixture
    assert doc["n_lines"] == len(doc["lines"]) == 1347 and doc["n_golden"] == 678
    assert len(doc["pixels"]) == len(doc["data"]) == 1408 == doc["batch_size"] * 22
    assert doc["data"][-1] == create_batch.PAD_DATA and tuple(doc["records"][-1]) == encode_ref.PAD_RECORD
    # parsers (oracle's and product's) agree with the reference's '_'-joined data strings
    for line, data in zip(doc["lines"], doc["data"]):
        rec, _ = encode_ref.parse_tsv_line(line)
        assert "_".join(line.split("\t")[1:13]) == data
        assert tuple(rec) == create_batch.parse_data_fields(data.split("_"))
    # the hostile part really is hostile
    n_clip = sum(1 for r in doc["records"][678:1347]
                 if any(not (0 <= v <= max(r[10], r[11])) for v in (r[0], r[2], r[3], r[5], r[7], r[8])))
    assert n_clip > 150
    assert sum(1 for r in doc["records"] if max(r[10], r[11]) < 227) > 100            # ratio clamps to 1
    assert sum(1 for r in doc["records"] if r[3] - r[2] <= 0 or r[8] - r[7] <= 0) > 30  # zero / negative lengths
    assert sum(1 for r in doc["records"] if r[4] == 0 or r[9] == 0) > 400             # reverse segments


def test_python_oracle_equals_reference_images(fixture):
    want = dense(fixture)
    got = encode_ref.encode_records(fixture["records"])
    assert got.dtype == want.dtype == np.float32
    bad = [i for i in range(len(want)) if not np.array_equal(got[i], want[i])]
    assert not bad, "images differ from the reference's: %s" % bad[:10]


@pytest.mark.parametrize("layout", ["NHWC", "NCHW"])
def test_c_oracle_equals_reference_images(fixture, oracle_lib, layout):
    from oracle import cbind
    got = cbind.rasterize(fixture["records"], layout)
    assert np.array_equal(got, dense(fixture, layout=layout))


def test_product_batch_generator_text_side(fixture):
    """Product BatchGenerator (host side only: no device call) keeps the reference's data strings and padding."""
    gen = create_batch.BatchGenerator([ln for ln in fixture["lines"]], shuffle=False, nb_classes=5, batch_size=fixture["batch_size"])
    assert gen.images == fixture["data"]
    assert np.array_equal(gen.records, fixture["records"])
    assert gen.data_size == 1408 and gen.labels[-1] == create_batch.PAD_LABEL


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["NHWC", "NCHW"])
def test_hip_rasteriser_equals_reference_images(fixture, layout):
    import torch
    from svision_amd import kernels
    rec = torch.from_numpy(fixture["records"]).to("cuda:0")
    got = kernels.rasterize(rec, layout=layout).cpu().numpy()
    want = dense(fixture, layout=layout)
    bad = [i for i in range(len(want)) if not np.array_equal(got[i], want[i])]
    assert not bad, "HIP images differ from the reference's: %s" % bad[:10]


@pytest.mark.gpu
def test_product_next_batch_equals_reference_batches(fixture):
    gen = create_batch.BatchGenerator(list(fixture["lines"]), shuffle=False, nb_classes=5, batch_size=fixture["batch_size"],
                                      device="cuda:0", layout="NHWC")
    b = fixture["batch_size"]
    for k in range(gen.data_size // b):
        images, labels = gen.next_batch(b)
        assert tuple(images.shape) == (b, IMG, IMG, 3)
        assert np.array_equal(images.cpu().numpy(), dense(fixture, range(k * b, (k + 1) * b)))
    assert labels[-1] == create_batch.PAD_LABEL


@pytest.mark.gpu
def test_encode_conv1_touched_masks_equal_reference_pixels(fixture):
    """svx_encode_conv1 never materialises the image: its touched-pixel masks must be exactly the pooled
    pixels whose receptive field holds a pixel the reference drew."""
    import torch
    from svision_amd import kernels
    rec = torch.from_numpy(fixture["records"]).to("cuda:0")
    g = torch.Generator().manual_seed(3)
    w1 = (torch.randn((11, 11, 3, 96), generator=g) * 0.01).to("cuda:0")
    base = torch.randn(96, generator=g).to("cuda:0")
    _, mask = kernels.encode_conv1(rec, w1, base, touched=True)
    got = mask.cpu().numpy().view(np.uint32)
    want = np.stack([touched_from_pixels(p) for p in fixture["pixels"]])
    bad = np.flatnonzero((got != want).any(1))
    assert bad.size == 0, "touched masks differ for images %s" % bad[:10].tolist()
