"""-m boundary: TF1 V2 checkpoint (tensor bundle) reader, validated by round trip through the
own writer (real SVision weights are not available offline)."""
import numpy as np
import pytest

from svision_amd.network import tf_checkpoint as ck
from svision_amd.network.alexnet import AlexNet, checkpoint_shapes


def test_roundtrip_with_extra_variables(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {k: rng.standard_normal(s).astype(np.float32) for k, s in
               {"conv1/weights": (11, 11, 3, 96), "conv1/biases": (96,), "fc8/weights": (4096, 5), "fc8/biases": (5,),
                "fc8/weights/Adam": (4096, 5), "beta1_power": ()}.items()}
    prefix = str(tmp_path / "model.ckpt")
    ck.write_checkpoint(prefix, tensors, crc_tensors=True)
    idx = ck.read_index(prefix)
    assert idx[""]["num_shards"] == 1
    assert idx["conv1/weights"]["shape"] == (11, 11, 3, 96) and idx["beta1_power"]["shape"] == ()
    got = ck.read_checkpoint(prefix)
    assert set(got) == set(tensors)
    for k in tensors:
        assert np.array_equal(got[k], tensors[k])
    only = ck.read_checkpoint(prefix, names=["fc8/weights", "fc8/biases"])
    assert set(only) == {"fc8/weights", "fc8/biases"}
    with pytest.raises(KeyError):
        ck.read_checkpoint(prefix, names=["fc7/weights"])


def test_missing_prefix_and_bad_magic(tmp_path):
    with pytest.raises(FileNotFoundError):
        ck.read_checkpoint(str(tmp_path / "nope.ckpt"))
    p = tmp_path / "bad.ckpt.index"
    p.write_bytes(b"\x00" * 64)
    with pytest.raises(ValueError):
        ck.read_index(str(tmp_path / "bad.ckpt"))


def test_crc32c_known_answer():
    # RFC 3720 test vector: 32 bytes of zeros -> 0x8A9136AA ; "123456789" -> 0xE3069283
    assert ck._crc32c(b"\x00" * 32) == 0x8A9136AA
    assert ck._crc32c(b"123456789") == 0xE3069283


def test_alexnet_requires_all_tensors():
    shapes = checkpoint_shapes()
    params = {k: np.zeros(s, np.float32) for k, s in shapes.items()}
    del params["fc7/biases"]
    with pytest.raises(KeyError):
        AlexNet(params, device="cpu")
    params["fc7/biases"] = np.zeros(4096, np.float32)
    params["conv2/weights"] = np.zeros((5, 5, 96, 256), np.float32)      # ungrouped shape is wrong
    with pytest.raises(ValueError):
        AlexNet(params, device="cpu")


def test_checksums_are_verified_on_read(tmp_path):
    """A flipped byte in a tensor or in the index, or a truncated data shard, is an error -- not wrong weights."""
    from svision_amd import _lib
    lib = _lib.load()
    buf = np.frombuffer(b"123456789" * 1000 + b"xyz", np.uint8)
    assert lib.svx_crc32c(buf.ctypes.data, buf.size) == ck._crc32c(buf.tobytes())          # native slicing-by-8 == table loop
    assert lib.svx_crc32c(np.zeros(32, np.uint8).ctypes.data, 32) == 0x8A9136AA
    rng = np.random.default_rng(1)
    tensors = {"conv1/weights": rng.standard_normal((11, 11, 3, 96)).astype(np.float32), "conv1/biases": rng.standard_normal(96).astype(np.float32)}
    prefix = str(tmp_path / "m.ckpt")
    ck.write_checkpoint(prefix, tensors, crc_tensors=True)
    assert np.array_equal(ck.read_checkpoint(prefix)["conv1/weights"], tensors["conv1/weights"])
    shard = prefix + ".data-00000-of-00001"
    raw = bytearray(open(shard, "rb").read())
    raw[1000] ^= 0x40
    open(shard, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum mismatch in tensor"):
        ck.read_checkpoint(prefix)
    raw[1000] ^= 0x40
    open(shard, "wb").write(bytes(raw[:-100]))
    with pytest.raises(ValueError, match="truncated"):
        ck.read_checkpoint(prefix)
    open(shard, "wb").write(bytes(raw))
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[10] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError, match="block checksum"):
        ck.read_checkpoint(prefix)
