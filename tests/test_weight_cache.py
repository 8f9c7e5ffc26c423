"""The device-layout weights of a checkpoint cached next to it (svision_amd/network/weight_cache.py): the blob round-trips
every buffer bit for bit, its name follows the bundle's bytes, and anything foreign or short is ignored.  (GPU: the network
adopted from the cache predicts what the network built from the checkpoint predicts -- tests/test_gpu_pipeline.py.)"""
import os

import numpy as np
import torch

from svision_amd.network import tf_checkpoint as ck, weight_cache
from svision_amd.network.alexnet import AlexNet


def _weights(seed=0):
    from svision_amd.network.alexnet import checkpoint_shapes
    rng = np.random.default_rng(seed)
    return {k: rng.standard_normal(shp).astype(np.float32) * np.float32(0.01) for k, shp in checkpoint_shapes().items()}


def test_blob_round_trip_and_digest(tmp_path, monkeypatch):
    prefix = str(tmp_path / "m.ckpt")
    params = _weights(1)
    ck.write_checkpoint(prefix, params)
    d1 = weight_cache.checkpoint_digest(prefix, abi=420)
    assert d1 and d1 == weight_cache.checkpoint_digest(prefix, abi=420) and d1 != weight_cache.checkpoint_digest(prefix, abi=421)
    net = AlexNet(params, device="cpu")
    path = weight_cache.cache_path(prefix, d1)
    assert os.path.dirname(path) == str(tmp_path) and d1[:16] in os.path.basename(path)
    assert weight_cache.save(path, net.packed_tensors())
    blob, names = weight_cache.load(path)
    again = AlexNet(None, device="cpu", packed=(blob, names))
    fresh = dict(net.named_buffers())
    got = dict(again.named_buffers())
    assert set(fresh) == set(got) and len(fresh) == 16
    for k in fresh:
        assert fresh[k].shape == got[k].shape and torch.equal(fresh[k], got[k]), k
    assert again._background is None                       # backgrounds are computed with the kernels: only a GPU process caches them
    # another checkpoint under the same name: another digest, so the old blob is never picked up
    params["fc8/biases"] = params["fc8/biases"] + np.float32(1)
    ck.write_checkpoint(prefix, params)
    d2 = weight_cache.checkpoint_digest(prefix, abi=420)
    assert d2 != d1 and weight_cache.load(weight_cache.cache_path(prefix, d2)) is None
    # foreign, short and other-format files are ignored
    with open(path, "r+b") as f:
        f.truncate(os.path.getsize(path) - 4096)
    assert weight_cache.load(path) is None
    with open(path, "wb") as f:
        f.write(b"not a blob")
    assert weight_cache.load(path) is None
    assert weight_cache.checkpoint_digest(str(tmp_path / "absent"), abi=420) is None
    monkeypatch.setenv("SVX_CACHE_DIR", str(tmp_path / "elsewhere"))
    assert os.path.dirname(weight_cache.cache_path(prefix, d1)) == str(tmp_path / "elsewhere")
    assert weight_cache.save(weight_cache.cache_path(prefix, d1), {"x": np.ones(3, np.float32)})      # the directory is created
    monkeypatch.setenv("SVX_WEIGHT_CACHE", "0")
    assert not weight_cache.enabled()
