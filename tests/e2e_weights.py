"""The fixed AlexNet weights of the end-to-end fixture (tests/golden/make_e2e_fixture.py): seeded random weights
(oracle.alexnet_ref.random_params) with fc8 rescaled per class by the calibration the fixture stores.  228 MB: rebuilt
from the seed, never committed; ``params_crc`` guards against a NumPy whose generator stream differs."""
import gzip
import json
import os
import zlib

import numpy as np

SEED = 11
FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "e2e_small.expected.json.gz")


def apply_calibration(params, scale, bias):
    params = dict(params)
    params["fc8/weights"] = (params["fc8/weights"] * np.asarray(scale, np.float32)[None, :]).astype(np.float32)
    params["fc8/biases"] = np.asarray(bias, np.float32).copy()
    return params


def params_crc(params):
    crc = 0
    for key in sorted(params):
        crc = zlib.crc32(np.ascontiguousarray(params[key]).tobytes(), crc)
    return crc


def load_fixture():
    with gzip.open(FIXTURE, "rb") as f:
        return json.load(f)


def fixture_params(expected=None):
    """-> the checkpoint-named parameter dict the fixture's CNN outputs were computed with."""
    from oracle import alexnet_ref
    expected = expected or load_fixture()
    params = apply_calibration(alexnet_ref.random_params(seed=expected["seed"]),
                               np.asarray(expected["fc8_scale"], np.uint32).view(np.float32),
                               np.asarray(expected["fc8_bias"], np.uint32).view(np.float32))
    if params_crc(params) != expected["weights_crc"]:
        raise RuntimeError("the seeded weights differ from the ones the fixture was generated with (NumPy generator stream changed?)")
    return params
