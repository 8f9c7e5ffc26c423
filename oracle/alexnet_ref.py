"""ORACLE (test infrastructure, not product): AlexNet forward, NumPy fp32.

CPU restatement of the TF1 graph built by
/root/reference/src/network/alexnet.py:26-58 (layers :100-170) and evaluated at
/root/reference/src/network/predict.py:209 (fc8 logits, argmax, softmax).
TensorFlow 1.14 is not installed and its kernels are not vendored by the
reference, so the op semantics below (SAME/VALID padding, LRN window, NHWC
flatten order) follow TensorFlow's documented behaviour: **parity unpinned**
against TensorFlow itself; the product is checked against this restatement to
1e-3 on softmax (BASELINE.json north_star).

Parameters use the checkpoint's own names and layouts: ``convN/weights`` HWIO
``[kh, kw, Cin/groups, Cout]``, ``fcN/weights`` ``[in, out]``, ``*/biases``.
"""
import numpy as np

LAYERS = [  # name, kh, kw, cin_per_group, cout, stride, padding, groups
    ('conv1', 11, 11, 3, 96, 4, 'VALID', 1),
    ('conv2', 5, 5, 48, 256, 1, 'SAME', 2),
    ('conv3', 3, 3, 256, 384, 1, 'SAME', 1),
    ('conv4', 3, 3, 192, 384, 1, 'SAME', 2),
    ('conv5', 3, 3, 192, 256, 1, 'SAME', 2),
]
FCS = [('fc6', 9216, 4096), ('fc7', 4096, 4096), ('fc8', 4096, 5)]


def param_shapes():
    shapes = {}
    for name, kh, kw, cin, cout, _s, _p, _g in LAYERS:
        shapes[name + '/weights'] = (kh, kw, cin, cout)
        shapes[name + '/biases'] = (cout,)
    for name, nin, nout in FCS:
        shapes[name + '/weights'] = (nin, nout)
        shapes[name + '/biases'] = (nout,)
    return shapes


def random_params(seed=0, scale=1.0, input_scale=0.02):
    """He-style random weights; conv1 is scaled down by ``input_scale`` because the
    inputs are O(100) (0/255 minus mean), so that the 5 logits come out O(1-10) and
    the softmax is not saturated (a saturated softmax would make the 1e-3 check vacuous)."""
    rng = np.random.default_rng(seed)
    params = {}
    for key, shp in param_shapes().items():
        if key.endswith('biases'):
            params[key] = (rng.standard_normal(shp) * 0.1).astype(np.float32)
        else:
            fan_in = int(np.prod(shp[:-1]))
            params[key] = (rng.standard_normal(shp) * scale * np.sqrt(2.0 / fan_in)).astype(np.float32)
    params['conv1/weights'] *= np.float32(input_scale)
    return params


def _conv2d_nhwc(x, w, stride, padding):
    """tf.nn.conv2d, NHWC x HWIO.  SAME: out=ceil(in/stride), pad_before=pad_total//2."""
    n, h, wd, c = x.shape
    kh, kw, cin, cout = w.shape
    assert cin == c
    if padding == 'SAME':
        oh, ow = -(-h // stride), -(-wd // stride)
        ph = max((oh - 1) * stride + kh - h, 0)
        pw = max((ow - 1) * stride + kw - wd, 0)
        x = np.pad(x, ((0, 0), (ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)))
    else:
        oh, ow = (h - kh) // stride + 1, (wd - kw) // stride + 1
    s0, s1, s2, s3 = x.strides
    patches = np.lib.stride_tricks.as_strided(
        x, shape=(n, oh, ow, kh, kw, c),
        strides=(s0, s1 * stride, s2 * stride, s1, s2, s3), writeable=False)
    cols = patches.reshape(n * oh * ow, kh * kw * c)
    out = cols @ w.reshape(kh * kw * c, cout)
    return out.reshape(n, oh, ow, cout)


def _conv_layer(x, w, b, stride, padding, groups):
    """alexnet.py:100-135: (grouped) conv + bias + relu."""
    if groups == 1:
        y = _conv2d_nhwc(x, w, stride, padding)
    else:
        xs = np.split(x, groups, axis=3)
        ws = np.split(w, groups, axis=3)
        y = np.concatenate([_conv2d_nhwc(xi, wi, stride, padding) for xi, wi in zip(xs, ws)], axis=3)
    return np.maximum(y + b, 0).astype(np.float32)


def _max_pool_3x3s2_valid(x):
    n, h, w, c = x.shape
    oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    s0, s1, s2, s3 = x.strides
    win = np.lib.stride_tricks.as_strided(
        x, shape=(n, oh, ow, 3, 3, c), strides=(s0, s1 * 2, s2 * 2, s1, s2, s3), writeable=False)
    return win.max(axis=(3, 4))


def _lrn(x, radius=2, alpha=2e-05, beta=0.75, bias=1.0):
    """tf.nn.local_response_normalization: alpha is NOT divided by the window size."""
    sq = np.square(x)
    c = x.shape[-1]
    acc = np.zeros_like(x)
    for d in range(-radius, radius + 1):
        lo, hi = max(0, d), min(c, c + d)
        acc[..., lo - d:hi - d] += sq[..., lo:hi]
    return (x / np.power(bias + alpha * acc, beta)).astype(np.float32)


def forward(params, x):
    """x: float32 [B,227,227,3] NHWC (already mean-subtracted) -> logits [B,5]."""
    x = np.ascontiguousarray(x, np.float32)
    acts = x
    for name, _kh, _kw, _cin, _cout, stride, padding, groups in LAYERS:
        acts = _conv_layer(np.ascontiguousarray(acts), params[name + '/weights'],
                           params[name + '/biases'], stride, padding, groups)
        if name in ('conv1', 'conv2'):
            acts = _lrn(np.ascontiguousarray(_max_pool_3x3s2_valid(acts)))
        elif name == 'conv5':
            acts = np.ascontiguousarray(_max_pool_3x3s2_valid(acts))
    flat = acts.reshape(acts.shape[0], 6 * 6 * 256)          # NHWC flatten: (h*6+w)*256+c
    h6 = np.maximum(flat @ params['fc6/weights'] + params['fc6/biases'], 0)
    h7 = np.maximum(h6 @ params['fc7/weights'] + params['fc7/biases'], 0)
    return (h7 @ params['fc8/weights'] + params['fc8/biases']).astype(np.float32)


def softmax(logits):
    z = logits - logits.max(axis=1, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)


def predict(params, x):
    """(logits, argmax, softmax) as fetched at predict.py:209."""
    logits = forward(params, x)
    return logits, logits.argmax(axis=1), softmax(logits)
