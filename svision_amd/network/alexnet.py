"""AlexNet classifier of the similarity images on MI355X (fp32), one product path.

Mirrors the TF1 graph of the reference (src/network/alexnet.py:26-58, layer
helpers :100-170): conv1 11x11/4 VALID 3->96, pool, LRN; conv2 5x5 SAME g=2,
pool, LRN; conv3/4/5 3x3 SAME (g=1,2,2), pool; fc6/fc7 (ReLU), fc8 -> 5 logits
(DEL, INS, INV, DUP, tDUP).  Dropout is the identity at inference
(predict.py:22,210).  The input is the packed 12-int segment-pair record of a
candidate image (TSV columns 1..12), not the image: rasterisation and the first
layer are one kernel.

    svx_encode_conv1          rasterise + conv1 + relu + pool1 + norm1 (sparse: the image is a few thin lines)
    svx_alexnet_active_sets   which outputs of conv2..conv5 can differ from the response to an empty image
    svx_conv2d_same           conv2..conv5 on the fp32 matrix cores, active pixels only (list mode), bias+relu fused
    svx_bias_relu_pool_lrn    conv2 / conv5 epilogues
    svx_fc_bias_act           fc6 / fc7 (+ bias + relu) as a weight stream feeding fp32 MFMAs, split-K, ordered reduce
    svx_fc8_softmax           fc8 + softmax + argmax + packing

Parameters keep the checkpoint's names and layouts (``convN/weights`` HWIO,
``fcN/weights`` [in,out]) at the ``-m`` boundary and are re-laid out once for the device:
conv2..5 weights packed for svx_conv2d_same (kernels.pack_conv_weights), fc6 rows
permuted from the reference's NHWC flatten ((h*6+w)*256+c) to the C8 flatten of pool5
(((c/8)*36 + h*6+w)*8 + c%8), fc6 / fc7 weights packed for svx_fc_bias_act, fc8 stored [out,in].  Activations between the kernels
are in the C8 layout of include/svx.h.  There is no CPU or library fallback: the plain
PyTorch restatement used as a cross-check lives in oracle/alexnet_torch.py (tests only).
"""
import numpy as np
import torch

CLASSES = ("DEL", "INS", "INV", "DUP", "tDUP")

# name, kernel, cin/groups, cout, stride, pad, groups
_CONVS = (
    ("conv1", 11, 3, 96, 4, 0, 1),
    ("conv2", 5, 48, 256, 1, 2, 2),
    ("conv3", 3, 256, 384, 1, 1, 1),
    ("conv4", 3, 192, 384, 1, 1, 2),
    ("conv5", 3, 192, 256, 1, 1, 2),
)
_FCS = (("fc6", 9216, 4096), ("fc7", 4096, 4096), ("fc8", 4096, 5))


def checkpoint_shapes():
    """Tensor names/shapes the reference graph restores by name (predict.py:183-184)."""
    shapes = {}
    for name, k, cin, cout, _s, _p, _g in _CONVS:
        shapes[f"{name}/weights"] = (k, k, cin, cout)
        shapes[f"{name}/biases"] = (cout,)
    for name, nin, nout in _FCS:
        shapes[f"{name}/weights"] = (nin, nout)
        shapes[f"{name}/biases"] = (nout,)
    return shapes


def validate_params(params):
    """Missing tensor -> KeyError (TF raises NotFoundError), wrong shape -> ValueError."""
    want = checkpoint_shapes()
    missing = [k for k in want if k not in params]
    if missing:
        raise KeyError(f"checkpoint lacks tensors {missing}")
    for k, shp in want.items():
        if tuple(np.shape(params[k])) != shp:
            raise ValueError(f"{k} has shape {tuple(np.shape(params[k]))}, expected {shp}")


class AlexNet(torch.nn.Module):
    """Inference-only AlexNet holding device-layout parameters.  ``active=False`` computes conv2..conv5 at every pixel
    (same kernels in dense mode; the test suite requires both settings to agree bit for bit)."""

    def __init__(self, params, device="cuda", mean=(104.0, 117.0, 124.0), active=True, packed=None):
        super().__init__()
        from .. import kernels
        self.active = bool(active)
        self._background = None
        self.executed = None            # optional int64 device tensor [5]: running executed-pixel / image counts (bench)
        if packed is not None:          # device-layout tensors of an earlier process (weight_cache.load): one upload, views
            self._adopt(packed, device)
            return
        validate_params(params)

        def f32(a):
            return torch.from_numpy(np.array(a, np.float32, copy=True))
        for name, _k, _cin, _cout, _s, _p, groups in _CONVS[1:]:
            self.register_buffer(f"{name}_w", kernels.pack_conv_weights(f32(params[f"{name}/weights"])).to(device))
            self.register_buffer(f"{name}_b", f32(params[f"{name}/biases"]).to(device))
        # sparse first layer (svx_encode_conv1): checkpoint-layout weights + the constant response of
        # the all-background image, base[k] = bias[k] - sum mean[ch] * w[..., ch, k] (float64 on the host)
        w1 = np.asarray(params["conv1/weights"], np.float64)
        base = np.asarray(params["conv1/biases"], np.float64) - np.einsum("hwck,c->k", w1, np.asarray(mean, np.float64))
        self.register_buffer("conv1_hwio", f32(params["conv1/weights"]).to(device))
        self.register_buffer("conv1_base", torch.from_numpy(base.astype(np.float32)).to(device))
        for name, nin, nout in _FCS:
            # the checkpoint's [in, out] matrix goes to the device as it is and is permuted THERE (pure permutations: the same
            # values bit for bit) -- on the host the three transposes were 0.43 s of a command line's 0.86 s before its first window
            w = f32(params[f"{name}/weights"]).to(device)
            if name == "fc6":
                # rows (h,w,c) of the reference's NHWC flatten -> (c/8, h, w, c%8): pool5 is flattened in C8
                w = w.reshape(6, 6, 32, 8, nout).permute(2, 0, 1, 3, 4).reshape(nin, nout)
            wt = w.t()                                                                                    # [out,in]
            self.register_buffer(f"{name}_w", kernels.pack_fc_weights(wt) if name != "fc8" else wt.contiguous())
            self.register_buffer(f"{name}_b", f32(params[f"{name}/biases"]).to(device))

    _BG = ("conv2", "conv3", "conv4", "conv5")

    def packed_tensors(self, with_background=True):
        """{name: CPU float32 tensor} of everything a later process needs instead of the checkpoint: the buffers in their
        device layouts and (GPU only: they are computed with the kernels) the background activations."""
        out = {name: buf.detach().cpu() for name, buf in self.named_buffers()}
        if with_background and self.conv1_base.is_cuda:
            for k, v in self.background().items():
                out["background/" + k] = v.detach().cpu()
        return out

    def _adopt(self, packed, device):
        blob, names = packed
        want = {"conv1_hwio", "conv1_base"} | {f"{n}_{s}" for n in ("conv2", "conv3", "conv4", "conv5", "fc6", "fc7", "fc8") for s in "wb"}
        if not want <= set(names):
            raise ValueError("packed weights lack %s" % sorted(want - set(names)))
        import warnings
        with warnings.catch_warnings():                                  # (torch warns about tensors over read-only memory; this one is only copied from)
            warnings.simplefilter("ignore", UserWarning)
            host = torch.from_numpy(np.asarray(blob))
        dev_blob = host.to(device) if torch.device(device).type != "cpu" else host.clone()       # ONE host-to-device copy

        def view(name):
            shape, off = names[name]
            n = int(np.prod(shape)) if shape else 1
            return dev_blob[off:off + n].view(*shape)
        for name in sorted(want):
            self.register_buffer(name, view(name))
        if all("background/" + k in names for k in self._BG):
            self._background = {k: view("background/" + k) for k in self._BG}

    def _convs(self, records):
        """records int32 [B,12] -> pool5 activations, C8 [B,32,6,6,8]."""
        from .. import kernels
        if not self.active:
            x = kernels.encode_conv1(records, self.conv1_hwio, self.conv1_base)
            x = kernels.conv2d_same(x, self.conv2_w, None, groups=2)
            x = kernels.bias_relu_pool_lrn(x, self.conv2_b, lrn=True)
            x = kernels.conv2d_same(x, self.conv3_w, self.conv3_b, groups=1, relu=True)
            x = kernels.conv2d_same(x, self.conv4_w, self.conv4_b, groups=2, relu=True)
            x = kernels.conv2d_same(x, self.conv5_w, None, groups=2)
            return kernels.bias_relu_pool_lrn(x, self.conv5_b, lrn=False)
        bg = self.background()
        x, touched = kernels.encode_conv1(records, self.conv1_hwio, self.conv1_base, touched=True)
        l2, l3, l4, l5, counts, rows2 = kernels.alexnet_active_sets(touched, totals=self.executed, rows=True)

        def conv(name, x, pixels, k, bias, relu, groups):
            # active pixels computed, the others copied from the background by the workgroups behind the compute tiles
            return kernels.conv2d_same(x, getattr(self, name + "_w"), bias, groups=groups, relu=relu, pixels=pixels,
                                       pixel_count=counts[k:k + 1], background=bg[name])
        # conv2 writes its active pixels only (61 % of its output would be copies of the background, and that copy is
        # not hidden behind the matrix work): the pool reads the background for the others itself
        x = kernels.conv2d_same(x, self.conv2_w, None, groups=2, pixels=l2, pixel_count=counts[0:1],
                                out=torch.empty((records.shape[0], 32, 27, 27, 8), dtype=torch.float32, device=records.device))
        x = kernels.bias_relu_pool_lrn(x, self.conv2_b, lrn=True, active_rows=rows2, background=bg["conv2"])
        x = conv("conv3", x, l3, 1, self.conv3_b, True, 1)
        x = conv("conv4", x, l4, 2, self.conv4_b, True, 2)
        x = conv("conv5", x, l5, 3, None, False, 2)
        return kernels.bias_relu_pool_lrn(x, self.conv5_b, lrn=False)

    @torch.no_grad()
    def background(self):
        """Outputs of conv2..conv5 for an empty image (C8 [C/8,H,W,8] each), computed once with the same kernels: the
        first layer's constant vector (any pooled pixel without a set tap under it) pushed through the dense path."""
        if self._background is None:
            from .. import kernels
            from .create_batch import PAD_DATA, parse_data_fields
            dev = self.conv1_base.device
            rec = torch.tensor([parse_data_fields(PAD_DATA.split("_"))], dtype=torch.int32, device=dev)
            x1, touched = kernels.encode_conv1(rec, self.conv1_hwio, self.conv1_base, touched=True)
            rows = touched[0].cpu().numpy().astype("int64") & ((1 << 27) - 1)
            free = [(y, x) for y in range(27) for x in range(27) if not (int(rows[y]) >> x) & 1]
            if not free:
                raise RuntimeError("the padding record touches every pooled pixel")
            c = x1[0, :, free[0][0], free[0][1], :]                      # [12, 8]: the constant vector
            x = c.reshape(1, 12, 1, 1, 8).expand(1, 12, 27, 27, 8).contiguous()
            bg = {}
            bg["conv2"] = kernels.conv2d_same(x, self.conv2_w, None, groups=2)
            x = kernels.bias_relu_pool_lrn(bg["conv2"], self.conv2_b, lrn=True)
            bg["conv3"] = kernels.conv2d_same(x, self.conv3_w, self.conv3_b, groups=1, relu=True)
            bg["conv4"] = kernels.conv2d_same(bg["conv3"], self.conv4_w, self.conv4_b, groups=2, relu=True)
            bg["conv5"] = kernels.conv2d_same(bg["conv4"], self.conv5_w, None, groups=2)
            self._background = bg
        return self._background

    @torch.no_grad()
    def predict_records_packed(self, records, out=None):
        """records int32 device tensor [B,12] (TSV columns 1..12) -> float32 [B,12] = softmax[5], class, logits[5], 0:
        the three fetches of predict.py:209 in one packed row per image."""
        from .. import kernels
        x = self._convs(records)
        x = x.reshape(x.shape[0], 9216)
        x = kernels.fc_bias_act(x, self.fc6_w, self.fc6_b, relu=True)
        x = kernels.fc_bias_act(x, self.fc7_w, self.fc7_b, relu=True)
        return kernels.fc8_softmax(x, self.fc8_w, self.fc8_b, out=out)

    @torch.no_grad()
    def predict_records(self, records):
        """(logits, argmax, softmax) -- the three fetches of predict.py:209."""
        packed = self.predict_records_packed(records)
        return packed[:, 6:11], packed[:, 5].to(torch.int64), packed[:, :5]
