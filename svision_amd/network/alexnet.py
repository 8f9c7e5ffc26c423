"""AlexNet classifier of the similarity images, on PyTorch-ROCm (fp32).

Mirrors the TF1 graph of the reference (src/network/alexnet.py:26-58, layer
helpers :100-170): conv1 11x11/4 VALID 3->96, pool, LRN; conv2 5x5 SAME g=2,
pool, LRN; conv3/4/5 3x3 SAME (g=1,2,2), pool; fc6/fc7 (ReLU), fc8 -> 5 logits
(DEL, INS, INV, DUP, tDUP).  Dropout is the identity at inference
(predict.py:22,210).  Parameters keep the checkpoint's names and layouts
(``convN/weights`` HWIO, ``fcN/weights`` [in,out]) at the ``-m`` boundary and are
re-laid out once for the device:

* conv HWIO -> OIHW (grouped conv = split of input channels and of the output
  axis, identical to ``groups=2``);
* fc6 rows permuted from the reference's NHWC flatten ((h*6+w)*256+c) to NCHW;
* TF LRN (alpha not divided by the window) == torch LRN with alpha*5.

The dense contractions (conv via MIOpen, fc via hipBLASLt) are the only MFMA
users; everything feeding them comes from the hand-written HIP rasteriser.
"""
import os

# MIOpen's default "hybrid" find mode picks Winograd f2x3 for the 13x13 layers; the exhaustive
# find (mode 1) selects faster solvers for these shapes (measured on MI355X: -12 % per batch).
os.environ.setdefault("MIOPEN_FIND_MODE", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

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


class AlexNet(torch.nn.Module):
    """Inference-only AlexNet holding device-layout parameters."""

    def __init__(self, params, device="cuda", channels_last=False, fused=None, mean=(104.0, 117.0, 124.0), own_conv=None,
                 active=None):
        super().__init__()
        # own_conv: layer names whose convolution runs on the hand-written MFMA implicit-GEMM kernel
        # (svx_conv2d_same) instead of MIOpen; default on the GPU: conv2..conv5 (measured 7 % faster per batch)
        if own_conv is None:
            own_conv = ("conv2", "conv3", "conv4", "conv5") if (torch.device(device).type == "cuda" and not channels_last) else ()
        self.own_conv = tuple(own_conv)
        # fused: conv epilogues (bias+relu+pool+LRN) as one hand-written HIP kernel; default on the GPU
        self.fused = (torch.device(device).type == "cuda" and not channels_last) if fused is None else fused
        # active: conv2..conv5 compute only the outputs with a line of the similarity image in their receptive field;
        # all others are copied from the network's (image independent) response to an empty image -- exact, every
        # operation between the first layer and pool5 being local (records path only; default with the own kernels)
        self.active = (len(self.own_conv) == 4 and self.fused) if active is None else active
        self._background = None
        want = checkpoint_shapes()
        missing = [k for k in want if k not in params]
        if missing:
            raise KeyError(f"checkpoint lacks tensors {missing}")   # TF raises NotFoundError
        self.channels_last = channels_last
        for name, _k, _cin, _cout, _s, _p, _g in _CONVS:
            w = np.asarray(params[f"{name}/weights"], np.float32)
            if tuple(w.shape) != want[f"{name}/weights"]:
                raise ValueError(f"{name}/weights has shape {w.shape}, expected {want[f'{name}/weights']}")
            wt = torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1)))          # HWIO -> OIHW
            if channels_last:
                wt = wt.contiguous(memory_format=torch.channels_last)
            self.register_buffer(f"{name}_w", wt.to(device))
            self.register_buffer(f"{name}_b", torch.from_numpy(np.asarray(params[f"{name}/biases"], np.float32).copy()).to(device))
        for name in ("conv2", "conv3", "conv4", "conv5"):          # checkpoint-layout weights for svx_conv2d_same
            self.register_buffer(f"{name}_hwio", torch.from_numpy(np.array(params[f"{name}/weights"], np.float32, copy=True)).to(device))
        # sparse first layer (svx_encode_conv1): checkpoint-layout weights + the constant response of
        # the all-background image, base[k] = bias[k] - sum mean[ch] * w[..., ch, k] (float64 on the host)
        w1 = np.asarray(params["conv1/weights"], np.float64)
        base = np.asarray(params["conv1/biases"], np.float64) - np.einsum("hwck,c->k", w1, np.asarray(mean, np.float64))
        self.register_buffer("conv1_hwio", torch.from_numpy(np.array(params["conv1/weights"], np.float32, copy=True)).to(device))
        self.register_buffer("conv1_base", torch.from_numpy(base.astype(np.float32)).to(device))
        for name, nin, nout in _FCS:
            w = np.asarray(params[f"{name}/weights"], np.float32)
            if tuple(w.shape) != (nin, nout):
                raise ValueError(f"{name}/weights has shape {w.shape}, expected {(nin, nout)}")
            if name == "fc6":
                # rows (h,w,c) -> (c,h,w): our activations are flattened NCHW
                w = w.reshape(6, 6, 256, nout).transpose(2, 0, 1, 3).reshape(nin, nout)
            # store [out,in] for F.linear
            self.register_buffer(f"{name}_w", torch.from_numpy(np.ascontiguousarray(w.T)).to(device))
            self.register_buffer(f"{name}_b", torch.from_numpy(np.asarray(params[f"{name}/biases"], np.float32).copy()).to(device))

    @torch.no_grad()
    def forward_records(self, records):
        """records: int32 device tensor [B,12] (TSV columns 1..12) -> logits [B,5].  The image is never
        materialised: rasterisation + conv1 + relu + pool1 + norm1 run as one sparse HIP kernel."""
        return self._body_records(records, upto_fc7=False)

    def _body_records(self, records, upto_fc7):
        from .. import kernels
        if not self.active:
            x = kernels.encode_conv1(records, self.conv1_hwio, self.conv1_base)
            return self._tail(x, first=1, upto_fc7=upto_fc7)
        bg = self.background()
        n = records.shape[0]
        x, touched = kernels.encode_conv1(records, self.conv1_hwio, self.conv1_base, touched=True)
        l2, l3, l4, l5, counts = kernels.alexnet_active_sets(touched)

        def conv(name, x, pixels, k, bias, relu, groups):
            # active pixels computed, the others copied from the background by the workgroups the active tiles leave over
            return kernels.conv2d_same(x, getattr(self, name + "_hwio"), bias, groups=groups, relu=relu, pixels=pixels,
                                       pixel_count=counts[k:k + 1], background=bg[name])
        x = conv("conv2", x, l2, 0, None, False, 2)
        x = kernels.bias_relu_pool_lrn(x, self.conv2_b, lrn=True)
        x = conv("conv3", x, l3, 1, self.conv3_b, True, 1)
        x = conv("conv4", x, l4, 2, self.conv4_b, True, 2)
        x = conv("conv5", x, l5, 3, None, False, 2)
        x = kernels.bias_relu_pool_lrn(x, self.conv5_b, lrn=False)
        return self._fc(x, upto_fc7)

    @torch.no_grad()
    def background(self):
        """Outputs of conv2..conv5 for an empty image ([1,C,H,W] each), computed once with the same kernels: the first
        layer's constant vector (any pooled pixel without a set tap under it) pushed through the dense path."""
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
            c = x1[0, :, free[0][0], free[0][1]]
            x = c.reshape(1, 96, 1, 1).expand(1, 96, 27, 27).contiguous()
            bg = {}
            bg["conv2"] = kernels.conv2d_same(x, self.conv2_hwio, None, groups=2)
            x = kernels.bias_relu_pool_lrn(bg["conv2"], self.conv2_b, lrn=True)
            bg["conv3"] = kernels.conv2d_same(x, self.conv3_hwio, self.conv3_b, groups=1, relu=True)
            bg["conv4"] = kernels.conv2d_same(bg["conv3"], self.conv4_hwio, self.conv4_b, groups=2, relu=True)
            bg["conv5"] = kernels.conv2d_same(bg["conv4"], self.conv5_hwio, None, groups=2)
            self._background = bg
        return self._background

    @torch.no_grad()
    def predict_records(self, records):
        logits = self.forward_records(records)
        return logits, torch.argmax(logits, dim=1), torch.softmax(logits, dim=1)

    @torch.no_grad()
    def forward(self, x):
        """x: float32 [B,3,227,227] (mean-subtracted, as produced by the rasteriser) -> logits [B,5]."""
        if self.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        return self._tail(x, first=0)

    @torch.no_grad()
    def predict_records_packed(self, records, out=None):
        """records int32 [B,12] -> float32 [B,12] = softmax[5], class, logits[5], 0 (all hand-written kernels except fc6/fc7)."""
        from .. import kernels
        h7 = self._body_records(records, upto_fc7=True)
        return kernels.fc8_softmax(h7, self.fc8_w, self.fc8_b, out=out)

    def _tail(self, x, first, upto_fc7=False):
        for name, _k, _cin, _cout, stride, pad, groups in _CONVS[first:]:
            if name in self.own_conv:
                from .. import kernels
                if name in ("conv2", "conv5"):
                    x = kernels.conv2d_same(x, getattr(self, f"{name}_hwio"), None, groups=groups)
                    x = kernels.bias_relu_pool_lrn(x, getattr(self, f"{name}_b"), lrn=name != "conv5")
                else:
                    x = kernels.conv2d_same(x, getattr(self, f"{name}_hwio"), getattr(self, f"{name}_b"), groups=groups, relu=True)
                continue
            if self.fused and name in ("conv1", "conv2", "conv5"):
                from .. import kernels
                x = F.conv2d(x, getattr(self, f"{name}_w"), None, stride=stride, padding=pad, groups=groups)
                x = kernels.bias_relu_pool_lrn(x, getattr(self, f"{name}_b"), lrn=name != "conv5")
                continue
            if self.fused:
                from .. import kernels
                x = F.conv2d(x, getattr(self, f"{name}_w"), None, stride=stride, padding=pad, groups=groups)
                x = kernels.bias_relu_(x, getattr(self, f"{name}_b"))
                continue
            x = F.relu_(F.conv2d(x, getattr(self, f"{name}_w"), getattr(self, f"{name}_b"),
                                 stride=stride, padding=pad, groups=groups))
            if name in ("conv1", "conv2"):
                x = F.max_pool2d(x, 3, 2)
                x = F.local_response_norm(x, size=5, alpha=2e-05 * 5, beta=0.75, k=1.0)
            elif name == "conv5":
                x = F.max_pool2d(x, 3, 2)
        return self._fc(x, upto_fc7)

    def _fc(self, x, upto_fc7=False):
        x = x.reshape(x.shape[0], 9216) if not self.channels_last else x.contiguous().reshape(x.shape[0], 9216)
        if x.is_cuda:                                       # bias + ReLU in the hipBLASLt epilogue
            x = torch._addmm_activation(self.fc6_b, x, self.fc6_w.t(), use_gelu=False)
            x = torch._addmm_activation(self.fc7_b, x, self.fc7_w.t(), use_gelu=False)
        else:
            x = F.relu_(F.linear(x, self.fc6_w, self.fc6_b))
            x = F.relu_(F.linear(x, self.fc7_w, self.fc7_b))
        if upto_fc7:
            return x
        return F.linear(x, self.fc8_w, self.fc8_b)

    @torch.no_grad()
    def predict(self, x):
        """(logits, argmax, softmax) -- the three fetches of predict.py:209."""
        logits = self.forward(x)
        return logits, torch.argmax(logits, dim=1), torch.softmax(logits, dim=1)
