"""ORACLE (test infrastructure, not product): the AlexNet of src/network/alexnet.py:26-58 as plain PyTorch fp32
ops on dense mean-subtracted images (NCHW), on the CPU or on a GPU through the vendor libraries.

Used by tests/ as an independent cross-check of the hand-written kernels, and by bench.py's ``cpu_baseline``
leg as the CPU port of the classifier (the reference runs CPU TensorFlow 1.14, which is not installed here).
The NumPy restatement with TF's documented op semantics is oracle/alexnet_ref.py; the two are compared with
each other in tests/test_oracle_cpu.py.  TF LRN (alpha not divided by the window) == torch LRN with alpha*5.
"""
import numpy as np
import torch
import torch.nn.functional as F

_CONVS = (("conv1", 4, 0, 1), ("conv2", 1, 2, 2), ("conv3", 1, 1, 1), ("conv4", 1, 1, 2), ("conv5", 1, 1, 2))


class TorchAlexNet:
    def __init__(self, params, device="cpu"):
        self.p = {}
        for name, _s, _p, _g in _CONVS:
            w = np.asarray(params[f"{name}/weights"], np.float32)
            self.p[name + "_w"] = torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1))).to(device)     # HWIO -> OIHW
            self.p[name + "_b"] = torch.from_numpy(np.array(params[f"{name}/biases"], np.float32, copy=True)).to(device)
        for name in ("fc6", "fc7", "fc8"):
            w = np.asarray(params[f"{name}/weights"], np.float32)
            if name == "fc6":                                      # reference flatten is NHWC, ours NCHW
                w = w.reshape(6, 6, 256, -1).transpose(2, 0, 1, 3).reshape(9216, -1)
            self.p[name + "_w"] = torch.from_numpy(np.ascontiguousarray(w.T)).to(device)
            self.p[name + "_b"] = torch.from_numpy(np.array(params[f"{name}/biases"], np.float32, copy=True)).to(device)

    @torch.no_grad()
    def forward(self, x):
        """x float32 [B,3,227,227] mean-subtracted -> logits [B,5]."""
        p = self.p
        for name, stride, pad, groups in _CONVS:
            x = F.relu_(F.conv2d(x, p[name + "_w"], p[name + "_b"], stride=stride, padding=pad, groups=groups))
            if name in ("conv1", "conv2"):
                x = F.local_response_norm(F.max_pool2d(x, 3, 2), size=5, alpha=2e-05 * 5, beta=0.75, k=1.0)
            elif name == "conv5":
                x = F.max_pool2d(x, 3, 2)
        x = x.reshape(x.shape[0], 9216)
        x = F.relu_(F.linear(x, p["fc6_w"], p["fc6_b"]))
        x = F.relu_(F.linear(x, p["fc7_w"], p["fc7_b"]))
        return F.linear(x, p["fc8_w"], p["fc8_b"])

    def predict(self, x):
        logits = self.forward(x)
        return logits, torch.argmax(logits, dim=1), torch.softmax(logits, dim=1)
