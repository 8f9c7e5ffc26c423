"""Segment TSV -> batches of similarity images on the device.

Mirror of the reference's ``BatchGenerator`` (src/network/create_batch.py:11-155): same
constructor arguments, same padding with ``complement`` rows to a multiple of the batch
size (:54-59), same label strings (:45-49).  ``next_batch`` hands the packed 12-int records
to the HIP rasteriser (``svx_rasterize``) instead of looping over images in Python (:103-152).
"""
import math

import numpy as np
import torch

from .. import kernels

PAD_DATA = "0_1_0_1_True_1_1_1_1_True_2_2"        # create_batch.py:55
PAD_LABEL = "complement-complement"               # create_batch.py:56


def parse_data_fields(fields):
    """12 TSV strings (columns 1..12) -> 12 ints; a strand that is neither 'True' nor 'False'
    is None in the reference (:111-116) and therefore drawn as reverse."""
    f = fields
    return (int(f[0]), int(f[1]), int(f[2]), int(f[3]), 1 if f[4] == "True" else 0,
            int(f[5]), int(f[6]), int(f[7]), int(f[8]), 1 if f[9] == "True" else 0, int(f[10]), int(f[11]))


class BatchGenerator:
    def __init__(self, segments_file, horizontal_flip=False, shuffle=False, mean=np.array([104., 117., 124.]),
                 scale_size=(227, 227), nb_classes=2, batch_size=128, device="cuda", layout="NHWC"):
        if horizontal_flip or shuffle:
            raise NotImplementedError("training-time augmentation is outside the inference hot path")
        if tuple(scale_size) != (227, 227):
            raise ValueError("the similarity image is 227x227")
        self.mean = tuple(float(m) for m in mean)
        self.n_classes = nb_classes
        self.batch_size = batch_size
        self.pointer = 0
        self.device = torch.device(device)
        self.layout = layout
        self.read_class_list(segments_file)

    def read_class_list(self, segments_file):
        """Parse the TSV (:29-61): data = columns 1..12, label = 10 columns joined by 'svision'."""
        self.images, self.labels, recs = [], [], []
        if isinstance(segments_file, str):
            with open(segments_file) as f:
                lines = f.readlines()
        else:
            lines = list(segments_file)                      # already-split text lines
        for line in lines:
            c = line.strip("\n").split("\t")
            self.images.append("_".join(c[1:13]))
            self.labels.append("svision".join([c[13], c[0], c[15], c[16], c[17], c[18], c[19], c[20], c[21], c[22]]))
            recs.append(parse_data_fields(c[1:13]))
        n = len(self.labels)
        pad = self.batch_size * math.ceil(n / self.batch_size) - n
        self.images += [PAD_DATA] * pad
        self.labels += [PAD_LABEL] * pad
        recs += [parse_data_fields(PAD_DATA.split("_"))] * pad
        self.data_size = len(self.labels)
        self.records = np.asarray(recs, np.int32).reshape(-1, 12)
        self._d_records = None

    def reset_pointer(self):
        self.pointer = 0

    def next_records(self, batch_size):
        """-> (records int32 device tensor [B,12], labels): the packed form of the next batch, for the
        classifier that fuses encoding and the first CNN layer (svx_encode_conv1)."""
        if self._d_records is None:
            self._d_records = torch.from_numpy(self.records).to(self.device)
        lo = self.pointer
        self.pointer += batch_size
        return self._d_records[lo:lo + batch_size], self.labels[lo:lo + batch_size]

    def next_labels(self, batch_size):
        """Advance the pointer without rasterising (labels only)."""
        lo = self.pointer
        self.pointer += batch_size
        return self.labels[lo:lo + batch_size]

    def next_batch(self, batch_size):
        """-> (images, labels): float32 device tensor [B,227,227,3] (``layout='NHWC'``, the
        reference's batch layout) or [B,3,227,227] (``'NCHW'``), mean-subtracted."""
        if self._d_records is None:                          # one upload for the whole file
            self._d_records = torch.from_numpy(self.records).to(self.device)
        lo = self.pointer
        self.pointer += batch_size
        rec = self._d_records[lo:lo + batch_size]
        return kernels.rasterize(rec, layout=self.layout, mean=self.mean), self.labels[lo:lo + batch_size]
