"""In-memory streaming form of the hot path: one collection window at a time, no
intermediate files, host work of window k+1 overlapped with the device work of window k.

Same stages and the same (parity-tested) functions as the file-based driver (cli.py):

  device  svx_cigar_scan over the window's alignment block          (collection, analyze_reads.py:828-853)
  host    reads -> segments -> signatures -> clusters -> pair lines   (run_collection.py:15-47)
  device  svx_rasterize + AlexNet, batches of `batch_size` images     (create_batch.py:88, predict.py:206-210)
  host    per-site vote -> VCF body lines + scores                    (predict.py:213-300, output.py:469)

Used by bench.py (throughput) and available to embedders; the CLI keeps the reference's
on-disk TSV boundary.
"""
import io

import numpy as np
import torch

from . import kernels
from .collection.output_clusters import collect_pair_lines
from .collection.run_collection import detect_window
from .network.create_batch import PAD_DATA, parse_data_fields
from .network.predict import Predict, SiteVoter

_PAD_REC = parse_data_fields(PAD_DATA.split("_"))


class WindowResult:
    __slots__ = ("chrom", "start", "end", "lines", "n_images", "packed", "vcf", "scores", "n_sites", "n_records")


class HotPath:
    def __init__(self, sample, options, net, device="cuda"):
        self.sample, self.options, self.net = sample, options, net
        self.device = torch.device(device)
        self.batch = options.batch_size

    # ---- stage 1+2: device scan of the window's rows + host collection ---------------------
    def collect(self, chrom, start, end, rescan=True):
        res = WindowResult()
        res.chrom, res.start, res.end = chrom, start, end
        if rescan:
            self.sample.rescan_window(chrom, start, end)
        _sigs, clusters = detect_window(self.options, self.sample, chrom, start, end)
        res.lines = collect_pair_lines(clusters, self.options)
        res.n_images = len(res.lines)
        res.packed = None
        return res

    # ---- stage 3: encode + CNN, enqueued asynchronously on the current stream -----------------
    def launch(self, res):
        n = res.n_images
        if n == 0:
            return res
        b = self.batch
        pad = (-n) % b
        recs = np.asarray([ln.record() for ln in res.lines] + [_PAD_REC] * pad, np.int32)
        d_rec = torch.from_numpy(recs).to(self.device, non_blocking=True)
        outs = []
        for lo in range(0, n + pad, b):
            img = kernels.rasterize(d_rec[lo:lo + b], layout="NCHW")
            logits, cls, prob = self.net.predict(img)
            outs.append(torch.cat([prob, cls.to(prob.dtype).unsqueeze(1)], dim=1))
        res.packed = torch.cat(outs, dim=0)
        return res

    # ---- stage 4: one D2H copy, per-site vote, VCF body lines -----------------------------------
    def finish(self, res):
        vcf, score = io.StringIO(), io.StringIO()
        res.n_sites = 0
        if res.n_images:
            packed = res.packed.cpu().numpy()
            probs, classes = packed[:, :5], packed[:, 5].astype(np.int64)
            voter = SiteVoter(Predict(res.chrom, None), vcf, score, self.options, self.sample)
            labels = [ln.label() for ln in res.lines]
            voter.feed_batch(labels, classes[:len(labels)], probs[:len(labels)])
            voter.finish()
            res.n_sites = voter.n_sites
        res.vcf, res.scores = vcf.getvalue(), score.getvalue()
        res.n_records = res.vcf.count("\n")
        res.packed = None
        return res

    def run_windows(self, windows):
        """Software pipeline over [(chrom, start, end)]: the CNN of window k runs on the device
        while the host collects window k+1.  Yields WindowResults in order."""
        prev = None
        for chrom, start, end in windows:
            cur = self.collect(chrom, start, end)
            if prev is not None:
                yield self.finish(prev)
            prev = self.launch(cur)
        if prev is not None:
            yield self.finish(prev)
