"""CNN prediction over one chromosome's segment TSV, per-site vote, VCF body output.

Mirror of the reference's ``Predict`` (src/network/predict.py:14-303): same constructor,
``run(out_path_prefix, options)`` writes ``{prefix}.vcf`` and ``{prefix}.score.txt``.  The
TF1 session (:165-210) is replaced by the PyTorch-ROCm AlexNet fed directly by the HIP
rasteriser; the ``-m`` checkpoint prefix is read with :mod:`tf_checkpoint`.
"""
import logging

import numpy as np

from .create_batch import BatchGenerator
from .output import write_results_to_vcf

_TYPE_NAMES = {"0": "DEL", "1": "INS", "2": "INV", "3": "DUP", "4": "tDUP"}
_MODEL_CACHE = {}


def load_network(model_path, device="cuda"):
    from .alexnet import AlexNet
    from .tf_checkpoint import read_checkpoint
    key = (model_path, str(device))
    if key not in _MODEL_CACHE:
        _MODEL_CACHE[key] = AlexNet(read_checkpoint(model_path), device=device)
    return _MODEL_CACHE[key]


class RecordClassifier:
    """callable(records int32 device tensor [B,12]) -> (logits, argmax, softmax) numpy arrays.  Encoding is fused
    into the first layer (svx_encode_conv1): the images of create_batch.py:103-152 are never materialised."""
    from_records = True

    def __init__(self, net):
        self.net = net

    def __call__(self, records):
        packed = self.net.predict_records_packed(records).cpu().numpy()          # one D2H copy per batch
        return packed[:, 6:11], packed[:, 5].astype(np.int64), packed[:, :5]


def load_classifier(model_path, device="cuda"):
    return RecordClassifier(load_network(model_path, device))


class Predict:
    def __init__(self, chrom, segments_out_file):
        self.segments_out_file = segments_out_file
        self.chrom = chrom
        self.dropout_rate = 1.
        self.num_classes = 5

    def get_region_potential_svtypes(self, reads_dict):
        """Group a site's reads by their set of predicted classes and average breakpoints (:87-145).
        reads_dict: {read_id: {class: [start, end, len]}} -> [(type string, [read ids], [[s,e,len],...])]"""
        stats = {}
        for read_id, infos in reads_dict.items():
            key = "".join(str(c) for c in sorted(infos.keys()))
            bkps = [infos[int(ch)] for ch in key]
            if key not in stats:
                stats[key] = [[read_id], bkps]
                continue
            old = stats[key][1]
            n = len(stats[key][0])
            stats[key][1] = [[int((b[0] + o[0] * n) / (n + 1)), int((b[1] + o[1] * n) / (n + 1)), int((b[2] + o[2] * n) / (n + 1))]
                             for b, o in zip(bkps, old)]
            stats[key][0].append(read_id)
        ranked = sorted(stats.items(), key=lambda kv: len(kv[1][0]), reverse=True)
        return [("+".join(_TYPE_NAMES.get(ch, "") for ch in key), reads, bkps) for key, (reads, bkps) in ranked]

    def run(self, out_path_prefix, options, classifier=None, sample=None):
        from .. import sample as _sample
        if sample is None:
            sample = _sample.resolve(options.bam_path, options)
        batch_size = options.batch_size
        gen = BatchGenerator(self.segments_out_file, shuffle=False, nb_classes=self.num_classes, batch_size=batch_size,
                             layout="NCHW")
        if classifier is None:
            classifier = load_classifier(options.model_path)
        n_batches = np.floor(gen.data_size / batch_size).astype(np.int16)        # predict.py:175 (int16, as upstream)
        if int(n_batches) * batch_size != gen.data_size:
            logging.warning("batch count wrapped in int16 (%d images): tail images are skipped, as upstream", gen.data_size)
        with open(out_path_prefix + ".score.txt", "w") as score_out, open(out_path_prefix + ".vcf", "w") as vcf_out:
            voter = SiteVoter(self, vcf_out, score_out, options, sample)
            logging.info("Predicting " + self.chrom)
            for _ in range(n_batches):
                if getattr(classifier, "from_records", False):
                    images, labels = gen.next_records(batch_size)
                elif getattr(classifier, "needs_images", True):
                    images, labels = gen.next_batch(batch_size)
                else:                                                        # predictions injected by a test
                    images, labels = None, gen.next_labels(batch_size)
                _logits, classes, probs = classifier(images)
                voter.feed_batch(labels, classes, probs)
            voter.finish()


class SiteVoter:
    """The per-image bookkeeping of predict.py:213-300: consumes (label, class, softmax row) in
    TSV order, groups by region, and emits one site through write_results_to_vcf at every
    region change and at the end."""

    def __init__(self, predictor, vcf_out, score_out, options, sample):
        self.predictor, self.vcf_out, self.score_out = predictor, vcf_out, score_out
        self.options, self.sample = options, sample
        self.site = _SiteState()
        self.n_sites = 0

    def feed_batch(self, labels, classes, probs):
        site = self.site
        classes = np.asarray(classes)
        probs = np.asarray(probs)
        # round(probs[i][cls], 2) of predict.py:270 for the whole batch at once (np.float32.__round__ is np.round)
        scores = np.round(probs[np.arange(len(labels)), classes[:len(labels)].astype(np.int64)], 2) if len(labels) else probs[:0, 0]
        for i, label in enumerate(labels):
            if "complement" in label:
                continue
            f = label.split("svision")
            read_num, region, read_name = f[0], f[1], f[2]
            cls = int(classes[i])
            if f[7] == "True" and cls == 2:                           # :229-231 forward pairs cannot be INV
                continue
            if region != site.region:
                if site.region != "":
                    self._flush(site)
                site = self.site = _SiteState(region)
            rid = read_num.replace("m", "")
            site.read_names[rid] = read_name
            site.sig_types.append(f[3])
            site.predict_scores.append(scores[i])
            site.sig_scores[rid] = f[6]
            site.mechanisms[rid] = f[8]
            if "m" not in read_num and cls in (0, 1):                 # :278-280 only main x main pairs call INS/DEL
                continue
            site.reads.setdefault(rid, {})[cls] = [int(f[4]), int(f[5]), int(f[9])]

    def finish(self):
        self._flush(self.site)
        self.site = _SiteState()

    def _flush(self, site):
        self.n_sites += 1 if site.region != "" else 0
        write_results_to_vcf(self.vcf_out, self.score_out, self.predictor.get_region_potential_svtypes(site.reads),
                             site.region, site.read_names, site.sig_types, site.sig_scores, site.predict_scores,
                             site.mechanisms, self.options, self.sample)


class _SiteState:
    """Everything predict.py accumulates between two region changes (:186-203, :240-246)."""

    def __init__(self, region=""):
        self.region = region
        self.reads = {}
        self.read_names = {}
        self.sig_scores = {}
        self.mechanisms = {}
        self.sig_types = []
        self.predict_scores = []
