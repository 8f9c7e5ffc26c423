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
        # the device-layout weights of this checkpoint as an earlier process left them (weight_cache.py: digest of the bundle's
        # bytes in the file name), else the bundle itself -- and the layouts are left behind for the next process
        from . import weight_cache
        net = path = None
        if weight_cache.enabled():
            from .. import _lib
            digest = weight_cache.checkpoint_digest(model_path, abi=_lib.ABI_VERSION)
            if digest is not None:
                path = weight_cache.cache_path(model_path, digest)
                packed = weight_cache.load(path)
                if packed is not None:
                    try:
                        net = AlexNet(None, device=device, packed=packed)
                    except (ValueError, RuntimeError):
                        net = None
        if net is None:
            net = AlexNet(read_checkpoint(model_path), device=device)
            if path is not None:
                weight_cache.save(path, net.packed_tensors())
        _MODEL_CACHE[key] = net
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
    region change and at the end.

    ``hold_edges``: the voter sees ONE collection window of a chromosome.  The reference votes over the concatenation of
    the windows' TSVs (SVision:284-288 -> predict.py:235-247), where a region string that ends window k and opens
    window k+1 (reads overlapping the boundary are collected by both, run_collection.py:26) is a single site.  So the
    first and the last site of the window are not written: their accepted (label, class, score) items are kept in
    ``head`` / ``tail`` and replayed, in window order, through the chromosome's own voter (:class:`ChromosomeVote`)."""

    def __init__(self, predictor, vcf_out, score_out, options, sample, hold_edges=False, hold_range=None):
        self.predictor, self.vcf_out, self.score_out = predictor, vcf_out, score_out
        self.options, self.sample = options, sample
        self.site = _SiteState()
        self.n_sites = 0
        self.hold_edges, self.head, self.tail, self._seen_first = hold_edges, None, None, False
        # hold_range = (lo, hi): an edge site is only held back when it can reach into the neighbouring window -- the
        # first site when its start is <= lo, the last one when its end is >= hi (pipeline._vote derives both from the
        # longest alignment of the sample); any other edge site is written at once like an interior one
        self.hold_range = hold_range
        # file-driven runs hand a chromosome over in slices (svision_amd/ingest.py): ``next_sample`` = the Sample of the window
        # whose lines are being fed.  A site is written on the records of the window that reported its FIRST line (a site that
        # continues into the next window lies within reach of both, Sample.reach()).
        self.next_sample = None

    def feed_batch(self, labels, classes, probs):
        classes = np.asarray(classes)
        probs = np.asarray(probs)
        # round(probs[i][cls], 2) of predict.py:270 for the whole batch at once (np.float32.__round__ is np.round)
        scores = np.round(probs[np.arange(len(labels)), classes[:len(labels)].astype(np.int64)], 2) if len(labels) else probs[:0, 0]
        self.feed_items(zip(labels, classes, scores))

    def feed_items(self, items):
        site = self.site
        for label, cls, score in items:
            if "complement" in label:
                continue
            f = label.split("svision")
            read_num, region, read_name = f[0], f[1], f[2]
            cls = int(cls)
            if f[7] == "True" and cls == 2:                           # :229-231 forward pairs cannot be INV
                continue
            if region != site.region:
                if site.region != "":
                    self._flush(site)
                if self.next_sample is not None:
                    self.sample = self.next_sample
                site = self.site = _SiteState(region)
            if self.hold_edges:
                site.items.append((label, cls, score))
            rid = read_num.replace("m", "")
            site.read_names[rid] = read_name
            site.sig_types.append(f[3])
            site.predict_scores.append(score)
            site.sig_scores[rid] = f[6]
            site.mechanisms[rid] = f[8]
            if "m" not in read_num and cls in (0, 1):                 # :278-280 only main x main pairs call INS/DEL
                continue
            site.reads.setdefault(rid, {})[cls] = [int(f[4]), int(f[5]), int(f[9])]

    def close_site(self):
        """Write the open site, if any (the next lines belong to other sites)."""
        if self.site.region != "":
            self._flush(self.site)
        self.site = _SiteState()

    @staticmethod
    def _span(region):
        f = region.rsplit("+", 3)                                 # chrom+start+end+coverage; the contig name may hold a '+'
        return int(f[1]), int(f[2])

    def finish(self):
        if self.hold_edges and self.site.region != "":
            lo, hi = self.hold_range if self.hold_range is not None else (float("inf"), float("-inf"))
            start, end = self._span(self.site.region)
            if self._seen_first:
                if end >= hi:
                    self.tail = self.site.items                       # last site of the window: may continue in the next one
                else:
                    self._flush(self.site)
            elif start <= lo or end >= hi:
                self.head = self.site.items                           # the window's only site: first and last at once
                self.n_sites += 1
            else:
                self._flush(self.site)
        elif not self.hold_edges:
            self._flush(self.site)
        self.site = _SiteState()

    def _flush(self, site):
        self.n_sites += 1 if site.region != "" else 0
        if self.hold_edges and not self._seen_first and site.region != "":
            self._seen_first = True
            if self.hold_range is None or self._span(site.region)[0] <= self.hold_range[0]:
                self.head = site.items                                # first site of the window: may continue the previous one
                return
        write_results_to_vcf(self.vcf_out, self.score_out, self.predictor.get_region_potential_svtypes(site.reads),
                             site.region, site.read_names, site.sig_types, site.sig_scores, site.predict_scores,
                             site.mechanisms, self.options, self.sample)


class ChromosomeVote:
    """One chromosome's vote over per-window results (SiteVoter(hold_edges=True)) delivered in task order: interior
    sites were written by the window's own voter; the edge sites are replayed here, so a site that spans a window
    boundary is written once, exactly as a vote over the concatenated TSV writes it."""

    def __init__(self, chrom, vcf_out, score_out, options, sample):
        self.vcf_out, self.score_out = vcf_out, score_out
        self.voter = SiteVoter(Predict(chrom, None), vcf_out, score_out, options, sample)

    def add(self, head, vcf_text, score_text, tail, sample=None):
        """``sample``: the Sample that served this window (a slice of the chromosome in a file-driven run): a site opened by this
        window's head or tail is written on its records; a site that the previous window's tail opened and this window's head
        continues stays with the previous window's (it lies within reach of both)."""
        self.voter.next_sample = sample
        if head:
            self.voter.feed_items(head)
        if vcf_text or score_text or tail:
            self.voter.close_site()
            self.vcf_out.write(vcf_text)
            self.score_out.write(score_text)
            if tail:
                self.voter.feed_items(tail)

    def finish(self):
        self.voter.close_site()


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
        self.items = []                  # accepted (label, class, score) of the site (SiteVoter(hold_edges=True) only)
