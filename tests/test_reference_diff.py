"""Differential check against the reference itself, when it is mounted (this container; skipped elsewhere): a few random
samples and option sets through the reference's run_detect / Predict.run / merge_split_vcfs (imported with stand-ins for
pysam, cv2, tensorflow: tests/golden/refdriver.py) and through the product's host code, outputs compared byte for byte.
Run in subprocesses: the stand-in modules must not leak into the other tests.  tools/diff_ref*.py take a seed range for
longer runs (580 collection cases / 230,000 TSV lines and 230 prediction cases were compared that way in round 1)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference not mounted")


def _run(tool, first, n):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(first), str(n)], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("%d cases, 0 mismatching" % n), r.stdout[-3000:]
    return int(last.split(",")[2].split()[0])


def test_collection_equals_reference_on_random_samples(oracle_lib):
    assert _run("diff_ref.py", 7000, 4) > 200          # TSV lines compared


def test_vote_vcf_merge_equal_reference_on_random_samples(oracle_lib):
    assert _run("diff_ref_predict.py", 7100, 3) > 20   # VCF records compared
