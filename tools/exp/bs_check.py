import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svision_amd import cli
from svision_amd.io import bam
from svision_amd.network import tf_checkpoint as ck
from oracle import alexnet_ref
from tests import helpers
tmp = tempfile.mkdtemp()
prefix = os.path.join(tmp, "m.ckpt")
ck.write_checkpoint(prefix, alexnet_ref.random_params(seed=7))
fasta = helpers.load_golden_fasta()
fa = os.path.join(tmp, "g.fa")
bam.write_fasta(fa, {n: fasta._seq[n] for n in fasta.references})
outs = {}
for bs in (16, 32, 64, 128, 200):
    out = os.path.join(tmp, "o%d" % bs)
    opts = cli.parse_arguments(["-o", out, "-b", os.path.join(helpers.GOLDEN, "collect_small.bam"), "-m", prefix, "-g", fa, "-n", "S", "-s", "3",
                                "--window_size", "60000", "--batch_size", str(bs)])
    merged = cli.run(opts)
    outs[bs] = open(merged).read()
    print("batch", bs, "records", outs[bs].count("\n"))
print("all equal:", len(set(outs.values())) == 1)
