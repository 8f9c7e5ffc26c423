"""Where does the command line's 0.86 s before its first window go?  (checkpoint read, weight packing + upload, graph capture)"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
t0 = time.time(); import torch; torch.zeros(1, device="cuda"); torch.cuda.synchronize(); print("import torch + context %.3f s" % (time.time() - t0))
from bench import random_weights
from svision_amd.network import tf_checkpoint as ck
from svision_amd.network.predict import load_network
from svision_amd import kernels, _lib
d = tempfile.mkdtemp(); prefix = os.path.join(d, "m.ckpt"); ck.write_checkpoint(prefix, random_weights(0))
t = time.time(); _lib.load(); print("libsvx load %.3f" % (time.time() - t))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t = time.time(); net = load_network(prefix); torch.cuda.synchronize(); t1 = time.time(); print("load_network %.3f" % (t1 - t))
from svision_amd.pipeline import HotPath
from tests import helpers
opts = helpers.default_options(min_support=3, batch_size=64, bam_path="<x>")
hot = HotPath(None, opts, net, n_streams=3); torch.cuda.synchronize(); print("HotPath (graphs) %.3f" % (time.time() - t1))
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
