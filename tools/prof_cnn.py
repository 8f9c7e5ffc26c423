#!/usr/bin/env python3
"""Run N eager batches of the device stage (for rocprofv3 --kernel-trace --stats)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import random_weights
from svision_amd.network.alexnet import AlexNet
from tests import datagen
dev = torch.device("cuda:0")
net = AlexNet(random_weights(0), device=dev)
rec = torch.from_numpy(datagen.random_records(64, seed=1, hostile=False)).to(dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    net.predict_records(rec)
torch.cuda.synchronize()
