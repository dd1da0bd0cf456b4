#!/usr/bin/env python3
"""Diagnostic: ONE run of the pipeline up to a stage on an 800-Mbase sub-batch of the bench's generator (for counter
passes over truncated kernels, scripts/phase_insts.sh: a truncated kernel leaves its scratch dirty, so one run per process).
    python scripts/stage_once.py [stage=wall|rel|class_rel|class|labels]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from classpro_amd.synth_dev import DeviceSynth
from classpro_amd import api
from classpro_amd.api import Classifier, Batch, hist_covs
stage = {"wall": api.STAGE_WALL, "rel": api.STAGE_REL, "class_rel": api.STAGE_CLASS_REL, "class": api.STAGE_CLASS_ALL, "labels": api.STAGE_LABELS}[sys.argv[1] if len(sys.argv) > 1 else "wall"]
sy = DeviceSynth(genome_len=20_000_000, cov=40, read_len=20000, seed=1)
low, high, il, ih, h = sy.hist
hc, dc = hist_covs(h, low, high, il, ih, 0)
b = Batch.from_device(sy.reads(0, sy.n_reads))
clf = Classifier(40, 20000, hc, dc)
try:
    clf.run(b, stage)
except Exception as ex:                                   # a truncated kernel may leave counts that later stages reject
    print("run:", ex)
torch.cuda.synchronize()
print("reads", b.nreads, "bases", b.total_bases)
