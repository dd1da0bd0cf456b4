#!/usr/bin/env python3
"""Diagnostic: distribution of wall candidates / intervals / reliable intervals per read on the bench batch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from classpro_amd import synth
from classpro_amd.api import Classifier, Batch, hist_covs, STAGE_REL
ds = synth.make_dataset(genome_len=5_000_000, cov=40, read_len=20000, K=40, het=0.001, n_repeats=62, min_len=3000, seed=1)
low, high, il, ih, h = ds["hist"]
hc, dc = hist_covs(h, low, high, il, ih, 0)
clf = Classifier(40, 20000, hc, dc)
b = Batch.from_reads(ds["seqs"], ds["profiles"])
clf.run(b, STAGE_REL)
nc, ni, nr, off = clf.counts(b)
q = [0, 10, 50, 90, 95, 99, 99.9, 100]
for nm, v in (("ncand", nc), ("N intervals", ni), ("M reliable", nr)):
    print("%-12s" % nm, " ".join("%7.0f" % x for x in np.percentile(v, q)), "  mean %.1f" % v.mean())
print("percentiles  ", " ".join("%7s" % x for x in q))
print("M>128:", int((nr > 128).sum()), " N>192:", int((ni > 192).sum()), " N>256:", int((ni > 256).sum()), " M>128 & N<=192:", int(((nr > 128) & (ni <= 192)).sum()))
print("corr(ncand,N)=%.3f corr(ncand,M)=%.3f" % (np.corrcoef(nc, ni)[0, 1], np.corrcoef(nc, nr)[0, 1]))
top = np.argsort(-nr)[:5]
print("top M reads:", [(int(nr[i]), int(ni[i]), int(nc[i])) for i in top])
top = np.argsort(-ni)[:5]
print("top N reads:", [(int(nr[i]), int(ni[i]), int(nc[i])) for i in top])
