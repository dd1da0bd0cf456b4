#!/usr/bin/env python3
"""Diagnostic: distribution of wall candidates / intervals / reliable intervals per read on the bench batch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from classpro_amd import synth
from classpro_amd.api import Classifier, Batch, hist_covs, STAGE_REL
if len(sys.argv) > 1 and sys.argv[1] == "python-synth":
    ds = synth.make_dataset(genome_len=5_000_000, cov=40, read_len=20000, K=40, het=0.001, n_repeats=62, min_len=3000, seed=1)
    low, high, il, ih, h = ds["hist"]
    b = Batch.from_reads(ds["seqs"], ds["profiles"])
else:                                                     # the bench's generator
    from classpro_amd.synth_dev import DeviceSynth
    sy = DeviceSynth(genome_len=20_000_000, cov=40, read_len=20000, seed=1)
    low, high, il, ih, h = sy.hist
    b = Batch.from_device(sy.reads(0, sy.n_reads))
hc, dc = hist_covs(h, low, high, il, ih, 0)
clf = Classifier(40, 20000, hc, dc)
clf.run(b, STAGE_REL)
nc, ni, nr, off = clf.counts(b)
q = [0, 10, 50, 90, 95, 99, 99.9, 100]
for nm, v in (("ncand", nc), ("N intervals", ni), ("M reliable", nr)):
    print("%-12s" % nm, " ".join("%7.0f" % x for x in np.percentile(v, q)), "  mean %.1f" % v.mean())
print("percentiles  ", " ".join("%7s" % x for x in q))
print("M>128:", int((nr > 128).sum()), " N>192:", int((ni > 192).sum()), " N>256:", int((ni > 256).sum()), " M>128 & N<=192:", int(((nr > 128) & (ni <= 192)).sum()))
print("corr(ncand,N)=%.3f corr(ncand,M)=%.3f" % (np.corrcoef(nc, ni)[0, 1], np.corrcoef(nc, nr)[0, 1]))
for nm, v in (("ncand", nc), ("N intervals", ni), ("M reliable", nr)):
    print("%-12s 64-lane steps: mean %.2f, mean fill of a step %.2f" % (nm, np.ceil(v / 64).mean(), (v / 64).sum() / np.ceil(v / 64).sum()))
try:                                                      # a -DCP_PROF_WALK build (CLASSPRO_AMD_LIB=build_diag/lib_walk.so) also says the task counts
    import ctypes as C
    from classpro_amd._lib import lib
    fw = np.zeros((b.nreads, 4), np.int32)
    if lib().cp_debug_task_counts(clf.ws, fw.ctypes.data_as(C.c_void_p), C.c_int64(b.nreads)) == 0:
        nt, ns = fw[:, 1], fw[:, 2]
        print("%-12s" % "tasks", " ".join("%7.0f" % x for x in np.percentile(nt, q)), "  mean %.1f  (SELF %.1f)" % (nt.mean(), ns.mean()))
        print("%-12s 64-lane steps: mean %.2f, mean fill of a step %.2f; reads with 65-80 tasks: %.1f %%" %
              ("tasks", np.ceil(nt / 64).mean(), (nt / 64).sum() / np.ceil(nt / 64).sum(), 100. * ((nt > 64) & (nt <= 80)).mean()))
except AttributeError:
    pass
top = np.argsort(-nr)[:5]
print("top M reads:", [(int(nr[i]), int(ni[i]), int(nc[i])) for i in top])
top = np.argsort(-ni)[:5]
print("top N reads:", [(int(nr[i]), int(ni[i]), int(nc[i])) for i in top])
