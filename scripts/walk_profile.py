#!/usr/bin/env python3
"""Diagnostic: where k_wall_tasks and k_find_wall spend their cycles on the WHOLE-PATH call (compact records, find_rel inside the emission loop) (needs build_diag/lib_walk.so, built
with -DCP_PROF_WALK; run with CLASSPRO_AMD_LIB=build_diag/lib_walk.so on the GPU box)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from classpro_amd import synth
from classpro_amd.api import Classifier, Batch, hist_covs, STAGE_REL
from classpro_amd._lib import lib

if len(sys.argv) > 1 and sys.argv[1] == "python-synth":
    ds = synth.make_dataset(genome_len=5_000_000, cov=40, read_len=20000, K=40, het=0.001, n_repeats=62, min_len=3000, seed=1)
    low, high, il, ih, h = ds["hist"]
    hc, dc = hist_covs(h, low, high, il, ih, 0)
    b = Batch.from_reads(ds["seqs"], ds["profiles"])
else:                                                     # the bench's generator: one 800-Mbase sub-batch
    from classpro_amd.synth_dev import DeviceSynth
    sy = DeviceSynth(genome_len=20_000_000, cov=40, read_len=20000, seed=1)
    low, high, il, ih, h = sy.hist
    hc, dc = hist_covs(h, low, high, il, ih, 0)
    b = Batch.from_device(sy.reads(0, sy.n_reads))
clf = Classifier(40, 20000, hc, dc)
ph = (C.c_ulonglong * 36)()
lv = (C.c_ulonglong * 8)()
em = (C.c_ulonglong * 8)()
clf.classify(b)
lib().cp_debug_phase_prof(ph)
lib().cp_debug_live_prof(lv)
lib().cp_debug_emit_prof(em)
clf.classify(b)
lib().cp_debug_phase_prof(ph)
lib().cp_debug_live_prof(lv)
lib().cp_debug_emit_prof(em)
pn = {0: "k_wall_tasks: candidate list", 7: "k_wall_tasks: prelude + filters", 6: "k_wall_tasks: live tasks",
      1: "k_find_wall: replay", 2: "unwall/sort/olist", 3: "multi-error search", 4: "merge + sorts",
      8: "components + boundaries", 5: "records + find_rel"}
print("phase                          max over reads (ticks)   mean      argmax read / its ncand   (100 MHz ticks)")
for k in (0, 7, 6, 1, 2, 3, 4, 8, 5):
    print("  %-34s %12d %12.1f      %d / %d" % (pn[k], ph[k], ph[12 + k] / b.nreads, ph[24 + k] >> 32, ph[24 + k] & 0xffffffff))
print("replay chunks %d (%.1f tasks each): dependency rounds per chunk %.2f with the hashed table, %.2f with exact key comparisons"
      % (lv[2], lv[3] / max(1, lv[2]), lv[0] / max(1, lv[2]), lv[1] / max(1, lv[2])))
print("E-interval lists of at most 64 (the wave-parallel forms): before the un-wall %d reads (mean length %.1f), before the merge %d (%.1f), before the components %d (%.1f)"
      % (em[0], em[1] / b.nreads, em[2], em[3] / b.nreads, em[4], em[5] / b.nreads))
print("classify_unrel (main class): %.1f non-fixed intervals per read, %.1f speculation rounds in the first sweep (four slots per round), %.1f in the second"
      % (ph[12 + 10] / max(1, ph[12 + 11]), ph[12 + 9] / max(1, ph[12 + 11]), ph[9] / max(1, ph[12 + 11])))
print("classify_rel: %d of %d (read, direction) passes are repeated with adjusted coverages (class_rel.c:629-650); %d of %d waves run the DP a second time"
      % (em[6] & 0xffffffff, em[6] >> 32, em[7] & 0xffffffff, em[7] >> 32))
print("reads %d: memo on chip %d, flags on chip to the end %d, sent their flags to the arrays after the replay (more off-list SELF walls than slots) %d, flags on chip after the walk %d" %
      (b.nreads, lv[4], lv[5], lv[6], lv[7]))
