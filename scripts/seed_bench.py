#!/usr/bin/env python3
"""Rate of the -s seed path (cp_find_seeds_batch) on a configs[4]-like set (60x, r=25000), after classification.
    python scripts/seed_bench.py [genome_len=20e6]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from classpro_amd.synth_dev import DeviceSynth
from classpro_amd.api import Classifier, Batch, hist_covs
from classpro_amd._lib import lib, check
import ctypes as C
G = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
s4 = DeviceSynth(genome_len=G, cov=60, read_len=25000, seed=5)
h4, d4 = hist_covs(s4.hist[4], 1, 32767, 0, 0, 0)
c4 = Classifier(K=40, read_len=25000, hcov=h4, dcov=d4)
b4 = Batch.from_device(s4.reads(0, s4.n_reads))
seeds = torch.zeros(b4.total_bases, dtype=torch.uint8, device="cuda:0")
L = lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
c4.run(b4); torch.cuda.synchronize()
for it in range(3):
    t0 = time.perf_counter()
    check(L.cp_find_seeds_batch(c4.p, c4.ws, b4.seq.data_ptr(), b4.seq_off.data_ptr(), b4.prof.data_ptr(), b4.prof_off.data_ptr(),
                                b4.labels.data_ptr(), b4.nreads, b4.total_bases, b4.total_kmers, seeds.data_ptr(), st))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c4.check()
    if hasattr(L, "cp_debug_seed_prof"):
        out = (C.c_ulonglong * 8)()
        L.cp_debug_seed_prof(out)
        print("   lane-0 ms per read (100 MHz ticks): anno %.3f  segments %.3f  window counts %.3f  sort %.3f  whole-window takes %.3f  group walk %.3f" %
              tuple(out[k] / b4.nreads / 1e5 for k in (0, 1, 5, 2, 3, 4)), flush=True)
        print("   per read, summed over the three selections: k-mers of taken segments %.0f  segments %.0f" %
              tuple(out[k] / b4.nreads for k in (6, 7)), flush=True)
    print("seeds pass %d: %.1f ms, %.1f Gbases/s (%d reads, %d bases); seeds %.4f" % (it, dt * 1e3, b4.total_bases / dt / 1e9, b4.nreads, b4.total_bases,
          float((seeds != ord("E")).float().mean())), flush=True)
