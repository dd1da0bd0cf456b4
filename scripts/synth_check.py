#!/usr/bin/env python3
"""Bring-up check of the device-side synthesiser (classpro_amd/synth_dev.py) on a GPU box:
generation time, histogram peaks, label parity with the oracle on a sample, accuracy against the
generator's ground truth, and the sub-batched classification rate on the configs[2]-sized set.

    python scripts/synth_check.py [genome_len] [batch_mbases]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from classpro_amd.synth_dev import DeviceSynth
from classpro_amd.api import Classifier, Batch, hist_covs
from oracle.oracle import Oracle

G = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000
BATCH = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 800_000_000
K = 40


def truth_acc(lab, rd):
    tmap = torch.full((256,), ord("R"), dtype=torch.uint8, device=lab.device)
    tmap[0], tmap[1], tmap[2] = ord("E"), ord("H"), ord("D")
    kpos = torch.ones(rd["total_bases"], dtype=torch.bool, device=lab.device)
    so = rd["seq_off"][:-1]
    for k in range(K - 1):
        kpos[so + k] = False
    est = lab[:rd["total_bases"]][kpos]
    return float((est == tmap[rd["truth"].long()]).float().mean())


t0 = time.time()
ds = DeviceSynth(genome_len=G, cov=40, read_len=20000, seed=1)
torch.cuda.synchronize()
print("synth setup: G=%d reads=%d bases=%d in %.2f s; err k-mers %d" % (G, ds.n_reads, ds.total_bases, time.time() - t0, ds.n_err_kmers), flush=True)
low, high, il, ih, h = ds.hist
hcov, dcov = hist_covs(h, low, high, il, ih, 0)
print("hist peaks: H=%d D=%d; hist[1..5]=%s hist[15..25]=%s hist[35..45]=%s" % (hcov, dcov, h[0:5], h[14:25], h[34:45]), flush=True)

clf = Classifier(K=K, read_len=20000, hcov=hcov, dcov=dcov)
plan = ds.plan_batches(BATCH)
print("batches:", len(plan), plan[:3], flush=True)
t0 = time.time()
batches = []
for (a, n) in plan:
    rd = ds.reads(a, n, truth=(a == 0))
    batches.append((rd, Batch.from_device(rd)))
torch.cuda.synchronize()
ds.check()
print("generated all reads in %.2f s" % (time.time() - t0), flush=True)
rd0, b0 = batches[0]
print("read 0: rlen %d seq %s prof %s" % (rd0["seq_off_h"][1], bytes(rd0["seq"][:60].cpu().numpy()), rd0["prof"][:30].cpu().numpy().view(np.uint16)), flush=True)

for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rd, b in batches:
        clf.run(b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    clf.check()
    print("pass %d: %.1f ms, %.1f Gbases/s" % (it, dt * 1e3, ds.total_bases / dt / 1e9), flush=True)

# alphabet / N-prefix on everything
ok = True
for rd, b in batches:
    lab = b.labels[:b.total_bases]
    cnt = torch.bincount(lab.long(), minlength=256)
    known = sum(int(cnt[ord(c)]) for c in "NEHDR")
    nN = int(cnt[ord("N")])
    ok &= known == b.total_bases and nN == b.nreads * (K - 1)
print("alphabet + N-prefix count ok:", ok, flush=True)
print("accuracy vs truth (batch 0): %.5f" % truth_acc(b0.labels, rd0), flush=True)
cnt = torch.bincount(b0.labels[:b0.total_bases].long(), minlength=256)
print("label mix batch 0: " + " ".join("%s=%.3f" % (c, int(cnt[ord(c)]) / b0.total_bases) for c in "NEHDR"), flush=True)

# oracle on the first reads of batch 0
ns = min(10000, b0.nreads)
so, po = rd0["seq_off_h"][:ns + 1], rd0["prof_off_h"][:ns + 1]
seq = rd0["seq"][:so[-1]].cpu().numpy()
prof = rd0["prof"][:po[-1]].cpu().numpy().view(np.uint16)
O = Oracle(K, 20000, hcov, dcov)
t0 = time.perf_counter()
want = O.classify_batch(seq, so, prof, po, nthreads=16)
tc = time.perf_counter() - t0
got = b0.labels[:so[-1]].cpu().numpy()
nbad = int((got != want).sum())
print("oracle: %d reads %d bases in %.2f s (%.1f Mbases/s, 16 threads); label mismatches %d" % (ns, so[-1], tc, so[-1] / tc / 1e6, nbad), flush=True)
if nbad:
    bad = np.nonzero(got != want)[0]
    r = np.searchsorted(so, bad[0], side="right") - 1
    print("first mismatch: read", r, "pos", bad[0] - so[r], "of", so[r + 1] - so[r])
# determinism / range independence: regenerate a middle range on its own
a, n = plan[0][0] + 100, 50
rd2 = ds.reads(a, n)
s0 = rd0["seq_off_h"]
same = torch.equal(rd2["seq"][:rd2["total_bases"]], rd0["seq"][s0[100]:s0[150]])
p0 = rd0["prof_off_h"]
same &= torch.equal(rd2["prof"][:rd2["total_kmers"]], rd0["prof"][p0[100]:p0[150]])
print("range regeneration identical:", bool(same), flush=True)
print("workspace GB: %.2f" % (clf.workspace_bytes() / 1e9))
