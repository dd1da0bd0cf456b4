#!/usr/bin/env python3
"""Diagnostic: end-to-end rate of the drop-in `ClassPro` binary (FASTA + FASTK files in, .class out) on the
bench data set, written to a scratch directory.  Run on the GPU box."""
import os, struct, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from classpro_amd import synth, build
from classpro_amd.api import encode_profiles

genome = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
ds = synth.make_dataset(genome_len=genome, cov=40, read_len=20000, K=40, het=0.001, n_repeats=max(3, genome // 80000), min_len=3000, seed=1)
d = tempfile.mkdtemp(prefix="cp_e2e_", dir="/tmp")
with open(os.path.join(d, "reads.fasta"), "wb") as f:
    for n, s in zip(ds["names"], ds["seqs"]):
        f.write(b">" + n.encode() + b"\n" + s + b"\n")
codes, off = encode_profiles(ds["profiles"])
low, high, il, ih, h = ds["hist"]
with open(os.path.join(d, "reads.hist"), "wb") as f:
    f.write(struct.pack("<iii", 40, low, high)); f.write(struct.pack("<qq", il, ih)); f.write(np.asarray(h, "<i8").tobytes())
with open(os.path.join(d, "reads.prof"), "wb") as f:
    f.write(struct.pack("<ii", 40, 1))
with open(os.path.join(d, ".reads.pidx.1"), "wb") as f:
    f.write(struct.pack("<i", 40)); f.write(struct.pack("<qq", 0, len(ds["seqs"]))); f.write(off[1:].astype("<i8").tobytes())
with open(os.path.join(d, ".reads.prof.1"), "wb") as f:
    f.write(codes.tobytes())
cli = os.path.join(os.path.dirname(build.OUT), "ClassPro")
for rep in range(2):
    t0 = time.time()
    r = subprocess.run([cli, "-v", "-T16", os.path.join(d, "reads.fasta")], capture_output=True, text=True)
    dt = time.time() - t0
    print("run %d: exit %d, %.2f s wall;" % (rep, r.returncode, dt), [l for l in r.stderr.splitlines() if "Resources" in l or "host:" in l])
print("bases", sum(len(s) for s in ds["seqs"]), "class file bytes", os.path.getsize(os.path.join(d, "reads.class")))
