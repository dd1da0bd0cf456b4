#!/usr/bin/env python3
"""End-to-end rate of the drop-in `ClassPro` binary (FASTA + FASTK files in, .class out) on files written to
tmpfs from the device synthesiser's reads.  Run on the GPU box.

    python scripts/cli_e2e.py [mbases=800] [threads=4,16,32] [devices=0] [gz=0]
"""
import os
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_inputs(d, rd_host, seq_off, hist, K=40, gz=False, root="reads"):
    """rd_host: (seq uint8, prof uint16) flat host arrays of the reads; FASTA + single-part FASTK files under d."""
    from classpro_amd.api import encode_profiles
    seq, prof = rd_host
    n = len(seq_off) - 1
    prof_off = seq_off - np.arange(n + 1) * (K - 1)
    with open(os.path.join(d, root + ".fasta"), "wb") as f:
        for i in range(n):
            f.write(b">read%d\n" % (i + 1))
            f.write(seq[seq_off[i]:seq_off[i + 1]].data)
            f.write(b"\n")
    if gz:
        subprocess.check_call(["gzip", "-1", "-f", os.path.join(d, root + ".fasta")])
    codes, off = encode_profiles([prof[prof_off[i]:prof_off[i + 1]] for i in range(n)])
    low, high, il, ih, h = hist
    with open(os.path.join(d, root + ".hist"), "wb") as f:
        f.write(struct.pack("<iii", K, low, high))
        f.write(struct.pack("<qq", il, ih))
        f.write(np.asarray(h, "<i8").tobytes())
    with open(os.path.join(d, root + ".prof"), "wb") as f:
        f.write(struct.pack("<ii", K, 1))
    with open(os.path.join(d, "." + root + ".pidx.1"), "wb") as f:
        f.write(struct.pack("<i", K))
        f.write(struct.pack("<qq", 0, n))
        f.write(off[1:].astype("<i8").tobytes())
    with open(os.path.join(d, "." + root + ".prof.1"), "wb") as f:
        f.write(codes.tobytes())
    return os.path.join(d, root + (".fasta.gz" if gz else ".fasta"))


def run_cli(path, threads, devices=None, reps=2):
    from classpro_amd import build
    cli = os.path.join(os.path.dirname(build.OUT), "ClassPro")
    env = dict(os.environ)
    if devices:
        env["CLASSPRO_DEVICES"] = devices
    best, lines = None, []
    out = os.path.join(os.path.dirname(path), os.path.basename(path).split(".")[0] + ".class")
    for _ in range(reps):
        if os.path.exists(out):
            os.unlink(out)                     # a fresh output file every time (freeing the old one's pages is not the run's work)
        t0 = time.time()
        r = subprocess.run([cli, "-v", "-T%d" % threads, path], capture_output=True, text=True, env=env)
        dt = time.time() - t0
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
        lines = [l.strip() for l in r.stderr.splitlines() if "Resources" in l or "host:" in l]
        best = dt if best is None else min(best, dt)
    return best, lines


if __name__ == "__main__":
    import shutil
    import torch
    from classpro_amd.synth_dev import DeviceSynth
    mb = float(sys.argv[1]) if len(sys.argv) > 1 else 800.0
    tlist = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 16, 32]
    threads = tlist[0]
    devices = sys.argv[3] if len(sys.argv) > 3 else None
    gz = len(sys.argv) > 4 and sys.argv[4] == "1"
    ds = DeviceSynth(genome_len=200_000_000, cov=40, read_len=20000, seed=1)
    n = int(np.searchsorted(ds.seq_off_all, mb * 1e6))
    a = 0
    rd = ds.reads(a, n)
    seq = rd["seq"][:rd["total_bases"]].cpu().numpy()
    prof = rd["prof"][:rd["total_kmers"]].cpu().numpy().view(np.uint16)
    so, hist, nb = rd["seq_off_h"], ds.hist, rd["total_bases"]
    del rd, ds
    torch.cuda.empty_cache()
    d = tempfile.mkdtemp(prefix="cp_e2e_", dir="/dev/shm")
    try:
        t0 = time.time()
        path = write_inputs(d, (seq, prof), so, hist, gz=gz)
        print("wrote %d reads, %.1f Mbases to %s in %.1f s" % (n, nb / 1e6, d, time.time() - t0), flush=True)
        for t in tlist:
            dt, lines = run_cli(path, t, devices)
            print("-T%d devices=%s: %.3f s wall, %.1f Mbases/s end to end" % (t, devices or "all", dt, nb / dt / 1e6), lines, flush=True)
        print("class file bytes", os.path.getsize(os.path.join(d, "reads.class")))
    finally:
        shutil.rmtree(d, ignore_errors=True)
