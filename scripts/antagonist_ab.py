#!/usr/bin/env python3
"""DIAGNOSTIC: the classification pipeline beside kernels that each load ONE shared resource (scripts/microbench/antagonist.hip).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ant -- python scripts/antagonist_ab.py run
    python scripts/antagonist_ab.py report gpurun_out/ant/*/*_kernel_trace.csv profiles/r05_antagonists.txt

`run`: a 1-Gbase sub-batch of the bench workload is classified alone and then beside 1 and 2 antagonist waves per SIMD of
every kind (the antagonists are launched first, on their own stream, and stopped after the pipeline has finished).
`report`: per product kernel, its duration alone and beside each antagonist (which antagonist ran is read off the trace).
"""
import ctypes as C
import collections
import csv
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KINDS = ["sleep", "salu", "valu32", "valu64", "lds", "mem", "regs"]


def run():
    import torch
    from classpro_amd.synth_dev import DeviceSynth
    from classpro_amd.api import Classifier, Batch, hist_covs
    so = os.path.join(ROOT, "build_diag", "libcp_antagonist.so")
    if not os.path.exists(so):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17",
                               os.path.join(ROOT, "scripts", "microbench", "antagonist.hip"), "-o", so])
    A = C.CDLL(so)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    assert A.ant_init() == 0
    ds = DeviceSynth(genome_len=25_000_000, cov=40, read_len=20000, K=40, seed=1, device=str(dev))
    low, high, il, ih, h = ds.hist
    hc, dc = hist_covs(h, low, high, il, ih, 0)
    clf = Classifier(K=40, read_len=20000, hcov=hc, dcov=dc, device=str(dev))
    b = Batch.from_device(ds.reads(0, ds.n_reads))
    sb = torch.cuda.Stream(dev)
    for _ in range(3):
        clf.run(b)
    torch.cuda.synchronize()
    print("batch: %d reads, %.2f Gbases" % (b.nreads, b.total_bases / 1e9), flush=True)
    out = []
    for kind in [-1] + list(range(len(KINDS))):
        for w in ((0,) if kind < 0 else (1, 2)):
            if kind >= 0:
                # every wave leaves on the stop flag or after at most about a second of turns
                iters = {0: 1 << 20, 6: 1 << 20, 5: 1 << 16}.get(kind, 1 << 20)
                assert A.ant_launch(kind, 1024 * w, iters, C.c_void_p(sb.cuda_stream)) == 0
                time.sleep(0.003)                               # the antagonists are resident before the pipeline starts
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(2):
                clf.run(b)
            e1.record()
            e1.synchronize()
            A.ant_stop()
            torch.cuda.synchronize()
            clf.check()
            ms = e0.elapsed_time(e1) / 2
            out.append((KINDS[kind] if kind >= 0 else "alone", w, ms))
            print("%-8s x%d  %.3f ms per sub-batch  (%.1f Gbases/s)" % (out[-1][0], w, ms, b.total_bases / ms / 1e6), flush=True)
    clf.close()


def report(trace, dest):
    rows = []
    for r in csv.DictReader(open(trace)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace(" ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, int(r.get("Grid_Size", 0) or r.get("Grid_Size_X", 0) or 0)))
    rows.sort()
    ants = [(s, e, n[6:] + ("x%d" % (g // 65536)), g) for s, e, n, g in rows if n.startswith("k_ant_")]
    prod = [(s, e, n) for s, e, n, g in rows if n.startswith("k_") and not n.startswith("k_ant_") and "_table" not in n and not n.startswith("k_sg_")]
    first_ant = ants[0][0] if ants else 1 << 62
    # the warm-up runs come before the first measured "alone" pair: keep the last 2 pipeline runs before the first antagonist
    scans = [s for s, e, n in prod if n == "k_scan_candidates" and s < first_ant]
    t_alone = scans[-2] if len(scans) >= 2 else 0
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for s, e, n in prod:
        tag = None
        if t_alone <= s < first_ant:
            tag = "alone"
        else:
            for a0, a1, an, g in ants:
                if a0 <= s and e <= a1:
                    tag = an
                    break
        if tag:
            acc[n][tag].append((e - s) / 1e3)
    cols = ["alone"] + [a[2] for a in ants]
    with open(dest, "w") as f:
        f.write("# rocprofv3 --kernel-trace -- python scripts/antagonist_ab.py run   (one 1-Gbase sub-batch, one stream, two runs per column)\n")
        f.write("# average duration in us of each product kernel alone and beside N antagonist waves per SIMD (x1 / x2) that load one resource\n")
        f.write("%-34s" % "kernel" + "".join("%10s" % c for c in cols) + "\n")
        order = sorted(acc, key=lambda k: -sum(acc[k].get("alone", [0])))
        for k in order:
            f.write("%-34s" % k[:34] + "".join("%10.0f" % (sum(acc[k][c]) / len(acc[k][c])) if acc[k].get(c) else "%10s" % "-" for c in cols) + "\n")
        tot = {c: sum(sum(acc[k][c]) / len(acc[k][c]) for k in acc if acc[k].get(c)) for c in cols}
        f.write("%-34s" % "sum" + "".join("%10.0f" % tot[c] for c in cols) + "\n")
    print(open(dest).read())


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2], sys.argv[3])
