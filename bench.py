#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric, Mbases/s classified (k=40, 40x HiFi), on N MI355X, one process per GPU.

A "step" is one pass of the whole hot path (cp_classify_batch: candidate scan, find_wall,
find_rel_intvl, classify_rel, classify_unrel, label paint) over one batch of synthetic reads that is
already resident in HBM.  Reads shard trivially: every rank classifies its own reads, there is no
data-path collective (torch.distributed is used for the barrier and the max-over-ranks time only).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genome", type=int, default=5_000_000, help="synthetic diploid genome length (MHC-like: 5 Mbp)")
    ap.add_argument("--cov", type=int, default=40)
    ap.add_argument("--read-len", type=int, default=20000)
    ap.add_argument("--tile", type=int, default=1, help="replicate the read set this many times in the batch")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-sample-reads", type=int, default=1 << 30, help="reads of the batch timed on the CPU (default: all)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    return ap.parse_args()


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from classpro_amd import synth
    from classpro_amd.api import Classifier, Batch, hist_covs
    from classpro_amd._lib import lib, check
    import ctypes as C

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    # ---- synthetic workload (weak scaling: every rank gets its own read set of the same size) ----
    t0 = time.time()
    ds = synth.make_dataset(genome_len=a.genome, cov=a.cov, read_len=a.read_len, K=40, het=0.001,
                            n_repeats=max(3, a.genome // 80000), min_len=3000, seed=a.seed + rank)
    seq, seq_off, prof, prof_off = synth.pack_batch(ds["seqs"], ds["profiles"])
    low, high, il, ih, h = ds["hist"]
    hcov, dcov = hist_covs(h, low, high, il, ih, 0)
    if a.tile > 1:
        n = len(seq_off) - 1
        seq_off = np.concatenate([[0], np.cumsum(np.tile(np.diff(seq_off), a.tile))]).astype(np.int64)
        prof_off = np.concatenate([[0], np.cumsum(np.tile(np.diff(prof_off), a.tile))]).astype(np.int64)
        seq = np.tile(seq, a.tile)
        prof = np.tile(prof, a.tile)
    t_gen = time.time() - t0

    clf = Classifier(K=40, read_len=a.read_len, hcov=hcov, dcov=dcov, device=str(dev))
    t0 = time.time()
    b = Batch(seq, seq_off, prof, prof_off, device=str(dev))
    torch.cuda.synchronize()
    t_h2d = time.time() - t0
    L = lib()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def step():
        check(L.cp_classify_batch(clf.p, clf.ws, b.seq.data_ptr(), b.seq_off.data_ptr(), b.prof.data_ptr(),
                                  b.prof_off.data_ptr(), b.nreads, b.total_bases, b.total_kmers,
                                  b.labels.data_ptr(), stream))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    clf.check()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        nb = torch.tensor([b.total_bases], dtype=torch.int64, device=dev)
        dist.all_reduce(nb, op=dist.ReduceOp.SUM)
        total_bases_all = int(nb.item())
    else:
        total_bases_all = b.total_bases
    value = total_bases_all * a.steps / dt / 1e6

    out = None
    if rank == 0:
        # ---- roofline of the profile-scan kernel: HIP events on the launch stream ----------------
        nw = b.total_kmers // 64 + 2
        bm = torch.zeros(nw, dtype=torch.int64, device=dev)
        iters = 200                                     # ~16 ms of back-to-back launches: a stable average
        for _ in range(10):
            check(L.cp_scan_candidates(clf.p, b.prof.data_ptr(), b.total_kmers, bm.data_ptr(), stream))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
        for _ in range(iters):
            check(L.cp_scan_candidates(clf.p, b.prof.data_ptr(), b.total_kmers, bm.data_ptr(), stream))
        e1.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize()
        scan_ms = e0.elapsed_time(e1) / iters          # back-to-back launches on the stream the kernel runs on
        alg_bytes = 2.0 * b.total_kmers                 # 2 B (uint16 count) per position
        achieved = alg_bytes / (scan_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(_ROOT, "profiles", "scan_pmc.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                traffic = j["hbm_bytes_per_position"] * b.total_kmers
            except Exception:
                traffic = None
        roof = {"kernel": "k_scan_candidates", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "ms_per_launch": round(scan_ms, 4), "algorithmic_bytes_per_launch": alg_bytes}

        # ---- PCIe-inclusive rate (never `value`): H2D of inputs + step + D2H of labels -------------
        t0 = time.perf_counter()
        b2 = Batch(seq, seq_off, prof, prof_off, device=str(dev))
        lab = clf.classify(b2)
        t_e2e = time.perf_counter() - t0
        del b2
        # the drop-in's own transfer pattern: pinned bases + FASTK code strings in, decode on the device
        # (cp_decode_profiles), classify, pinned labels out
        from classpro_amd.api import encode_profiles
        prs = ds["profiles"] * a.tile
        codes, code_off = encode_profiles(prs)
        h_seq = torch.from_numpy(seq).pin_memory()
        h_code = torch.from_numpy(codes).pin_memory()
        h_lab = torch.empty(b.total_bases, dtype=torch.uint8).pin_memory()
        d_coff = torch.from_numpy(code_off).to(dev)
        d_prof = torch.empty_like(b.prof)
        torch.cuda.synchronize()
        t_codes = []
        for _ in range(3):
            t0 = time.perf_counter()
            d_seq = h_seq.to(dev, non_blocking=True)
            d_code = h_code.to(dev, non_blocking=True)
            check(L.cp_decode_profiles(clf.ws, d_code.data_ptr(), d_coff.data_ptr(), b.prof_off.data_ptr(), b.nreads,
                                       d_prof.data_ptr(), stream))
            check(L.cp_classify_batch(clf.p, clf.ws, d_seq.data_ptr(), b.seq_off.data_ptr(), d_prof.data_ptr(),
                                      b.prof_off.data_ptr(), b.nreads, b.total_bases, b.total_kmers,
                                      b.labels.data_ptr(), stream))
            h_lab.copy_(b.labels[:b.total_bases], non_blocking=True)
            torch.cuda.synchronize()
            t_codes.append(time.perf_counter() - t0)
        clf.check()
        decode_ok = bool(torch.equal(d_prof[:b.total_kmers], b.prof[:b.total_kmers])) and \
            bool(np.array_equal(h_lab.numpy(), lab))
        e0.record(torch.cuda.current_stream(dev))
        for _ in range(10):
            check(L.cp_decode_profiles(clf.ws, d_code.data_ptr(), d_coff.data_ptr(), b.prof_off.data_ptr(), b.nreads,
                                       d_prof.data_ptr(), stream))
        e1.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize()
        decode_ms = e0.elapsed_time(e1) / 10

        # ---- accuracy against the generator's ground truth (prof2class.c:210-229 mapping of the genomic
        #      multiplicity of every k-mer: 0 E, 1 H, 2 D, >=3 R); what class2acc reports as "Accuracy" ----
        tmap = np.full(32768, ord("R"), np.uint8)
        tmap[0], tmap[1], tmap[2] = ord("E"), ord("H"), ord("D")
        rel = np.concatenate(ds["rel_profiles"] * a.tile)
        kpos = np.ones(b.total_bases, bool)
        for k in range(39):
            kpos[seq_off[:-1] + k] = False
        est = lab[kpos]
        acc = float((est == tmap[rel]).mean()) if len(rel) == len(est) else None

        # ---- CPU baseline: the oracle (a port, pthreads) on a bounded sample of the same workload --
        cpu = None
        if not a.no_cpu:
            from oracle.oracle import Oracle
            ncpu = min(16, os.cpu_count() or 1)
            ns = min(a.cpu_sample_reads, len(ds["seqs"]))
            so, po = seq_off[:ns + 1], prof_off[:ns + 1]
            O = Oracle(40, a.read_len, hcov, dcov)
            O.classify_batch(seq[:so[min(64, ns)]], so[:min(64, ns) + 1], prof[:po[min(64, ns)]], po[:min(64, ns) + 1], nthreads=ncpu)
            t0 = time.perf_counter()
            want = O.classify_batch(seq[:so[-1]], so, prof[:po[-1]], po, nthreads=ncpu)
            tc = time.perf_counter() - t0
            cpu = {"value": round(int(so[-1]) / tc / 1e6, 2), "unit": "Mbases/s", "cores": ncpu, "kind": "port",
                   "sample": "first %d reads (%d bases) of the same batch, oracle/classpro_oracle.c with %d pthreads" % (ns, int(so[-1]), ncpu),
                   "seconds": round(tc, 2)}
            if not a.no_verify:
                nbad = int((lab[:so[-1]] != want).sum())
                cpu["label_mismatches_vs_hip"] = nbad
        out = {
            "metric": "Mbases/s classified (k=40, 40x HiFi) at 1/2/4/8 MI355X vs CPU -T16", "value": round(value, 2), "unit": "Mbases/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16/f64",
            "data": "synthetic",
            "config": {"workload": "MHC-like synthetic diploid %.1f Mbp, %dx HiFi, r=%d, k=40 (BASELINE configs[1]), x%d tile" % (
                           a.genome / 1e6, a.cov, a.read_len, a.tile),
                       "reads_per_gpu": b.nreads, "bases_per_gpu": b.total_bases, "hcov": hcov, "dcov": dcov,
                       "parallelism": "read-sharded x%d, no collective" % world},
            "roofline": roof, "cpu_baseline": cpu,
            "extras": {"whole_step_algorithmic_gb_per_s": round((2.0 * b.total_kmers + 2.0 * b.total_bases) / (dt / a.steps) / 1e9, 1),
                       "pcie_inclusive_mbases_per_s": round(b.total_bases / t_e2e / 1e6, 2),
                       "pcie_inclusive_pinned_codes_mbases_per_s": round(b.total_bases / min(t_codes) / 1e6, 2),
                       "accuracy_vs_synthetic_truth": None if acc is None else round(acc, 5),
                       "code_bytes_per_base": round(len(codes) / b.total_bases, 4),
                       "decode_ms": round(decode_ms, 3), "decode_matches": decode_ok,
                       "h2d_seconds": round(t_h2d, 3), "gen_seconds": round(t_gen, 1),
                       "workspace_gb": round(clf.workspace_bytes() / 1e9, 2)},
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    clf.close()


if __name__ == "__main__":
    main()
