#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric, Mbases/s classified (k=40, 40x HiFi), on N MI355X, one process per GPU.

Workload (N=1 default): BASELINE configs[2], a synthetic 200 Mbp diploid genome at 40x HiFi, r=20000: 400 000
distinct seeded reads, 8.0 Gbases, 16 GB of uint16 count profiles, generated on the device
(classpro_amd/synth_dev.py) and RESIDENT IN HBM before the timed region.  A "step" is one pass of the whole hot
path (cp_classify_batch: candidate scan, find_wall, find_rel_intvl, classify_rel, classify_unrel, label paint)
over the whole resident read set, issued as sub-batches of --batch-mbases (the size that fills the machine).

N > 1 (strong scaling) defaults to BASELINE configs[3]: a synthetic 3 Gbp diploid genome at 40x, 6 000 000 reads,
120 Gbases, sharded over the ranks as contiguous read ranges balanced by bases (classpro_amd.shard.plan_shards);
every rank generates and classifies only its own range; there is no data-path collective (torch.distributed =
barrier + max-over-ranks time only).  A rank's share (15 Gbases = 60 GB of bases, counts and labels at N = 8) is
resident in HBM when it fits; a share that does not fit (N = 1, 2: 120 / 60 Gbases) is taken in resident WINDOWS of
at most --window-gbases: a window is generated (untimed), then the W warm-up and the K timed passes run over it
between barriers, then the next window; a step's time is the sum over the windows, so every step still classifies
every base of the share exactly once and the inputs of every timed region are resident when it starts.
`--shard r/N` runs rank r's share of an N-rank run in ONE process on one GPU (no process group): the way one shard
of configs[3] is measured on a single-GPU box.  `--scaling weak` gives every rank its own full-size set instead.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--genome 3e9 --shard r/N]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` without a launcher starts the N ranks itself (before any GPU call).
Prints ONE JSON line on rank 0 (fields: README / DESIGN.md section 6).
"""
import argparse
import hashlib
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec)
K = 40


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--genome", type=float, default=None,
                    help="synthetic diploid genome length; default 200e6 (configs[2]) for one GPU, 3e9 (configs[3]) for --gpus N > 1 or --shard")
    ap.add_argument("--shard", default=None, metavar="r/N",
                    help="classify only shard r of the N contiguous shards of the read set, in this one process (no process group)")
    ap.add_argument("--window-gbases", type=float, default=32.0,
                    help="a rank's share is resident in HBM in windows of at most this many Gbases (4 B of HBM per base + workspace)")
    ap.add_argument("--cov", type=int, default=40)
    ap.add_argument("--read-len", type=int, default=20000)
    ap.add_argument("--batch-mbases", type=float, default=4100.0,
                    help="sub-batch size of one cp_classify_batch call: 4-Gbase sub-batches (one per stream and step on the 8-Gbase set) 204.0-204.3 "
                         "Gbases/s, 2-Gbase ones 201.8-202.5, 1.4-Gbase ones 199.8-200.4 (A/B in one call, round 5), for 89 / 44 / 30 GB of workspace -- of 288")
    ap.add_argument("--streams", type=int, default=2, help="sub-batches alternate over this many streams / workspaces")
    ap.add_argument("--issue-threads", type=int, default=1, help="host threads that issue the sub-batches: 1 = one thread alternating over the streams "
                                                                  "(the default: 204.5-205.6 Gbases/s); 2 = one thread per stream, free-running or staggered by "
                                                                  "half a sub-batch (199.3-200.2: measured and dropped, round 5)")
    ap.add_argument("--no-stagger", dest="stagger", action="store_false", help="with a thread per stream: do not start stream k half a sub-batch late")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="target CPU time of each cpu_baseline leg (-T1, -T16, -T<all>)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the barrier / max-over-ranks reduction; gloo = rehearsal of N ranks on a box with "
                         "fewer GPUs (ranks beyond the device count share device 0)")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the process group even for one rank (--gpus 1 --backend nccl --force-dist runs the RCCL barrier and all_reduce on one device)")
    ap.add_argument("--pcie-gbases", type=float, default=2.0,
                    help="extras.pcie: this many Gbases of the resident set are staged in pinned host memory (bases as characters, FASTK codes) and streamed "
                         "through the three-slot pipeline, over and over, for at least --pcie-seconds; 8 = all of configs[2]")
    ap.add_argument("--pcie-seconds", type=float, default=1.5)
    ap.add_argument("--pcie-batch-mbases", type=float, default=250.0)
    ap.add_argument("--pcie-slots", type=int, default=6)
    ap.add_argument("--pcie-pack-threads", type=int, default=8, help="host threads packing a batch's bases to 2 bits (per slot, inside the timed region)")
    ap.add_argument("--host-feed-gbases", type=float, default=1.0, help="source reads per feeder group of extras.host_feed")
    ap.add_argument("--host-feed-seconds", type=float, default=1.5, help="span of each extras.host_feed leg")
    ap.add_argument("--no-rank-pcie", action="store_true", help="N > 1: skip the PCIe-inclusive leg that all ranks run at once")
    ap.add_argument("--only-pcie", action="store_true", help="skip the other extras (profiling the PCIe pipeline)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-cpu-ref", action="store_true", help="cpu_baseline from the port only (skip the reference-function legs)")
    ap.add_argument("--cpu-ref-max-threads", type=int, default=32,
                    help="cap on the thread count of the reference's -T<all> leg: every thread does 240 000 mallocs of 60 KB (alloc_rel_arg for "
                         "MAX_READ_LEN, ~1 GB touched) and the threads serialise on the address-space lock: 0.3 s at -T1, 9.6 s at -T16, 85 s at -T64 (measured)")
    ap.add_argument("--no-extras", action="store_true", help="skip the MHC-like / PCIe / CLI extras (profiling runs)")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` with no launcher: start the N ranks here, before anything touches the GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def scan_kernel_id():
    """Hash of the scan kernel's source text: a PMC traffic figure is only quoted for the kernel it was measured on."""
    src = open(os.path.join(_ROOT, "classpro_amd", "csrc", "kernels.hip")).read()
    m = re.search(r"typedef unsigned cp_u4v.*?\n// -{20,}\n//  Candidate count per read", src, re.S)
    return hashlib.sha1((m.group(0) if m else src).encode()).hexdigest()[:12]


def tables_estimate():
    """Device bytes of the look-up tables cp_params_create will ask for (capi.hip: the Skellam table, its exp() twin unless
    CLASSPRO_EXP_TABLE=0, 8 MB + 66 MB of small ones), from the same environment knobs -- for the pre-flight check, which runs
    before a cp_params exists; cp_params_tables() reports the real figures afterwards (extras.tables_bytes)."""
    if os.environ.get("CLASSPRO_TABLES", "1") == "0":
        return 2e6
    mb = int(os.environ.get("CLASSPRO_SKELLAM_TABLE_MB", "1024"))
    twin = os.environ.get("CLASSPRO_EXP_TABLE", "1") != "0"
    return mb * 2.0 ** 20 * (2 if twin else 1) + 80e6


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "0"))
    if world == 0:
        if a.gpus > 1:
            self_launch(a)
        world = 1
    if world != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (a.gpus, world))
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist
    import ctypes as C
    from classpro_amd.synth_dev import DeviceSynth
    from classpro_amd.api import Classifier, Batch, hist_covs
    from classpro_amd.shard import plan_shards
    from classpro_amd._lib import lib, check

    use_dist = world > 1 or a.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:  # --force-dist without a launcher: a rendezvous of one
            s_ = socket.socket()
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
            s_.close()
        dist.init_process_group(a.backend, rank=rank, world_size=world)
    if a.backend == "gloo" and local >= torch.cuda.device_count():
        local = 0                                           # rehearsal: more ranks than GPUs
    rdev = None if a.backend == "nccl" else "cpu"           # where the two scalars of the reduction live
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    L = lib()

    # ---- synthetic workload, generated in HBM -------------------------------------------------------------
    t0 = time.time()
    shard_r, shard_n = (rank, world)
    if a.shard:
        if world != 1:
            sys.stderr.write("bench.py: --shard runs in one process (--gpus 1)\n")
            sys.exit(2)
        shard_r, shard_n = (int(x) for x in a.shard.split("/"))
        if not 0 <= shard_r < shard_n:
            sys.stderr.write("bench.py: --shard r/N needs 0 <= r < N\n")
            sys.exit(2)
    if a.genome is None:
        a.genome = 3e9 if (world > 1 or a.shard) and a.scaling == "strong" else 200e6
    G = int(a.genome)
    cfg = "BASELINE configs[3]" if G >= 3_000_000_000 else "BASELINE configs[2]" if G == 200_000_000 else "configs[2]-like"
    seed = a.seed + (rank if a.scaling == "weak" else 0)
    ds = DeviceSynth(genome_len=G, cov=a.cov, read_len=a.read_len, K=K, seed=seed, device=str(dev))
    torch.cuda.synchronize()
    t_setup = time.time() - t0
    low, high, il, ih, h = ds.hist
    hcov, dcov = hist_covs(h, low, high, il, ih, 0)
    if a.scaling == "strong":
        bounds = plan_shards(ds.seq_off_all, shard_n)
        first, last = bounds[shard_r], bounds[shard_r + 1]
    else:
        first, last = 0, ds.n_reads
    share = int(ds.seq_off_all[last] - ds.seq_off_all[first])
    nst = max(1, a.streams)
    # resident windows of the share (one when it fits), every rank the same number of them (the barriers pair up)
    nwin = max(1, -(-share // int(a.window_gbases * 1e9)))
    if use_dist:
        t = torch.tensor([nwin], dtype=torch.int64, device=rdev or dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        nwin = int(t.item())
    windows = ds.plan_batches(-(-share // nwin) + 1, first, last)
    while len(windows) < nwin:
        windows.append((last, 0))

    # pre-flight: a window's tensors (4 B per base: bases, 2-byte counts, labels; + offsets), the workspaces (about 11 B
    # per base of a sub-batch each, measured: DESIGN section 2) and the tables must fit what the device has free NOW --
    # two ranks sharing one GPU, or a window sized for another card, end here with a message instead of a raw hipMalloc
    # failure in the middle of the run
    win_max = max((int(ds.seq_off_all[f + c] - ds.seq_off_all[f]) for f, c in windows if c), default=0)
    need = 4.05 * win_max + nst * 11.5 * min(win_max, a.batch_mbases * 1e6) + tables_estimate()
    torch.cuda.empty_cache()
    free_b, total_b = torch.cuda.mem_get_info(dev)
    sharers = 1
    if use_dist:                                            # ranks that sit on the same card (a rehearsal) share what is free
        pr = torch.cuda.get_device_properties(dev)
        me = "%s/%04x:%02x:%02x" % (socket.gethostname(), pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        ids = [None] * world
        dist.all_gather_object(ids, me)
        sharers = ids.count(me)
        fr = torch.tensor([free_b], dtype=torch.int64, device=rdev or dev)
        dist.all_reduce(fr, op=dist.ReduceOp.MIN)           # (the ranks look at different moments: take the lowest figure)
        free_b = int(fr.item()) if sharers > 1 else free_b
    need *= sharers
    short = need > free_b
    if use_dist:                                            # leave together: a rank that went on alone would wait at the next barrier for ever
        fl = torch.tensor([1 if short else 0], dtype=torch.int64, device=rdev or dev)
        dist.all_reduce(fl, op=dist.ReduceOp.MAX)
        if int(fl.item()) and not short:
            sys.stderr.write("bench.py: rank %d: another rank does not have the device memory it needs; leaving with it\n" % rank)
            dist.destroy_process_group()
            sys.exit(2)
    if short:
        sys.stderr.write("bench.py: rank %d: %d rank(s) on device %d need about %.0f GB of device memory (a %.1f-Gbase resident window + %d workspaces + "
                         "tables each), but the device has %.0f GB free of %.0f GB: lower --window-gbases (now %.0f), or give every rank its own GPU\n"
                         % (rank, sharers, local, need / 1e9, win_max / 1e9, nst, free_b / 1e9, total_b / 1e9, a.window_gbases))
        if use_dist:
            dist.destroy_process_group()
        sys.exit(2)
    clf = Classifier(K=K, read_len=a.read_len, hcov=hcov, dcov=dcov, device=str(dev))
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(nst - 1)]
    wss = [clf.ws]
    for _ in range(nst - 1):
        w = C.c_void_p()
        check(L.cp_workspace_create(C.byref(w)))
        wss.append(w)

    def make_batches(w_first, w_count, truth):
        """sub-batches of at most --batch-mbases, their number a multiple of the stream count (so that consecutive
        sub-batches, also across steps, alternate over the streams); a window is never a single sub-batch"""
        if w_count == 0:
            return []
        wb = int(ds.seq_off_all[w_first + w_count] - ds.seq_off_all[w_first])
        nb_ = max(nst, -(-wb // int(a.batch_mbases * 1e6)))
        nb_ = -(-nb_ // nst) * nst
        out = []
        for i, (r0, n) in enumerate(ds.plan_batches(-(-wb // nb_) + 1, w_first, w_first + w_count)):
            rd = ds.reads(r0, n, truth=(truth and i == 0))
            out.append((rd, Batch.from_device(rd)))
        torch.cuda.synchronize()
        ds.check()
        return out

    def call(k, b):
        check(L.cp_classify_batch(clf.p, wss[k], b.seq.data_ptr(), b.seq_off.data_ptr(), b.prof.data_ptr(),
                                  b.prof_off.data_ptr(), b.nreads, b.total_bases, b.total_kmers,
                                  b.labels.data_ptr(), C.c_void_p(streams[k].cuda_stream)))

    def step(batches):
        for i, (_, b) in enumerate(batches):
            call(i % nst, b)

    def run_steps(batches, nsteps, stagger=0.0):
        """nsteps passes over the window's sub-batches.  --issue-threads 1 (default): one host thread alternates over the streams;
        every cp_classify_batch call waits for its sub-batch's head, so the streams end up starting their sub-batches within a
        head's time of each other and run the same kinds of kernel side by side.  --issue-threads 2: a host thread per stream
        issues that stream's sub-batches back to back, stream k starting k * `stagger` seconds late (half a sub-batch: one
        stream's bandwidth-bound scan and paint beside the other's latency-bound kernels).  That was the idea; measured, it
        loses 2.5 % with or without the stagger (A/B in one call), so it is an option, not the default."""
        if a.issue_threads <= 1 or nst == 1:
            for _ in range(nsteps):
                step(batches)
            return
        import threading
        errs = []

        def worker(k):
            try:
                torch.cuda.set_device(dev)
                if k and stagger:
                    time.sleep(k * stagger)
                for _ in range(nsteps):
                    for i, (_, b) in enumerate(batches):
                        if i % nst == k:
                            call(k, b)
            except Exception as e:      # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=worker, args=(k,)) for k in range(nst)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    dt, t_gen, my_bases, my_kmers, nsub = 0.0, 0.0, 0, 0, 0
    batches = []
    for wi, (w_first, w_count) in enumerate(windows):
        del batches
        torch.cuda.empty_cache()                            # the previous window's 4 B per base go back before the next is generated
        tg = time.time()
        batches = make_batches(w_first, w_count, truth=(rank == 0))
        t_gen += time.time() - tg
        my_bases += sum(b.total_bases for _, b in batches)
        my_kmers += sum(b.total_kmers for _, b in batches)
        nsub += len(batches)
        run_steps(batches, a.warmup)
        barrier()
        stagger = 0.0
        if a.stagger and a.issue_threads > 1 and nst > 1 and a.warmup > 0:
            tw = time.perf_counter()                        # one more untimed step (the first ones grow the workspaces): half the time
            run_steps(batches, 1)                           # a stream takes for one of its sub-batches
            barrier()
            stagger = 0.5 * (time.perf_counter() - tw) / max(1, len(batches) // nst)
        t0 = time.perf_counter()
        run_steps(batches, a.steps, stagger)
        barrier()
        dt += time.perf_counter() - t0
        for w in wss:                                       # the error words are sticky: every launch since the last check is covered
            check(L.cp_workspace_check(w))
    # (the last window stays resident: the roofline loop, the CPU legs and the extras below run on it)
    win_bases = sum(b.total_bases for _, b in batches)
    win_kmers = sum(b.total_kmers for _, b in batches)
    my_dt = dt
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=rdev or dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        nb = torch.tensor([my_bases], dtype=torch.int64, device=rdev or dev)
        dist.all_reduce(nb, op=dist.ReduceOp.SUM)
        total_bases_all = int(nb.item())
    else:
        total_bases_all = my_bases
    value = total_bases_all * a.steps / dt / 1e6

    # ---- N > 1: every rank runs a short PCIe-inclusive leg AT THE SAME TIME (its own GPU, its own feeders, one shared
    #      host): `value` above is kernels on resident data and scales by construction; this is where 8 feeders contending
    #      for the host's cores and DRAM would show.  Reported per rank in extras.pcie_all_ranks, never as `value`. ----
    pcie_ranks = None
    if world > 1 and not a.no_extras and not a.no_rank_pcie:
        import copy
        a2 = copy.copy(a)
        a2.pcie_gbases, a2.pcie_seconds = min(a.pcie_gbases, 1.0), min(a.pcie_seconds, 1.0)
        barrier()
        try:
            r_ = pcie_pipeline(a2, ds, clf, batches, dev)
            mine = {"rank": rank, "mbases_per_s": r_.get("mbases_per_s"), "runs_mbases_per_s": r_.get("label_runs", {}).get("mbases_per_s"),
                    "frac": r_.get("frac"), "pinned_h2d_peak_gb_per_s": r_.get("pinned_h2d_peak_gb_per_s"),
                    "gpu_numa_node": r_.get("gpu_numa_node"), "feeder_cpus_on_gpu_node": r_.get("feeder_cpus_on_gpu_node"),
                    "pack_threads_per_slot": r_.get("pack_threads_per_slot"), "labels_match_resident_run": r_.get("labels_match_resident_run")}
        except Exception as e:                              # never takes the bench line down
            mine = {"rank": rank, "error": repr(e)[:200]}
        barrier()
        pcie_ranks = [None] * world
        dist.all_gather_object(pcie_ranks, mine)

    if rank == 0:
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        # ---- roofline of the profile-scan kernel: HIP events on the launch stream; every launch streams one
        #      sub-batch's profile (1.6 GB at the default size), the whole resident set (16 GB >> the 256 MiB
        #      Infinity Cache) in rotation ----------------------------------------------------------------------
        # (the library scans a batch in kernel launches of at most 2^31 positions -- capi.hip: launch_scan -- and so does this
        #  loop: one cp_scan_candidates call per such piece, i.e. per kernel launch)
        chunk = 1 << 31
        if os.environ.get("CLASSPRO_SCAN_CHUNK_KMERS"):
            chunk = max(4096, int(os.environ["CLASSPRO_SCAN_CHUNK_KMERS"]) & ~4095)
        pieces = []
        for _, b in batches:
            for p0 in range(0, max(b.total_kmers, 1), chunk):
                n = min(chunk, b.total_kmers - p0)
                pieces.append((b.prof.data_ptr() + 2 * p0, n, torch.zeros(n // 64 + 2, dtype=torch.int64, device=dev)))
        reps = max(1, 200 // len(pieces))

        def scan_all():
            for ptr, n, bm in pieces:
                check(L.cp_scan_candidates(clf.p, ptr, n, bm.data_ptr(), stream))
        scan_all()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
        for _ in range(reps):
            scan_all()
        e1.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize()
        nlaunch = reps * len(pieces)
        scan_ms = e0.elapsed_time(e1) / nlaunch
        alg_bytes = 2.0 * win_kmers / len(pieces)       # 2 B (uint16 count) per position; average launch
        achieved = alg_bytes / (scan_ms * 1e-3) / 1e9
        del pieces
        traffic, traffic_src = None, None
        pmc = os.path.join(_ROOT, "profiles", "scan_pmc.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                # quoted only for the kernel and launch size it was measured on (rocprofv3 --pmc pass of this command)
                if j.get("kernel_id") == scan_kernel_id() and abs(j["positions_per_launch"] - alg_bytes / 2) < 0.02 * alg_bytes / 2:
                    traffic = j["hbm_bytes_per_position"] * alg_bytes / 2
                    traffic_src = j.get("source")
            except Exception:
                traffic = None
        roof = {"kernel": "k_scan_candidates", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "ms_per_launch": round(scan_ms, 4), "algorithmic_bytes_per_launch": alg_bytes, "launches_timed": nlaunch,
                "working_set_bytes": 2.0 * win_kmers}

        rd0, b0 = batches[0]
        extras = {"pcie_all_ranks": None if pcie_ranks is None else
                  {"what": "PCIe-inclusive pipeline (host packing + H2D + format kernels + classification + 2-bit labels D2H), all ranks at once, 1 Gbase staged per rank",
                   "sum_mbases_per_s": round(sum((x.get("mbases_per_s") or 0) for x in pcie_ranks), 1), "ranks": pcie_ranks},
                  "per_gpu_mbases_per_s": round(total_bases_all * a.steps / dt / 1e6 / world, 1),
                  "rank0_mbases_per_s_own_clock": round(my_bases * a.steps / my_dt / 1e6, 1),
                  "process_group": ("%s, world %d" % (a.backend, world)) if use_dist else None,
                  "whole_step_algorithmic_gb_per_s": round((2.0 * my_kmers + 2.0 * my_bases) * world / (dt / a.steps) / 1e9, 1),
                  "synth_setup_seconds": round(t_setup, 2), "gen_seconds": round(t_gen, 2), "sub_batches_per_rank": nsub,
                  "resident_windows_per_rank": len(windows), "window_gbases": round(win_bases / 1e9, 2), "streams": nst,
                  "tables_bytes": clf.tables(), "hbm_peak_allocated_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),
                  "workspace_gb": round(sum(int(L.cp_workspace_bytes(w)) for w in wss) / 1e9, 2),
                  "synth_err_kmer_fraction": round(ds.n_err_kmers / max(1, ds.total_bases), 4)}

        # ---- accuracy against the generator's ground truth (prof2class.c:210-229: multiplicity 0 E, 1 H, 2 D, >=3 R) ----
        tmap = torch.full((256,), ord("R"), dtype=torch.uint8, device=dev)
        tmap[0], tmap[1], tmap[2] = ord("E"), ord("H"), ord("D")
        # (on the first reads of the sub-batch, up to a gigabase: torch's masked indexing counts in 32 bits)
        na = int(np.searchsorted(rd0["seq_off_h"], min(b0.total_bases, 1_000_000_000), side="right") - 1)
        nb_a, nk_a = int(rd0["seq_off_h"][na]), int(rd0["prof_off_h"][na])
        kpos = torch.ones(nb_a, dtype=torch.bool, device=dev)
        for k in range(K - 1):
            kpos[b0.seq_off[:na] + k] = False
        extras["accuracy_vs_synthetic_truth"] = round(float((b0.labels[:nb_a][kpos] == tmap[rd0["truth"][:nk_a].long()]).float().mean()), 5)
        del kpos

        # ---- CPU baseline: the oracle (a port, pthreads) on bounded samples of the same workload; every label
        #      of the samples is also compared with the HIP result --------------------------------------------
        cpu = None
        if not a.no_cpu and world == 1:                 # (the contract: on rank 0 at N = 1 only)
            cpu = cpu_baseline(a, ds, batches, hcov, dcov)
        if not a.no_extras and world == 1 and not a.shard and G <= 200_000_000:
            extras.update(extra_rates(a, ds, clf, batches, hcov, dcov, dev, stream))

        out = {
            "metric": "Mbases/s classified (k=40, 40x HiFi) at 1/2/4/8 MI355X vs CPU -T16", "value": round(value, 2), "unit": "Mbases/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "u16/f64",
            "data": "synthetic",
            "timed_region": "classification kernels only (cp_classify_batch per sub-batch): bases, uint16 count profiles and labels are "
                            "resident in HBM; H2D, D2H, profile decode, file I/O and the host are OUTSIDE the timed region.  The other rates "
                            "of the same run: extras.pcie.mbases_per_s (host packing + PCIe + format kernels + kernels + labels back), "
                            "extras.cli_end_to_end_mbases_per_s (the ClassPro binary on files, process start to exit), extras.host_feed (the host side alone)",
            "config": {"workload": "synthetic %.0f Mbp diploid, %dx HiFi, r=%d, k=40 (%s): %d distinct reads, %.2f Gbases%s; a rank's share resident in HBM "
                                   "%s, sub-batches of at most %.0f Mbases; value = kernels on HBM-resident inputs (see timed_region)"
                                   % (G / 1e6, a.cov, a.read_len, cfg, ds.n_reads, ds.total_bases / 1e9,
                                      (", of which shard %d of %d (%.2f Gbases) in this process" % (shard_r, shard_n, my_bases / 1e9)) if a.shard else "",
                                      "whole" if len(windows) == 1 else "in %d windows (each generated untimed, then warm-up + timed steps over it; a step = the sum over windows)" % len(windows),
                                      a.batch_mbases),
                       "reads_total": ds.n_reads if a.scaling == "strong" else ds.n_reads * world, "bases_total": total_bases_all,
                       "reads_rank0": last - first, "bases_rank0": my_bases, "hcov": hcov, "dcov": dcov,
                       "shard": a.shard,
                       "parallelism": "contiguous read ranges balanced by bases x%d (%s), no collective" % (shard_n if a.shard else world, a.scaling)},
            "roofline": roof, "cpu_baseline": cpu, "extras": extras,
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    for w in wss[1:]:
        L.cp_workspace_destroy(w)
    clf.close()


def cpu_baseline(a, ds, batches, hcov, dcov):
    """The CPU beside the GPU number, on bounded samples of the same resident read set, every label compared with the
    HIP result.  Two implementations, -T1 / -T16 / -T<all> each (about --cpu-seconds of work per leg):
      * "reference": the reference's OWN per-read functions -- calc_seq_context (context.c), find_wall + find_rel_intvl
        (the GSL-free part of wall.c), classify_rel, classify_unrel, the paint loop -- in the reference's thread loop
        without its file I/O (oracle/ref_driver.c: ref_classify_batch = ClassPro.c:114-143 scratch once per thread,
        146-271 per read).  oracle/_ref/libclasspro_ref.so is built in the build container and travels with the snapshot.
        The allocation phase (alloc_rel_arg: 240 000 mallocs of 60 KB per thread, SURVEY hazard 7) is timed apart.
      * "port": the oracle's restatement (oracle/classpro_oracle.c).
    `value` is the reference's -T16 when that library is present, else the port's."""
    from oracle.oracle import Oracle, Ref, ref_wall_available
    O = Oracle(K, a.read_len, hcov, dcov)
    ncores_seen = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = cpu_quota()                                     # a box may show 256 cores and grant 16 cores' worth of time (cgroup cpu.max):
    ncores = ncores_seen if not quota else max(1, min(ncores_seen, int(np.ceil(quota))))     # -T<all> means all it can run at once
    rd0, b0 = batches[0]

    def host_sample(rd, b, nreads):
        so, po = rd["seq_off_h"][:nreads + 1], rd["prof_off_h"][:nreads + 1]
        return (rd["seq"][:so[-1]].cpu().numpy(), so, rd["prof"][:po[-1]].cpu().numpy().view(np.uint16), po,
                b.labels[:so[-1]].cpu().numpy())

    def run(sample, nt):
        seq, so, prof, po, lab = sample
        t0 = time.perf_counter()
        want = O.classify_batch(seq, so, prof, po, nthreads=nt)
        tc = time.perf_counter() - t0
        return int(so[-1]) / tc / 1e6, tc, int((lab != want).sum()), int(so[-1])

    # calibrate on 64 reads, then size each leg for ~cpu_seconds
    cal = host_sample(rd0, b0, min(64, b0.nreads))
    r1, _, _, _ = run(cal, 1)
    thread_legs = (("T1", 1), ("T16", min(16, ncores)), ("Tall", ncores))
    legs = {}
    mism = 0
    for name, nt in thread_legs:
        if name == "Tall" and nt == min(16, ncores):
            legs[name] = dict(legs["T16"], threads=nt)
            continue
        want_bases = r1 * 1e6 * a.cpu_seconds * min(nt, 16) * 0.8       # -T<all> takes the -T16 sample
        parts, got = [], 0
        for rd, b in batches:
            if got >= want_bases:
                break
            n = int(np.searchsorted(rd["seq_off_h"], min(want_bases - got, rd["seq_off_h"][-1]), side="left"))
            n = min(max(n, 1), b.nreads)
            parts.append(host_sample(rd, b, n))
            got += int(rd["seq_off_h"][n])
        rate_b, secs, bad = 0, 0.0, 0
        for s in parts:
            _, tc, nb, nbases = run(s, nt)
            rate_b += nbases
            secs += tc
            bad += nb
        legs[name] = {"mbases_per_s": round(rate_b / secs / 1e6, 2), "threads": nt, "seconds": round(secs, 2), "bases": rate_b,
                      "label_mismatches_vs_hip": bad}
        mism += bad
    t16 = legs["T16"]
    port = {"value": t16["mbases_per_s"], "unit": "Mbases/s", "cores": t16["threads"], "kind": "port",
            "sample": "first %d bases of the same resident read set (%.1f s of CPU work), oracle/classpro_oracle.c with %d pthreads"
                      % (t16["bases"], t16["seconds"], t16["threads"]),
            "seconds": t16["seconds"], "label_mismatches_vs_hip": mism, "legs": legs, "host_cores": ncores,
            "host_cores_visible": ncores_seen, "cgroup_cpu_quota": quota}
    if a.no_cpu_ref or not ref_wall_available():
        port["sample"] += "; oracle/_ref (the reference's own functions) is not present on this box"
        return port
    if mism:
        # The reference exit(1)s on a read whose E-interval list overflows, and that would take this process and its JSON line
        # with it.  The port returns instead (the read's labels stay unwritten = a mismatch), and its samples contain the
        # reference's: any mismatch above means the reference legs are not run.
        port["sample"] += "; reference-function legs skipped: the port's labels differ from the HIP result on this sample"
        return port

    # ---- the reference's own functions.  (No read of the resident set makes the reference exit(1): the HIP path raises
    #      CP_EOVERFLOW at exactly its five abort sites -- tests/test_gpu_reference.py -- and cp_workspace_check above
    #      reported none.)  One sample from the first sub-batch per leg: every call pays the allocation phase again. ----
    R = Ref(a.read_len, hcov, dcov).wall_setup_from(O)
    rlegs, rmism = {}, 0
    for name, nt in (("T1", 1), ("T16", min(16, ncores)), ("Tall", min(ncores, a.cpu_ref_max_threads))):
        if name == "Tall" and nt == min(16, ncores):
            rlegs[name] = dict(rlegs["T16"], threads=nt)
            continue
        want_bases = r1 * 1e6 * a.cpu_seconds * min(nt, 16) * 0.8
        n = int(np.searchsorted(rd0["seq_off_h"], min(want_bases, rd0["seq_off_h"][-1]), side="left"))
        n = min(max(n, nt), b0.nreads)
        seq, so, prof, po, lab = host_sample(rd0, b0, n)
        want, (t_alloc, t_run) = R.classify_batch(seq, so, prof, po, K, nthreads=nt, rlen_max=0, defined=True)
        bad = int((lab != want).sum())
        rlegs[name] = {"mbases_per_s": round(int(so[-1]) / t_run / 1e6, 2), "threads": nt, "seconds": round(t_run, 2),
                       "startup_seconds": round(t_alloc, 2), "bases": int(so[-1]), "label_mismatches_vs_hip": bad}
        rmism += bad
        del seq, prof, lab, want
    r16 = rlegs["T16"]
    return {"value": r16["mbases_per_s"], "unit": "Mbases/s", "cores": r16["threads"], "kind": "reference",
            "detail": "the reference's own functions (context.c, wall.c:245-1051 = its GSL-free part, class_rel.c, class_unrel.c, the paint "
                      "loop of ClassPro.c:263-269) in its thread loop without file I/O: per thread one set of scratch sized for "
                      "MAX_READ_LEN (ClassPro.c:110-143), contiguous read ranges (ClassPro.c:530); the three stale-scratch cells "
                      "(DESIGN 3.3, hazards 1, 2, 8) are reset per read inside the timed region so that labels are a function of the read",
            "sample": "first %d bases of the same resident read set (%.1f s at -T%d after %.1f s of per-thread allocation, reported apart)"
                      % (r16["bases"], r16["seconds"], r16["threads"], r16["startup_seconds"]),
            "seconds": r16["seconds"], "startup_seconds": r16["startup_seconds"], "label_mismatches_vs_hip": rmism + mism,
            "legs": rlegs, "host_cores": ncores, "host_cores_visible": ncores_seen, "cgroup_cpu_quota": quota,
            "port": port,
            "port_over_reference": {k: round(legs[k]["mbases_per_s"] / rlegs[k]["mbases_per_s"], 3) for k in ("T1", "T16")}}


def gpu_numa(dev):
    """(numa node of the GPU or -1, the node's cpu set or None) from sysfs."""
    import torch
    try:
        pr = torch.cuda.get_device_properties(dev)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return node, None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        return node, cpus
    except Exception:
        return -1, None


def pcie_pipeline(a, ds, clf, batches, dev):
    """The north star's host-feed pipeline, PCIe included: reads as they come from the files -- bases as characters, FASTK
    code strings -- sit in pinned host memory; per batch a feeder thread packs the bases to 2 bits (cp_pack_bases_batch,
    INSIDE the timed region), sends packed bases + codes + the four offset arrays (hipMemcpyAsync), runs cp_unpack_bases,
    cp_decode_profiles, cp_classify_batch, cp_pack_labels, and brings the 2-bit labels back; --pcie-slots batches in
    flight (stream + workspace + staging each, a host thread per slot, as the command line has).  The span is at least
    --pcie-seconds; the pinned-copy peaks it is judged against are measured in the same run, on the same buffers."""
    import torch
    import ctypes as C
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from classpro_amd._lib import lib, check
    L = lib()
    res = {}
    node, cpus = gpu_numa(dev)
    aff0 = os.sched_getaffinity(0)
    res["gpu_numa_node"] = node
    res["host_cpus_allowed"] = len(aff0)
    pinned_to = None
    if cpus and (cpus & aff0):
        os.sched_setaffinity(0, cpus & aff0)                # feeder / packing threads and first-touch of the pinned buffers on the GPU's node
        pinned_to = len(cpus & aff0)
    res["feeder_cpus_on_gpu_node"] = pinned_to
    # a box that gives this process few cores: the slots' packing threads share them instead of oversubscribing them sixfold
    ncpu = pinned_to or len(aff0)
    res["cgroup_cpu_quota"] = cpu_quota()                   # (packing is short bursts -- 9 cores' worth on average at 88 Gbases/s -- so the
                                                            #  slots keep their threads under a quota; extras.host_feed sizes by it)
    if a.pcie_slots * (a.pcie_pack_threads + 1) > ncpu:
        a.pcie_pack_threads = max(1, ncpu // max(1, a.pcie_slots) - 1)
    try:
        # ---- stage the inputs in pinned memory (untimed) ----
        want = int(a.pcie_gbases * 1e9)
        parts = []
        got = 0
        for rd, b in batches:
            if got >= want:
                break
            parts.append((rd, b))
            got += b.total_bases
        tgt = int(a.pcie_batch_mbases * 1e6)
        hb = []                                             # host batches
        pool = ThreadPoolExecutor(16)
        for rd, b in parts:
            so_all, po_all = rd["seq_off_h"], rd["prof_off_h"]
            r0 = 0
            while r0 < b.nreads and sum(x["bases"] for x in hb) < want:
                r1 = int(np.searchsorted(so_all, so_all[r0] + tgt, side="left"))
                r1 = min(max(r1, r0 + 1), b.nreads)
                if b.nreads - r1 < (r1 - r0) // 4:
                    r1 = b.nreads
                n = r1 - r0
                so = (so_all[r0:r1 + 1] - so_all[r0]).astype(np.int64)
                po = (po_all[r0:r1 + 1] - po_all[r0]).astype(np.int64)
                h_seq = torch.empty(int(so[-1]), dtype=torch.uint8).pin_memory()
                h_seq.copy_(rd["seq"][so_all[r0]:so_all[r1]])
                prof = rd["prof"][po_all[r0]:po_all[r1]].cpu().numpy().view(np.uint16)
                # FASTK code strings of the reads (host encoder, 16 threads; untimed: they are what the .prof files hold)
                bufs = [None] * n

                def enc(i, prof=prof, po=po, bufs=bufs):
                    pr_ = prof[po[i]:po[i + 1]]
                    buf = np.empty(2 * len(pr_) + 2, np.uint8)
                    m = L.cp_encode_profile(pr_.ctypes.data, len(pr_), buf.ctypes.data, len(buf))
                    bufs[i] = buf[:m]
                list(pool.map(enc, range(n), chunksize=256))
                co = np.zeros(n + 1, np.int64)
                np.cumsum([len(x) for x in bufs], out=co[1:])
                h_code = torch.empty(int(co[-1]), dtype=torch.uint8).pin_memory()
                hc_np = h_code.numpy()
                for i in range(n):
                    hc_np[co[i]:co[i + 1]] = bufs[i]
                pko = np.zeros(n + 1, np.int64)
                np.cumsum((np.diff(so) + 3) // 4, out=pko[1:])
                offs = torch.from_numpy(np.concatenate([so, po, co, pko])).pin_memory()     # the four offset arrays, one copy
                hb.append(dict(n=n, bases=int(so[-1]), kmers=int(po[-1]), ncode=int(co[-1]), npk=int(pko[-1]), h_seq=h_seq, h_code=h_code,
                               offs=offs, so=so, pko=pko, src=(rd, b, r0, r1)))
                r0 = r1
        pool.shutdown()
        NS = max(1, a.pcie_slots)
        mx = lambda k: max(x[k] for x in hb)
        sl = []
        for _ in range(NS):
            w = C.c_void_p()
            check(L.cp_workspace_create(C.byref(w)))
            sl.append(dict(st=torch.cuda.Stream(dev), ws=w,
                           h_pk=[torch.empty(mx("npk") + 64, dtype=torch.uint8).pin_memory() for _ in range(2)], flip=0,
                           h_plab=torch.empty(mx("npk") + 64, dtype=torch.uint8).pin_memory(),
                           d_pk=torch.empty(mx("npk") + 64, dtype=torch.uint8, device=dev),
                           d_code=torch.empty(mx("ncode") + 64, dtype=torch.uint8, device=dev),
                           d_offs=torch.empty(4 * (mx("n") + 1), dtype=torch.int64, device=dev),
                           d_seq=torch.empty(mx("bases") + 64, dtype=torch.uint8, device=dev),
                           d_prof=torch.empty(mx("kmers") + 64, dtype=torch.int16, device=dev),
                           d_lab=torch.empty(mx("bases") + 64, dtype=torch.uint8, device=dev),
                           d_plab=torch.empty(mx("npk") + 64, dtype=torch.uint8, device=dev), t_pack=0.0, done=[],
                           # labels as runs: (end, class) per run at the capacity offsets of the interval arrays (about 0.01 entries per base)
                           rcap=int(0.03 * mx("bases")) + 4096,
                           d_rend=torch.empty(int(0.03 * mx("bases")) + 4096, dtype=torch.int32, device=dev),
                           d_rcls=torch.empty(int(0.03 * mx("bases")) + 4096, dtype=torch.uint8, device=dev),
                           d_rmeta=torch.empty(3 * (mx("n") + 1), dtype=torch.int64, device=dev),      # cap_off[n+1] | nruns[n] (int32 pairs)
                           h_rend=torch.empty(int(0.03 * mx("bases")) + 4096, dtype=torch.int32).pin_memory(),
                           h_rcls=torch.empty(int(0.03 * mx("bases")) + 4096, dtype=torch.uint8).pin_memory(),
                           h_rmeta=torch.empty(3 * (mx("n") + 1), dtype=torch.int64).pin_memory(), out_b=0, last_cap=0))

        # ---- the pinned-copy peaks of this box, this process, these buffers ----
        big = max(hb, key=lambda x: x["bases"])
        dbuf = torch.empty(big["bases"], dtype=torch.uint8, device=dev)
        hback = torch.empty(big["bases"], dtype=torch.uint8).pin_memory()

        def rate(fn, nbytes, reps=6, trials=3):         # best of three: the first pinned copies of a process run slow now and then
            best = 0.0
            for _ in range(trials):
                fn(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                best = max(best, nbytes * reps / (time.perf_counter() - t0) / 1e9)
            return best
        NCS = 4                                             # copy streams per direction: the pipeline has several copies in flight too
        ss1, ss2 = [torch.cuda.Stream(dev) for _ in range(NCS)], [torch.cuda.Stream(dev) for _ in range(NCS)]
        nb4 = big["bases"] // NCS

        def h2d():
            for k, st_ in enumerate(ss1):
                with torch.cuda.stream(st_):
                    dbuf[k * nb4:(k + 1) * nb4].copy_(big["h_seq"][k * nb4:(k + 1) * nb4], non_blocking=True)

        def d2h():
            for k, st_ in enumerate(ss2):
                with torch.cuda.stream(st_):
                    hback[k * nb4:(k + 1) * nb4].copy_(dbuf[k * nb4:(k + 1) * nb4], non_blocking=True)

        def both():
            h2d(); d2h()
        res["pinned_h2d_peak_gb_per_s"] = round(rate(h2d, nb4 * NCS), 1)
        res["pinned_d2h_peak_gb_per_s"] = round(rate(d2h, nb4 * NCS), 1)
        res["pinned_both_ways_gb_per_s_in_plus_out"] = round(2 * rate(both, nb4 * NCS), 1)
        del dbuf, hback

        # ---- the pipeline ----
        mode = {"labels": "2bit"}                            # "2bit": cp_pack_labels; "runs": cp_label_runs (no painted string at all)

        def feed(k, order):
            q = sl[k]
            torch.cuda.set_device(dev)
            with torch.cuda.stream(q["st"]):
                sp = C.c_void_p(q["st"].cuda_stream)
                for j in order:
                    x = hb[j]
                    n = x["n"]
                    # the slot has two staging buffers: this batch's bases are packed while the slot's previous batch is still
                    # on the device (its H2D copy left the other buffer long ago: the batch before it had to finish first)
                    h_pk = q["h_pk"][q["flip"]]
                    q["flip"] ^= 1
                    t0 = time.perf_counter()
                    ok = L.cp_pack_bases_batch(x["h_seq"].data_ptr(), x["so"].ctypes.data, n, h_pk.data_ptr(), x["pko"].ctypes.data,
                                               a.pcie_pack_threads)
                    q["t_pack"] += time.perf_counter() - t0
                    if ok != 1:
                        raise RuntimeError("the synthetic reads should be pure ACGT")
                    q["st"].synchronize()                   # the slot's previous batch has left (its labels are in h_plab)
                    q["d_pk"][:x["npk"]].copy_(h_pk[:x["npk"]], non_blocking=True)
                    q["d_code"][:x["ncode"]].copy_(x["h_code"], non_blocking=True)
                    q["d_offs"][:4 * (n + 1)].copy_(x["offs"], non_blocking=True)
                    o = q["d_offs"].data_ptr()
                    d_so, d_po, d_co, d_pko = o, o + 8 * (n + 1), o + 16 * (n + 1), o + 24 * (n + 1)
                    check(L.cp_unpack_bases(q["d_pk"].data_ptr(), d_pko, d_so, n, q["d_seq"].data_ptr(), sp))
                    check(L.cp_decode_profiles(q["ws"], q["d_code"].data_ptr(), d_co, d_po, n, q["d_prof"].data_ptr(), sp))
                    if mode["labels"] == "2bit":
                        check(L.cp_classify_batch(clf.p, q["ws"], q["d_seq"].data_ptr(), d_so, q["d_prof"].data_ptr(), d_po, n,
                                                  x["bases"], x["kmers"], q["d_lab"].data_ptr(), sp))
                        check(L.cp_pack_labels(q["d_lab"].data_ptr(), d_so, d_pko, n, q["d_plab"].data_ptr(), sp))
                        q["h_plab"][:x["npk"]].copy_(q["d_plab"][:x["npk"]], non_blocking=True)
                        q["out_b"] += x["npk"]
                    else:
                        check(L.cp_run_stages(clf.p, q["ws"], q["d_seq"].data_ptr(), d_so, q["d_prof"].data_ptr(), d_po, n,
                                              x["bases"], x["kmers"], None, 5, sp))                 # CP_STAGE_CLASS_ALL: no label string
                        cap = int(L.cp_label_runs_capacity(q["ws"]))
                        if cap > q["rcap"]:
                            raise RuntimeError("label-run buffers too small: %d > %d" % (cap, q["rcap"]))
                        m = q["d_rmeta"].data_ptr()
                        check(L.cp_label_runs(clf.p, q["ws"], q["d_rend"].data_ptr(), q["d_rcls"].data_ptr(), m + 8 * (n + 1), m, sp))
                        q["h_rend"][:cap].copy_(q["d_rend"][:cap], non_blocking=True)
                        q["h_rcls"][:cap].copy_(q["d_rcls"][:cap], non_blocking=True)
                        nm = (n + 1) + (n + 1) // 2 + 1
                        q["h_rmeta"][:nm].copy_(q["d_rmeta"][:nm], non_blocking=True)
                        q["out_b"] += 5 * cap + 8 * nm
                        q["last_cap"] = cap
                    q["done"].append(j)
                q["st"].synchronize()

        def run(order):
            errs = []

            def guarded(k):
                try:
                    feed(k, order[k::NS])
                except Exception as e:     # noqa: BLE001
                    errs.append(e)
            th = [threading.Thread(target=guarded, args=(k,)) for k in range(NS)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            if errs:
                raise errs[0]
        nbt = len(hb)
        from classpro_amd.api import unpack_labels, expand_label_runs
        for lab_mode in ("2bit", "runs"):
            mode["labels"] = lab_mode
            for q in sl:
                q["done"] = []
            run(list(range(min(nbt, 2 * NS))))              # warm-up: workspaces grow, pages are touched
            for q in sl:
                q["t_pack"], q["out_b"] = 0.0, 0
            t0 = time.perf_counter()
            run(list(range(nbt)))
            t1 = time.perf_counter() - t0
            passes = 1
            if t1 < a.pcie_seconds:                         # repeat the staged set until the span is long enough
                more = int(np.ceil(1.6 * a.pcie_seconds / t1))  # (the first pass is the slowest)
                for q in sl:
                    q["t_pack"], q["out_b"] = 0.0, 0
                t0 = time.perf_counter()
                run(list(range(nbt)) * more)
                t1 = time.perf_counter() - t0
                passes = more
            for q in sl:
                check(L.cp_workspace_check(q["ws"]))
            tot_bases = sum(x["bases"] for x in hb) * passes
            in_bytes = sum(x["npk"] + x["ncode"] + 32 * (x["n"] + 1) for x in hb) * passes
            out_bytes = sum(q["out_b"] for q in sl)
            # every slot's last batch: the labels that came back == the resident run's labels
            ok = True
            for q in sl:
                j = q["done"][-1]
                x = hb[j]
                rd, b, r0, r1 = x["src"]
                so_all = rd["seq_off_h"]
                want_lab = b.labels[so_all[r0]:so_all[r1]].cpu().numpy()
                if lab_mode == "2bit":
                    got = unpack_labels(q["h_plab"].numpy(), x["pko"], np.diff(x["so"]), K)
                    ok &= bool(np.array_equal(got, want_lab))
                else:
                    n = x["n"]
                    meta = q["h_rmeta"].numpy()
                    coff = meta[:n + 1]
                    nr = meta[n + 1:n + 1 + (n + 1) // 2 + 1].view(np.int32)[:n]
                    ends, cls = q["h_rend"].numpy(), q["h_rcls"].numpy()
                    for r in range(0, n, max(1, n // 200)):  # a sample of the batch's reads, expanded on the host
                        got = expand_label_runs(ends[coff[r]:coff[r] + nr[r]], cls[coff[r]:coff[r] + nr[r]], int(x["so"][r + 1] - x["so"][r]), K)
                        ok &= got == want_lab[x["so"][r]:x["so"][r + 1]].tobytes()
            one = {"mbases_per_s": round(tot_bases / t1 / 1e6, 1), "seconds": round(t1, 3), "bases": tot_bases,
                   "distinct_gbases_staged": round(sum(x["bases"] for x in hb) / 1e9, 2), "passes_over_staged_set": passes,
                   "batches": nbt * passes, "batch_mbases": round(np.mean([x["bases"] for x in hb]) / 1e6, 1), "slots": NS,
                   "pack_threads_per_slot": a.pcie_pack_threads,
                   "host_pack_seconds_per_slot": [round(q["t_pack"], 3) for q in sl],
                   "bytes_per_base_in_out": [round(in_bytes / tot_bases, 3), round(out_bytes / tot_bases, 3)],
                   "h2d_gb_per_s": round(in_bytes / t1 / 1e9, 2), "d2h_gb_per_s": round(out_bytes / t1 / 1e9, 2),
                   "labels_match_resident_run": ok}
            # the pipeline moves data both ways at once: it is judged against what pinned copies both ways at once reach
            one["link_gb_per_s_in_plus_out"] = round((in_bytes + out_bytes) / t1 / 1e9, 2)
            one["frac"] = round(one["link_gb_per_s_in_plus_out"] / max(res["pinned_both_ways_gb_per_s_in_plus_out"], res["pinned_h2d_peak_gb_per_s"]), 3)
            one["h2d_frac_of_one_way_peak"] = round(one["h2d_gb_per_s"] / res["pinned_h2d_peak_gb_per_s"], 3)
            if lab_mode == "2bit":
                one["labels_out"] = "2-bit codes (cp_pack_labels), the .class.data payload"
                res.update(one)
                res["frac_of"] = "pinned copies both ways at once (in + out), %d streams each way, measured in this run" % NCS
            else:
                one["labels_out"] = "runs (cp_label_runs): (end, class) per stretch of one class; the label string is never painted on the device"
                res["label_runs"] = one
        for q in sl:
            L.cp_workspace_destroy(q["ws"])
    finally:
        os.sched_setaffinity(0, aff0)
    return res


def numa_nodes():
    """[(node id, cpu set)] from sysfs, restricted to the cpus this process may use; [( -1, all )] when sysfs says nothing."""
    aff = os.sched_getaffinity(0)
    out = []
    try:
        for d in sorted(os.listdir("/sys/devices/system/node")):
            if not d.startswith("node") or not d[4:].isdigit():
                continue
            cpus = set()
            for part in open("/sys/devices/system/node/%s/cpulist" % d).read().strip().split(","):
                if part:
                    lo, _, hi = part.partition("-")
                    cpus.update(range(int(lo), int(hi or lo) + 1))
            if cpus & aff:
                out.append((int(d[4:]), cpus & aff))
    except Exception:
        out = []
    return out or [(-1, aff)]


def cpu_quota():
    """CPUs' worth of time the cgroup grants this process (cpu.max), or None when unlimited / unknown."""
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = open(f).read().split()
            if f.endswith("cpu.max"):
                return None if t[0] == "max" else round(int(t[0]) / int(t[1]), 2)
            q = int(t[0])
            return None if q <= 0 else round(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()), 2)
        except Exception:
            continue
    return None


def host_feed(a, batches, groups=(1, 2, 4, 8)):
    """The HOST half of the PCIe pipeline alone, no device: what one box sustains when it feeds 1, 2, 4, 8 GPUs at once.
    A feeder group stands for one GPU's feeders (pcie_pipeline: --pcie-slots slot threads, each packing its batch's bases
    to 2 bits on --pcie-pack-threads threads -- cp_pack_bases_batch -- into one of two staging buffers and copying the
    batch's FASTK code bytes into a staging buffer, the host-DRAM traffic of the DMA read); group g lives on NUMA node
    g mod nodes (threads pinned there, its source reads and staging buffers first touched there).  Each group has its own
    copy of the source reads (bases as characters + 0.274 B/base of code bytes), far larger than the last-level cache.
    Reports Gbases/s per group count, host-DRAM GB/s (1 B/base read + 0.25 written by the packing, 2 x 0.274 by the code
    copy) and the threads in use.  README.md:71 / io.c:353-354 of the reference name I/O as ITS bottleneck; this is ours."""
    import ctypes as C
    import threading
    from classpro_amd._lib import lib
    L = lib()
    nodes = numa_nodes()
    rd0, b0 = batches[0]
    per_group = int(min(a.host_feed_gbases * 1e9, b0.total_bases))
    tgt = int(a.pcie_batch_mbases * 1e6)
    so_all = rd0["seq_off_h"]
    nreads = int(np.searchsorted(so_all, per_group, side="right") - 1)
    nreads = max(nreads, 1)
    so = so_all[:nreads + 1].astype(np.int64)
    src = rd0["seq"][:so[-1]].cpu().numpy()
    code_bpb = 0.274                                        # FASTK code bytes per base of this workload (extras.code_bytes_per_base)
    # batches of ~tgt bases: (first read, last read)
    cuts = [0]
    while cuts[-1] < nreads:
        r1 = int(np.searchsorted(so, so[cuts[-1]] + tgt, side="left"))
        cuts.append(min(max(r1, cuts[-1] + 1), nreads))
    bt = list(zip(cuts[:-1], cuts[1:]))
    pko = np.zeros(nreads + 1, np.int64)
    np.cumsum((np.diff(so) + 3) // 4, out=pko[1:])
    aff0 = os.sched_getaffinity(0)
    quota = cpu_quota()
    cores = len(aff0) if not quota else max(1, min(len(aff0), int(quota)))       # what can run at once
    res = {"numa_nodes": [[n, len(c)] for n, c in nodes], "cpus_allowed": len(aff0), "cgroup_cpu_quota": quota, "cores_usable": cores,
           "batch_mbases": round(tgt / 1e6, 1), "source_gbases_per_group": round(int(so[-1]) / 1e9, 2),
           "dram_bytes_per_base": round(1.25 + 2 * code_bpb, 3), "legs": {}}
    maxb = max(int(so[r1] - so[r0]) for r0, r1 in bt)
    state = {}

    def plan(G):
        """slots per group and pack threads per slot: the pipeline's own (--pcie-slots x --pcie-pack-threads) when the box has the
        cores for G such groups, else as many busy threads as it can run at once, split evenly"""
        ns = max(1, min(a.pcie_slots, cores // G))
        return ns, max(1, min(a.pcie_pack_threads, cores // (G * ns)))

    def setup(g):                                           # runs on a thread pinned to the group's node: first touch there
        node, cpus = nodes[g % len(nodes)]
        os.sched_setaffinity(0, cpus)
        NSm = max(1, a.pcie_slots)
        st = dict(seq=src.copy(), code=np.ones(int(code_bpb * int(so[-1])) + 64, np.uint8),
                  pk=[[np.zeros(maxb // 4 + nreads + 64, np.uint8) for _ in range(2)] for _ in range(NSm)],
                  cst=[np.zeros(int(code_bpb * maxb) + 64, np.uint8) for _ in range(NSm)])
        state[g] = st

    def slot(g, k, NS, P, t_end, out):
        node, cpus = nodes[g % len(nodes)]
        os.sched_setaffinity(0, cpus)
        st = state[g]
        done, flip, j = 0, 0, k
        while time.perf_counter() < t_end:
            r0, r1 = bt[j % len(bt)]
            j += NS
            n = r1 - r0
            nb = int(so[r1] - so[r0])
            loc_so = np.ascontiguousarray(so[r0:r1 + 1])
            loc_pk = np.ascontiguousarray(pko[r0:r1 + 1] - pko[r0])
            ok = L.cp_pack_bases_batch(st["seq"].ctypes.data, loc_so.ctypes.data, n, st["pk"][k][flip].ctypes.data, loc_pk.ctypes.data, P)
            flip ^= 1
            if ok != 1:
                raise RuntimeError("the synthetic reads should be pure ACGT")
            nc = int(code_bpb * nb)
            C.memmove(st["cst"][k].ctypes.data, st["code"].ctypes.data + int(code_bpb * int(so[r0])), nc)
            done += nb
        out.append(done)

    def copier(g, t_end, out):                              # the yardstick: plain memmove of the group's bases, thread pinned like a feeder
        node, cpus = nodes[g % len(nodes)]
        os.sched_setaffinity(0, cpus)
        a_, n, done = state[g % len(state)]["seq"], int(so[-1]) // 2, 0
        while time.perf_counter() < t_end:
            C.memmove(a_.ctypes.data + n, a_.ctypes.data, n)
            done += 2 * n                                   # bytes read + written
        out.append(done)

    try:
        for G in groups:
            for g in range(G):
                if g not in state:
                    t = threading.Thread(target=setup, args=(g,))
                    t.start(); t.join()
            NS, P = plan(G)
            outs = []
            t_end = time.perf_counter() + a.host_feed_seconds
            t0 = time.perf_counter()
            th = [threading.Thread(target=slot, args=(g, k, NS, P, t_end, outs)) for g in range(G) for k in range(NS)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            dt = time.perf_counter() - t0
            tot = sum(outs)
            res["legs"]["%d" % G] = {"feeder_groups": G, "slots_per_group": NS, "pack_threads_per_slot": P, "busy_threads": G * NS * P,
                                     "gbases_per_s": round(tot / dt / 1e9, 1), "per_group_gbases_per_s": round(tot / dt / 1e9 / G, 1),
                                     "per_busy_thread_gbases_per_s": round(tot / dt / 1e9 / (G * NS * P), 2),
                                     "host_dram_gb_per_s": round(tot * (1.25 + 2 * code_bpb) / dt / 1e9, 1), "seconds": round(dt, 2)}
        # plain copies on as many threads as the box runs at once, spread over the nodes: the DRAM rate the figures above sit beside
        outs = []
        t_end = time.perf_counter() + 1.0
        t0 = time.perf_counter()
        th = [threading.Thread(target=copier, args=(g, t_end, outs)) for g in range(cores)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        res["memmove_gb_per_s_read_plus_write"] = round(sum(outs) / (time.perf_counter() - t0) / 1e9, 1)
        res["memmove_threads"] = cores
        best = max(v["per_busy_thread_gbases_per_s"] for v in res["legs"].values())
        res["per_core_gbases_per_s"] = best
    finally:
        os.sched_setaffinity(0, aff0)
        state.clear()
    return res


def extra_rates(a, ds, clf, batches, hcov, dcov, dev, stream):
    """Reported beside `value`, never as `value`: the MHC-like configs[1] batch, the PCIe-inclusive rate of the
    drop-in's own transfer pattern, and the drop-in binary end to end on files in tmpfs."""
    import torch
    import ctypes as C
    from classpro_amd.synth_dev import DeviceSynth
    from classpro_amd.api import Batch, encode_profiles
    from classpro_amd._lib import lib, check
    L = lib()
    ex = {}
    # the PCIe-inclusive pipeline on the bench workload itself (bases as characters + FASTK codes in pinned host memory)
    try:
        ex["pcie"] = pcie_pipeline(a, ds, clf, batches, dev)
    except Exception as e:                                  # the extras never take the bench line down
        ex["pcie_error"] = repr(e)[:300]
    try:
        ex["host_feed"] = host_feed(a, batches)
        one = ex.get("pcie", {}).get("mbases_per_s")
        if one:                                             # what 8 GPUs at the measured single-GPU PCIe-inclusive rate would ask of the host
            hf_ = ex["host_feed"]
            hf_["asked_by_8_gpus_gbases_per_s"] = round(8 * one / 1e3, 1)
            hf_["asked_by_8_gpus_host_dram_gb_per_s"] = round(8 * one / 1e3 * hf_["dram_bytes_per_base"], 1)
            hf_["asked_by_8_gpus_packing_cores"] = int(np.ceil(8 * one / 1e3 / max(hf_["per_core_gbases_per_s"], 1e-9)))
    except Exception as e:
        ex["host_feed_error"] = repr(e)[:300]
    if a.only_pcie:
        return ex
    # MHC-like set (BASELINE configs[1] stand-in): 5 Mbp x 40x = 10 000 reads, 200 Mbases, ONE batch per step
    m = DeviceSynth(genome_len=5_000_000, cov=a.cov, read_len=a.read_len, K=K, seed=a.seed, device=str(dev))
    rdm = m.reads(0, m.n_reads)
    bm = Batch.from_device(rdm)
    for _ in range(3):
        clf.run(bm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        clf.run(bm)
    torch.cuda.synchronize()
    ex["mhc_like_200mbase_batch_mbases_per_s"] = round(bm.total_bases * 20 / (time.perf_counter() - t0) / 1e6, 1)
    clf.check()

    # PCIe-inclusive, the drop-in's own transfer pattern on the MHC-like batch: pinned bases + FASTK code strings
    # in, cp_decode_profiles, classify, pinned labels out (un-overlapped)
    hm = m.to_host(rdm, names=False)
    codes, code_off = encode_profiles(hm["profiles"])
    h_seq = torch.from_numpy(np.frombuffer(b"".join(hm["seqs"]), np.uint8).copy()).pin_memory()
    h_code = torch.from_numpy(codes).pin_memory()
    h_lab = torch.empty(bm.total_bases, dtype=torch.uint8).pin_memory()
    d_coff = torch.from_numpy(code_off).to(dev)
    d_prof = torch.empty_like(bm.prof)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        d_seq = h_seq.to(dev, non_blocking=True)
        d_code = h_code.to(dev, non_blocking=True)
        check(L.cp_decode_profiles(clf.ws, d_code.data_ptr(), d_coff.data_ptr(), bm.prof_off.data_ptr(), bm.nreads,
                                   d_prof.data_ptr(), stream))
        check(L.cp_classify_batch(clf.p, clf.ws, d_seq.data_ptr(), bm.seq_off.data_ptr(), d_prof.data_ptr(),
                                  bm.prof_off.data_ptr(), bm.nreads, bm.total_bases, bm.total_kmers, bm.labels.data_ptr(), stream))
        h_lab.copy_(bm.labels[:bm.total_bases], non_blocking=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    clf.check()
    ex["pcie_inclusive_pinned_codes_mbases_per_s"] = round(bm.total_bases / min(ts) / 1e6, 1)
    ex["code_bytes_per_base"] = round(len(codes) / bm.total_bases, 4)
    ex["decode_matches"] = bool(torch.equal(d_prof[:bm.total_kmers], bm.prof[:bm.total_kmers]))
    del m, rdm, bm, d_prof, h_seq, h_code, h_lab

    # BASELINE configs[4] stand-in: 60x, r=25000, with the -s seed path (cp_find_seeds_batch after the classification)
    try:
        from classpro_amd.api import Classifier, hist_covs
        s4 = DeviceSynth(genome_len=20_000_000, cov=60, read_len=25000, K=K, seed=a.seed + 4, device=str(dev))
        h4, d4 = hist_covs(s4.hist[4], 1, 32767, 0, 0, 0)
        c4 = Classifier(K=K, read_len=25000, hcov=h4, dcov=d4, device=str(dev))
        b4 = Batch.from_device(s4.reads(0, s4.n_reads))
        seeds4 = torch.zeros(b4.total_bases, dtype=torch.uint8, device=dev)

        def both():
            c4.run(b4)
            check(L.cp_find_seeds_batch(c4.p, c4.ws, b4.seq.data_ptr(), b4.seq_off.data_ptr(), b4.prof.data_ptr(), b4.prof_off.data_ptr(),
                                        b4.labels.data_ptr(), b4.nreads, b4.total_bases, b4.total_kmers, seeds4.data_ptr(), stream))
        both()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            c4.run(b4)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            both()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        c4.check()
        ex["config4_60x_r25000"] = {"bases": b4.total_bases, "hcov": h4, "dcov": d4,
                                    "classify_mbases_per_s": round(b4.total_bases * 3 / (t1 - t0) / 1e6, 1),
                                    "classify_plus_seeds_mbases_per_s": round(b4.total_bases * 3 / (t2 - t1) / 1e6, 1),
                                    "seed_kmers_fraction": round(float((seeds4 != ord("E")).float().mean()), 5)}
        c4.close()
        del s4, b4, seeds4
    except Exception as e:
        ex["config4_error"] = repr(e)[:200]

    # the drop-in binary end to end: FASTA + FASTK files of the first ~1.6 Gbases of the resident set written to
    # tmpfs, `ClassPro -T16` from process start to exit, .class (2 B/base) written next to them
    try:
        import shutil
        import tempfile
        sys.path.insert(0, os.path.join(_ROOT, "scripts"))
        import cli_e2e
        parts, got = [], 0
        for rd, b in batches:
            if got >= 1.6e9:
                break
            parts.append((rd, b))
            got += b.total_bases
        seq = np.concatenate([rd["seq"][:b.total_bases].cpu().numpy() for rd, b in parts])
        prof = np.concatenate([rd["prof"][:b.total_kmers].cpu().numpy().view(np.uint16) for rd, b in parts])
        lens = np.concatenate([np.diff(rd["seq_off_h"]) for rd, b in parts])
        so = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        d = tempfile.mkdtemp(prefix="cp_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            path = cli_e2e.write_inputs(d, (seq, prof), so, ds.hist)
            dt, lines = cli_e2e.run_cli(path, 16, reps=1)
            ex["cli_end_to_end_mbases_per_s"] = round(int(so[-1]) / dt / 1e6, 1)
            ex["cli_end_to_end"] = {"bases": int(so[-1]), "seconds_process_wall": round(dt, 3), "threads": 16,
                                    "files": "plain FASTA + FASTK in, .class out, all on tmpfs", "phase_lines": lines}
            # the binary's labels == the resident run's labels (first 2000 records)
            lab = np.concatenate([b.labels[:b.total_bases].cpu().numpy() for rd, b in parts])
            ok, k = True, 0
            with open(os.path.join(d, "reads.class"), "rb") as f:
                for i in range(2000):
                    f.readline(); f.readline(); f.readline()
                    ok &= f.readline()[:-1] == lab[so[i]:so[i + 1]].tobytes()
            ex["cli_labels_match_resident_run"] = bool(ok)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    except Exception as e:                                  # the extras never take the bench line down
        ex["cli_end_to_end_error"] = repr(e)[:200]
    return ex


if __name__ == "__main__":
    main()
