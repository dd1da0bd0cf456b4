"""N>1 path: reads shard across ranks by bases with no data-path collective; the only exchange is the
ordered host-side gather of per-shard label fragments (the reference's merge_files, io.c:70-112).
Runs world_size 2 over gloo on CPU; the per-shard classifier here is the oracle (tests may use it)."""
import os
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from classpro_amd import synth
    from classpro_amd.shard import plan_shards, gather_fragments
    from oracle.oracle import Oracle
    ds = synth.make_dataset(genome_len=60000, cov=30, read_len=5000, seed=3)
    seq, so, prof, po = synth.pack_batch(ds["seqs"], ds["profiles"])
    bounds = plan_shards(so, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    O = Oracle(40, 20000, 15, 30)
    frag = O.classify_batch(seq[so[lo]:so[hi]], so[lo:hi + 1] - so[lo], prof[po[lo]:po[hi]], po[lo:hi + 1] - po[lo], nthreads=1)
    merged = gather_fragments(frag, rank, world)
    if rank == 0:
        whole = O.classify_batch(seq, so, prof, po, nthreads=2)
        q.put((bool(np.array_equal(merged, whole)), [int(b) for b in bounds], int(so[-1])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, bounds, total = q.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok
    assert bounds[0] == 0 and bounds[1] > 0 and bounds[-1] > bounds[1]


def test_plan_shards_balances_bases():
    from classpro_amd.shard import plan_shards
    rng = np.random.default_rng(0)
    lens = rng.integers(3000, 60000, 5000)
    so = np.concatenate([[0], np.cumsum(lens)])
    for w in (1, 2, 4, 8):
        b = plan_shards(so, w)
        assert b[0] == 0 and b[-1] == len(lens) and all(b[i] <= b[i + 1] for i in range(w))
        per = [so[b[i + 1]] - so[b[i]] for i in range(w)]
        assert max(per) - min(per) <= 2 * 60000
    assert plan_shards(np.array([0]), 4) == [0, 0, 0, 0, 0]
