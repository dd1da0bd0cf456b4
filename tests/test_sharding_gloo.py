"""N>1 path: reads shard across ranks by bases with no data-path collective; the only exchange is the
ordered host-side gather of per-shard label fragments (the reference's merge_files, io.c:70-112).
World size 2 over gloo on CPU through the PRODUCT's shard driver (classpro_amd.shard.classify_sharded: plan, slice,
rebase, gather).  There is no GPU in this container, so the per-shard classifier plugged into the driver here is the
oracle; tests/test_gpu_sharded.py runs the same driver with the HIP Classifier on two ranks that share one MI355X."""
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
    from classpro_amd.shard import classify_sharded
    from oracle.oracle import Oracle
    ds = synth.make_dataset(genome_len=60000, cov=30, read_len=5000, seed=3)
    seq, so, prof, po = synth.pack_batch(ds["seqs"], ds["profiles"])
    O = Oracle(40, 20000, 15, 30)
    calls = []

    def classify_fn(s, so_, p, po_):
        calls.append((len(so_) - 1, int(so_[0]), int(po_[0])))
        return O.classify_batch(s, so_, p, po_, nthreads=1)
    merged, bounds = classify_sharded(classify_fn, seq, so, prof, po, rank, world)
    assert len(calls) == 1 and calls[0][1] == 0 and calls[0][2] == 0          # one shard per rank, offsets rebased
    assert calls[0][0] == bounds[rank + 1] - bounds[rank]
    if rank == 0:
        whole = O.classify_batch(seq, so, prof, po, nthreads=2)
        q.put((bool(np.array_equal(merged, whole)), [int(b) for b in bounds], int(so[-1])))
    else:
        assert merged is None
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


def test_plan_shards_snaps_to_fastk_parts():
    """With the first reads of the FASTK profile parts given, a boundary moves to a part boundary when that costs at most
    5 % of a shard's bases -- and stays where the bases put it when no part boundary is near."""
    from classpro_amd.shard import plan_shards
    rng = np.random.default_rng(3)
    lens = rng.integers(3000, 30000, 4000)
    so = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    plain = plan_shards(so, 4)
    near = [0, plain[1] + 7, plain[2] - 11, 3500]             # parts that begin a few reads away from boundaries 1 and 2 (and one far from 3)
    b = plan_shards(so, 4, part_first=near)
    assert b[1] == plain[1] + 7 and b[2] == plain[2] - 11 and b[3] == plain[3] and b[0] == 0 and b[4] == 4000
    share = np.diff(so[b])
    assert share.max() - share.min() < 0.1 * so[-1] / 4
    far = [0, 100, 3900]
    assert plan_shards(so, 4, part_first=far) == plain
    assert plan_shards(so, 4, part_first=[]) == plain
