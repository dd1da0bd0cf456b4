"""The product's N>1 path with the HIP classifier (`-m gpu`): two ranks (gloo rendezvous; both on the one MI355X of
the box, as the driver's 8-GPU run puts one rank on each GPU) classify their shard of ONE read set through
classpro_amd.shard.classify_sharded; rank 0's ordered gather equals the single-rank labels and the CPU oracle's byte for byte."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from classpro_amd import synth
    from classpro_amd.api import Classifier, Batch
    from classpro_amd.shard import classify_sharded
    ds = synth.make_dataset(genome_len=300000, cov=40, read_len=10000, seed=12)
    seq, so, prof, po = synth.pack_batch(ds["seqs"], ds["profiles"])
    clf = Classifier(K=40, read_len=20000, hcov=20, dcov=40, device="cuda:0")
    merged, bounds = classify_sharded(lambda *a: clf.classify(Batch(*a)), seq, so, prof, po, rank, world)
    if rank == 0:
        whole = clf.classify(Batch(seq, so, prof, po))
        from oracle.oracle import Oracle                      # (the checker: the gathered labels against the CPU oracle as well)
        want = Oracle(40, 20000, 20, 40).classify_batch(seq, so, prof, po, nthreads=4)
        q.put((bool(np.array_equal(merged, whole)) and bool(np.array_equal(merged, want)), [int(b) for b in bounds], int(so[-1])))
    clf.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_read_set(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, bounds, total = q.get(timeout=600)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ok and 0 < bounds[1] < bounds[2]
