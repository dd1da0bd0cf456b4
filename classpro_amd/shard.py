"""Read sharding across GPUs.

Reads are independent once the global scalars are known, so the N-GPU path is: contiguous read ranges
balanced by *bases* (not reads), one process per GPU, no collective on the data path, and an ordered
host-side gather of the per-shard label fragments -- the role merge_files plays for the reference's
per-thread temp files (io.c:70-112).
"""
import numpy as np


def plan_shards(seq_off, world, part_first=None, slack=0.05):
    """Boundaries b[0..world] (read indices): shard r = reads [b[r], b[r+1]), about equal in bases.
    `part_first`: first read of every FASTK profile part (`.pidx.N` headers, libfastk.c:1305-1308, 1330-1335); a shard
    boundary then moves to the nearest part boundary when that costs at most `slack` of a shard's bases, so that a
    shard streams whole `.prof.N` files where the parts allow it ("ideally aligned to FASTK's own
    part boundaries")."""
    seq_off = np.asarray(seq_off, dtype=np.int64)
    n = len(seq_off) - 1
    total = int(seq_off[-1]) if n > 0 else 0
    parts = np.unique(np.asarray(part_first, dtype=np.int64)) if part_first is not None and len(part_first) else None
    b = [0]
    for r in range(1, world):
        target = total * r // world
        k = int(np.searchsorted(seq_off, target, side="left")) if n else 0
        if parts is not None and n:
            j = int(np.searchsorted(parts, k))
            cand = [int(parts[x]) for x in (j - 1, j) if 0 <= x < len(parts) and 0 <= parts[x] <= n]
            if cand:
                best = min(cand, key=lambda q: abs(int(seq_off[q]) - target))
                if abs(int(seq_off[best]) - target) * world <= slack * total:
                    k = best
        b.append(k)
    b.append(n)
    for i in range(1, len(b)):
        b[i] = min(max(b[i], b[i - 1]), n)
    return b


def gather_fragments(fragment, rank, world, group=None):
    """Concatenate per-rank label fragments in rank order on rank 0 (host side, gloo or nccl group)."""
    import torch.distributed as dist
    frag = np.ascontiguousarray(fragment, dtype=np.uint8)
    if world == 1:
        return frag
    out = [None] * world if rank == 0 else None
    dist.gather_object(frag, out, dst=0, group=group)
    if rank == 0:
        return np.concatenate(out)
    return None


def shard_slice(seq, seq_off, prof, prof_off, lo, hi):
    """Reads [lo,hi) of a flat read set as a flat read set of their own (offsets rebased to 0)."""
    seq_off = np.asarray(seq_off, dtype=np.int64)
    prof_off = np.asarray(prof_off, dtype=np.int64)
    return (seq[seq_off[lo]:seq_off[hi]], seq_off[lo:hi + 1] - seq_off[lo],
            prof[prof_off[lo]:prof_off[hi]], prof_off[lo:hi + 1] - prof_off[lo])


def classify_sharded(classify_fn, seq, seq_off, prof, prof_off, rank, world, group=None):
    """The N>1 path for a host-resident read set: this rank takes its contiguous read range (plan_shards: balanced by
    bases), classifies it with `classify_fn(seq, seq_off, prof, prof_off) -> label bytes` (on a GPU box:
    `lambda *a: clf.classify(Batch(*a))`), and the fragments are concatenated in rank order on rank 0 -- the ordered
    merge of the reference's per-thread files (io.c:70-112).  No collective touches the data path.
    Returns (labels on rank 0 / None elsewhere, the shard boundaries)."""
    bounds = plan_shards(seq_off, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    if hi > lo:
        frag = np.asarray(classify_fn(*shard_slice(seq, seq_off, prof, prof_off, lo, hi)), dtype=np.uint8)
    else:
        frag = np.zeros(0, np.uint8)
    return gather_fragments(frag, rank, world, group), bounds
