"""Adversarial classifier inputs shared by the host-logic (CPU) and device (GPU) parity tests."""
import numpy as np


def adversarial_reads(seed, n=250, K=40):
    """Inputs that no generator of 'plausible' data would make: long homopolymers and micro-satellites (context
    caps, partners far beyond the paired window), count profiles that jump between error, haploid, diploid,
    repeat and very high levels with noise on top (many candidates per read: the on-chip memo overflows to its
    global table), very short and very long reads."""
    rng = np.random.default_rng(seed)
    seqs, profs = [], []
    for _ in range(n):
        L = int(rng.choice([60, 90, 300, 1500, 4000, 9000]) + rng.integers(0, 40))
        L = max(L, K + 1)                                               # (K > 59: at least two k-mers)
        s = rng.integers(0, 4, L)
        for _r in range(int(rng.integers(0, 6))):                       # low-complexity runs
            unit = rng.integers(0, 4, int(rng.integers(1, 4)))
            run = int(rng.choice([8, 20, 60, 150, 300]))
            p0 = int(rng.integers(0, max(1, L - 1)))
            seg = np.resize(unit, run)[:max(0, min(run, L - p0))]
            s[p0:p0 + len(seg)] = seg
        plen = L - K + 1
        levels = rng.choice([1, 2, 4, 19, 21, 38, 41, 60, 75, 120, 400, 3000, 32767], size=plen // 40 + 2)
        c = np.repeat(levels, 40)[:plen].astype(np.int64)
        c = np.roll(c, int(rng.integers(0, 40)))
        noisy = rng.random(plen) < rng.choice([0.0, 0.02, 0.3])
        c = np.where(noisy, c + rng.integers(-6, 7, plen), c)
        seqs.append(bytes(b"ACGT"[x] for x in s))
        profs.append(np.clip(c, 0, 32767).astype(np.uint16))
    return seqs, profs


def tail_run_reads(seed, n=200, K=40):
    """Reads that END in a low-complexity run of >= K bases (homopolymer, di- or tri-nucleotide) with a slow count
    ramp over the last interval: the case where correct_wall_cnt's `last = I.b+lmax` (wall.c:976-978) equals plen and
    the reference reads profile[plen], a cell that belongs to no read (hazard 8, DESIGN 3.3: defined as 0).  Before
    that was defined, most such reads changed labels (H <-> D over the whole run) with the first count of the NEXT
    read of the batch."""
    rng = np.random.default_rng(seed)
    seqs, profs = [], []
    for k in range(n):
        body = int(rng.integers(200, 1500))
        ulen = 1 + k % 3
        run = int(rng.integers(K + 2, 121))                    # <= 127: the context values are capped there ...
        if k % 8 == 7:                                         # ... except here: a homopolymer of 130-320 bases, where rctx holds the
            ulen, run = 1, int(rng.integers(130, 321))         # reversed copy of CAPPED values (context.c:24-25): lmax > bases left
        s = rng.integers(0, 4, body)
        unit = rng.permutation(4)[:ulen]                       # distinct bases: a true period-`ulen` unit
        s[-1] = (unit[-1] + 1 + int(rng.integers(0, 3))) % 4   # the base before the run breaks its period
        s = np.concatenate([s, np.resize(unit, run)])
        plen = len(s) - K + 1
        # counts: diploid level up to the first k-mer whose last base lies in the run, then a drop to a haploid-ish
        # level and a slow ramp (a few unit or double steps, most of them in the first K-1 positions) to the end
        c = np.full(plen, int(rng.integers(38, 43)), np.int64)
        b = max(1, min(body - K + 1 + int(rng.integers(0, 3)), plen - K))
        lvl = int(rng.integers(20, 29))
        # (an interval this short is reliable only if its corrected end counts agree to within ~2, so: an upward ramp
        #  inside the first K-1 positions -- which correct_wall_cnt adds back to the begin count -- unless the phantom
        #  downward step into profile[plen] cancels it)
        steps = np.zeros(plen - b, np.int64)
        nst = int(rng.integers(4, 11))
        kind = rng.random()
        where = rng.integers(1, max(2, min(K - 1, plen - b)), nst) if kind < 0.8 else rng.integers(1, plen - b, nst)
        np.add.at(steps, where, rng.integers(1, 3, nst))
        sign = 1 if kind < 0.9 else -1
        c[b:] = np.clip(lvl + sign * np.cumsum(steps), 1, 70)
        seqs.append(bytes(b"ACGT"[x] for x in s))
        profs.append(c.astype(np.uint16))
    return seqs, profs
