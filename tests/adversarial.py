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
