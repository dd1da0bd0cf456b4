"""Seeded synthetic HiFi read sets with k-mer count profiles.

FastK and the MHC download are unavailable offline, so inputs are synthesised:

  genome   uniform ACGT haplotype A; a few segmental repeats (copies of 3 kb segments);
           haplotype B = A with SNPs at `het` rate outside one homozygous block
  reads    length ~ N(r, (0.2 r)^2) clipped to [min_len, 60000], random haplotype / strand,
           HiFi-like errors: substitutions, and 1-base indels biased to homopolymers
  profile  count of a read k-mer = number of error-free read k-mers (all reads, both strands)
           whose genomic k-mer has the same sequence (classes found by hashing all 2G genomic
           k-mers); k-mers overlapping a read error get count 1.  O(total bases).

Everything is numpy; `make_dataset` is deterministic in `seed`.
"""
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.array([3, 2, 1, 0], dtype=np.uint8)


def _kmer_classes(codes, K, rng):
    """Class id per k-mer start of the concatenated haplotypes `codes` (list of uint8 arrays)."""
    B = np.uint64(0x9E3779B97F4A7C15)           # odd multiplier, wraps mod 2^64
    Binv = np.uint64(pow(int(B), -1, 1 << 64))
    table = rng.integers(1, 1 << 63, size=4, dtype=np.uint64) | np.uint64(1)
    hashes = []
    with np.errstate(over="ignore"):
        for c in codes:
            n = len(c)
            pw = np.empty(n + 1, np.uint64)
            pw[0] = 1
            pw[1:] = Binv
            pw = np.cumprod(pw, dtype=np.uint64)            # Binv^i
            s = np.zeros(n + 1, np.uint64)
            np.cumsum(table[c] * pw[:n], dtype=np.uint64, out=s[1:])
            fw = np.empty(n + 1, np.uint64)
            fw[0] = 1
            fw[1:] = B
            fw = np.cumprod(fw, dtype=np.uint64)            # B^i
            h = (s[K:] - s[:n - K + 1]) * fw[:n - K + 1]
            hashes.append(h)
    allh = np.concatenate(hashes)
    _, inv = np.unique(allh, return_inverse=True)
    out, o = [], 0
    for h in hashes:
        out.append(inv[o:o + len(h)].astype(np.int64))
        o += len(h)
    return out, int(inv.max()) + 1


def make_dataset(genome_len=200_000, cov=40, read_len=10_000, K=40, het=0.002, err_sub=0.0006,
                 err_indel=0.0006, n_repeats=3, repeat_len=3000, homo_block=8000, min_len=2000,
                 seed=1, max_len=60000):
    """Returns dict(seqs=[bytes], profiles=[uint16 arrays], hist=(low, high, ilow, ihigh, int64[high-low+1]),
    names=[str], K=K).  `cov` is total (diploid) read coverage."""
    rng = np.random.default_rng(seed)
    G = genome_len
    A = rng.integers(0, 4, size=G, dtype=np.uint8)
    for _ in range(n_repeats):                               # segmental repeats: 2-4 extra copies
        L = min(repeat_len, G // 8)
        src = int(rng.integers(0, G - L))
        for _c in range(int(rng.integers(2, 5))):
            dst = int(rng.integers(0, G - L))
            A[dst:dst + L] = A[src:src + L]
    for _ in range(max(1, G // 20000)):                      # a few long homopolymers / microsatellites
        p = int(rng.integers(0, G - 64))
        kind = int(rng.integers(0, 3))
        unit = rng.integers(0, 4, size=kind + 1, dtype=np.uint8)
        reps = int(rng.integers(6, 16))
        seg = np.tile(unit, reps)
        A[p:p + len(seg)] = seg
    Bh = A.copy()
    snp = rng.random(G) < het
    if homo_block > 0 and G > 4 * homo_block:
        hb = int(rng.integers(0, G - homo_block))
        snp[hb:hb + homo_block] = False
    Bh[snp] = (Bh[snp] + rng.integers(1, 4, size=int(snp.sum()), dtype=np.uint8)) & 3
    haps = [A, Bh]
    classes, ncls = _kmer_classes(haps, K, rng)

    n_reads = max(1, int(cov * G / read_len))
    reads = []          # (codes, cls per k-mer (-1 = error), strand)
    cover = np.zeros(ncls, np.int64)
    for _r in range(n_reads):
        L = int(np.clip(rng.normal(read_len, 0.2 * read_len), min_len, min(max_len, G)))
        h = int(rng.integers(0, 2))
        s = int(rng.integers(0, G - L + 1))
        g = haps[h][s:s + L]
        coord = np.arange(s, s + L, dtype=np.int64)
        codes = g.copy()
        # substitutions
        m = rng.random(L) < err_sub
        if m.any():
            codes[m] = (codes[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
            coord[m] = -1
        # 1-base indels, 5x more likely inside homopolymers (the HiFi error mode)
        hp = np.zeros(L, bool)
        hp[1:] = g[1:] == g[:-1]
        pind = np.where(hp, 5.0, 1.0) * err_indel
        ev = np.nonzero(rng.random(L) < pind)[0]
        if len(ev):
            keep = np.ones(L, bool)
            ins_at = []
            for p in ev:
                if rng.random() < 0.5:
                    keep[p] = False                     # deletion
                else:
                    ins_at.append(p)                    # insertion (duplicate base p)
            codes_l, coord_l = codes[keep], coord[keep]
            if ins_at:
                # positions in the kept array
                idx = np.searchsorted(np.nonzero(keep)[0], np.array(ins_at))
                idx = idx[idx < len(codes_l)]
                codes_l = np.insert(codes_l, idx, codes_l[idx])
                coord_l = np.insert(coord_l, idx, -1)
            codes, coord = codes_l, coord_l
        L2 = len(codes)
        if L2 < K:
            continue
        # clean k-mer: no edited base inside, and genome coordinates contiguous
        bad = (coord < 0).astype(np.int64)
        junc = np.zeros(L2, np.int64)
        junc[1:] = ((coord[1:] != coord[:-1] + 1) | (coord[1:] < 0) | (coord[:-1] < 0)).astype(np.int64)
        cb = np.concatenate(([0], np.cumsum(bad)))
        cj = np.concatenate(([0], np.cumsum(junc)))
        nk = L2 - K + 1
        q = np.arange(nk)
        clean = ((cb[q + K] - cb[q]) == 0) & ((cj[q + K] - cj[q + 1]) == 0)
        cls = np.full(nk, -1, np.int64)
        cls[clean] = classes[h][coord[q[clean]]]
        np.add.at(cover, cls[clean], 1)
        strand = int(rng.integers(0, 2))
        reads.append((codes, cls, strand))

    # genomic multiplicity of every k-mer class over both haplotypes: the "relative profile" the
    # reference's evaluation derives its ground truth from (prof2class.c: 0->E, 1->H, 2->D, >=3->R)
    mult = np.bincount(np.concatenate([c.ravel() for c in classes]), minlength=ncls)
    seqs, profiles, names, rel_profiles = [], [], [], []
    n_err_kmers = 0
    for i, (codes, cls, strand) in enumerate(reads):
        prof = np.ones(len(cls), np.int64)
        ok = cls >= 0
        prof[ok] = cover[cls[ok]]
        n_err_kmers += int((~ok).sum())
        prof = np.minimum(prof, 32767).astype(np.uint16)
        rel = np.zeros(len(cls), np.int64)
        rel[ok] = mult[cls[ok]]
        rel = np.minimum(rel, 32767).astype(np.uint16)
        if strand:
            codes = _COMP[codes[::-1]]
            prof = prof[::-1].copy()
            rel = rel[::-1].copy()
        seqs.append(BASES[codes].tobytes())
        profiles.append(prof)
        rel_profiles.append(rel)
        names.append("read%d" % (i + 1))

    low, high = 1, 32767
    hist = np.zeros(high - low + 1, np.int64)                # unique counts: #distinct k-mers per count
    cc = np.bincount(np.minimum(cover[cover > 0], high), minlength=high + 1)
    hist[:] = cc[low:high + 1]
    hist[0] += n_err_kmers
    return dict(seqs=seqs, profiles=profiles, names=names, K=K, rel_profiles=rel_profiles,
                hist=(low, high, 0, 0, hist), genome_len=G, read_len=read_len)


def pack_batch(seqs, profiles):
    """Concatenate reads into the flat layout the C-ABI takes: seq bytes, seq_off, prof u16, prof_off."""
    n = len(seqs)
    seq_off = np.zeros(n + 1, np.int64)
    prof_off = np.zeros(n + 1, np.int64)
    for i in range(n):
        seq_off[i + 1] = seq_off[i] + len(seqs[i])
        prof_off[i + 1] = prof_off[i] + len(profiles[i])
    seq = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    prof = np.concatenate(profiles).astype(np.uint16) if n else np.zeros(0, np.uint16)
    return seq, seq_off, prof, prof_off


def write_himodel(path, kmer=40, seed=7, base=(0.004, 0.003, 0.0025), growth=(0.0016, 0.0022, 0.003)):
    """A synthetic HIsim error-model file in the layout load_himodel reads (wall.c:43-84): int kmer,
    0x4000 heptamer records (11 floats), then per unit length 1..3 a 4^ulen x (kmer/2-6) table of
    7-float micro-satellite records whose first float is the error rate of `2*ulen+c` bases of that
    unit.  Rates grow quadratically with the number of unit copies, with per-unit noise and some
    empty (0) cells, which the loader must skip.  Returns the per-type mean rates y[2..5] it wrote."""
    rng = np.random.default_rng(seed)
    krange = kmer // 2 - 6
    ys = []
    with open(path, "wb") as f:
        f.write(np.int32(kmer).tobytes())
        f.write(rng.random((0x4000, 11), dtype=np.float32).tobytes())
        for t in range(3):
            ulen = t + 1
            N = 1 << (2 * ulen)
            tab = np.zeros((N, krange, 7), np.float32)
            for c in range(krange):
                copies = (2 * ulen + c) / ulen
                rate = base[t] + growth[t] * copies * copies
                tab[:, c, 0] = (rate * (1 + 0.2 * rng.standard_normal(N))).clip(1e-5, 0.9)
                tab[:, c, 1:] = rng.random((N, 6), dtype=np.float32) * 0.01
            tab[rng.random((N, krange)) < 0.1, 0] = 0.0            # unobserved cells
            f.write(tab.tobytes())
            y = []
            for j in range(2, 6):
                col = tab[:, (j - 2) * ulen, 0].astype(np.float64)
                y.append(col[col > 0].sum() / (col > 0).sum())
            ys.append(y)
    return np.array(ys)
