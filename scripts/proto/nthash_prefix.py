#!/usr/bin/env python3
"""Prototype (CPU, plain Python): the canonical ntHash of every k-mer of a stretch of bases from TWO prefix-XOR arrays
instead of a K-step fold per k-mer -- the restatement DESIGN.md section 8.7 names for the marks of the seed kernel
(cp_seed_wave.h hashes a k-mer by K look-ups in a table of pre-rotated seeds today: 40 LDS reads + XORs each way).

ntHash (nthash.h:181-235) folds   fh = srol(fh) ^ seed[b],   rh = srol(rh) ^ seed[comp b'] (b' from the k-mer's end);
srol = rotate left by one + swap of bits 0 and 33, i.e. the low 33 bits and the high 31 bits each rotate among
themselves: a group action of order 33 * 31, srol^m = (rot33 by m mod 33, rot31 by m mod 31), so srol^-m exists and

    fh(i) = XOR_{j<K} srol^(K-1-j)(seed[b(i+j)])       = srol^(i+K-1) ( P(i+K) ^ P(i) ),   P(n) = XOR_{t<n} srol^-t (seed[b(t)])
    rh(i) = XOR_{j<K} srol^j      (seed[comp b(i+j)])  = srol^-i      ( Q(i+K) ^ Q(i) ),   Q(n) = XOR_{t<n} srol^t  (seed[comp b(t)])

Per k-mer: two reads of P, two of Q, four split rotations by amounts known from its position -- whatever K is.  P and Q
are XOR scans over the bases: 64 bases a step on a wave (lane l rotates its seed by -l / +l, constants per lane; six
DPP steps of a 64-bit XOR scan; the chunk's start j0 enters as ONE wave-uniform rotation: P(j0+l+1) = P(j0) ^
srol^-j0(X(l))).  `chunked()` below is that form, 64 "lanes" at a time, checked against the literal fold on random
strings with every kind of letter nthash.h's table knows (unknown letters hash as 0, as in the reference).

    python scripts/proto/nthash_prefix.py [trials=300]
"""
import random
import sys

M64 = (1 << 64) - 1
SEED = {ord(c): v for cs, v in (("Aa\x04\x05", 0x3c8bfbb395c60474), ("Cc\x07", 0x3193c18562a02b4c),
                                  ("Gg\x03", 0x20323ed082572324), ("TtUu\x01", 0x295549f54be24456)) for c in cs}   # nthash.h:26-59
MOD = 2147483647                                                                                                    # seed.c:26


def seed_fw(c):
    return SEED.get(c, 0)


def seed_rc(c):                                  # seedTab[c & cpOff], nthash.h:17
    return SEED.get(c & 7, 0)


def srol1(v):                                    # nthash.h:181-207, literally
    v = ((v << 1) | (v >> 63)) & M64
    x = (v ^ (v >> 33)) & 1
    return v ^ (x | (x << 33))


def fold(seq, j, K):                             # nthash.h:215-235 (what seed.c:28-55 stores, mod 2^31-1)
    fh = rh = 0
    for i in range(K):
        fh = srol1(fh) ^ seed_fw(seq[j + i])
        rh = srol1(rh) ^ seed_rc(seq[j + K - 1 - i])
    return min(fh, rh) % MOD


def srol(v, m):                                  # srol^m for any integer m: the two parts rotate on their own
    lo, hi = v & ((1 << 33) - 1), v >> 33
    a, b = m % 33, m % 31
    lo = ((lo << a) | (lo >> (33 - a))) & ((1 << 33) - 1)
    hi = ((hi << b) | (hi >> (31 - b))) & ((1 << 31) - 1)
    return lo | (hi << 33)


def chunked(seq, K, lanes=64):
    """hashes of all k-mers of seq; the scans 64 positions at a time with a carried prefix"""
    n = len(seq)
    P, Q = [0] * (n + 1), [0] * (n + 1)
    for j0 in range(0, n, lanes):
        w = min(lanes, n - j0)
        xf = [srol(seed_fw(seq[j0 + l]), -l) for l in range(w)]          # per-lane constants -l / +l
        xr = [srol(seed_rc(seq[j0 + l]), l) for l in range(w)]
        d = 1
        while d < lanes:                                                 # the wave's inclusive XOR scan (Hillis-Steele = row_shr / bcast DPP steps)
            xf = [xf[l] ^ (xf[l - d] if l >= d else 0) for l in range(w)]
            xr = [xr[l] ^ (xr[l - d] if l >= d else 0) for l in range(w)]
            d *= 2
        for l in range(w):                                               # the chunk's start as one uniform rotation
            P[j0 + l + 1] = P[j0] ^ srol(xf[l], -j0)
            Q[j0 + l + 1] = Q[j0] ^ srol(xr[l], j0)
    out = []
    for i in range(n - K + 1):
        fh = srol(P[i + K] ^ P[i], i + K - 1)
        rh = srol(Q[i + K] ^ Q[i], -i)
        out.append(min(fh, rh) % MOD)
    return out


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = random.Random(7)
    letters = b"ACGTacgtNnUuRYKM\x01\x03\x04\x05\x07\x00"
    for m in range(0, 2100, 37):                                         # srol^m against m single steps, and its inverse
        v = rng.getrandbits(64)
        w = v
        for _ in range(m):
            w = srol1(w)
        assert srol(v, m) == w and srol(w, -m) == v
    tot = 0
    for t in range(trials):
        K = rng.choice((15, 21, 31, 33, 40, 63, 64, 70))
        n = rng.randrange(K, K + 400)
        p_odd = rng.choice((0.0, 0.02, 0.3))
        seq = bytes(rng.choice(letters) if rng.random() < p_odd else rng.choice(b"ACGT") for _ in range(n))
        got = chunked(seq, K)
        for i in range(n - K + 1):
            assert got[i] == fold(seq, i, K), (t, K, n, i)
        tot += n - K + 1
    print("nthash_prefix: %d k-mers over %d strings (K 15-70, odd letters, chunks of 64): prefix form == fold" % (tot, trials))


if __name__ == "__main__":
    main()
