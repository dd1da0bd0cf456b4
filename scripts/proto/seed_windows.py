#!/usr/bin/env python3
"""Prototype (CPU, plain Python): the window counts of the seed selections (seed.c:218-324 / :694-810) WITHOUT the
sequential deque -- every segment's value from window-bounded searches and prefix operations -- checked against a
direct simulation of the reference's loop on random segment sequences (both selections' rules, skipped segments, ties,
small windows).  This is what cp_seed_wave.h implements wave-parallel; the derivation is in DESIGN.md section 9.6.

    python scripts/proto/seed_windows.py [trials]
"""
import random
import sys


def simulate(b, c, plen, W, rep):
    """the reference's loop, literally.  b[i] = begin of segment i, c[i] = its count (-1: skipped), e[i] = b[i+1] / plen"""
    N = len(b)
    e = [b[i + 1] if i + 1 < N else plen for i in range(N)]
    nw = [None] * N
    Q = []                                   # entries: segment ids
    last_oor, last_oor_pos = False, 0
    beats = (lambda x, y: x < y) if rep else (lambda x, y: x > y)
    standin = (lambda v: max(W - v, 0)) if rep else (lambda v: v)
    for i in range(N):
        if c[i] >= 0:
            if Q:
                f = Q[0]
                if beats(c[i], c[f]):
                    last_oor = False
                    for q in Q:
                        nw[q] = min(b[i] - b[q], W) if c[q] == c[f] else standin(c[q])
                    Q = []
            while Q and beats(c[i], c[Q[-1]]):
                nw[Q[-1]] = standin(c[Q[-1]])
                Q.pop()
            Q.append(i)
        if not Q:
            continue
        while Q and b[Q[0]] <= b[i] - W:
            f = Q[0]
            nw[f] = min(b[f] - last_oor_pos + 1, W) if last_oor else W
            if len(Q) > 1 and beats(c[f], c[Q[1]]):
                last_oor_pos = e[f]
            Q.pop(0)
            last_oor = True
    while Q:
        f = Q[0]
        nw[f] = min(b[f] - last_oor_pos + 1, W) if last_oor else W
        if len(Q) > 1 and c[f] > c[Q[1]]:       # both selections compare with `>` here
            last_oor_pos = e[f]
        Q.pop(0)
        last_oor = True
    return nw


def closed_form(b, c, plen, W, rep):
    N = len(b)
    bb = b + [plen]
    beats = (lambda x, y: x < y) if rep else (lambda x, y: x > y)
    standin = (lambda v: max(W - v, 0)) if rep else (lambda v: v)
    nw = [None] * N
    is_exp, flag, x_of, wipe = [False] * N, [False] * N, [N] * N, [False] * N
    for i in range(N):
        if c[i] < 0:
            continue
        # forward: up to and including the first segment that begins at or beyond b[i]+W
        g, eq, nonempty, x = -1, False, False, N
        j = i + 1
        while j < N:
            last = b[j] >= b[i] + W
            if c[j] >= 0:
                if beats(c[j], c[i]):
                    g = j
                    break
                nonempty = True
                if c[j] == c[i]:
                    eq = True
            if last:
                x = j
                break
            j += 1
        # backward: the segments still in the deque when segment i arrives (begin > b[i-1]-W)
        blocked, back_nonempty = False, False
        p_b = None                              # begin of the nearest strictly better segment
        if i > 0:
            lim = b[i - 1] - W
            j = i - 1
            while j >= 0 and b[j] > lim:
                if c[j] >= 0:
                    back_nonempty = True
                    if not beats(c[i], c[j]):
                        blocked = True
                    if beats(c[j], c[i]):
                        p_b = b[j]
                        break
                j -= 1
        wipe[i] = back_nonempty and not blocked
        if g >= 0:                              # beaten at step g
            if p_b is not None and p_b > bb[g - 1] - W:
                nw[i] = standin(c[i])
            else:
                nw[i] = min(b[g] - b[i], W)
        else:
            is_exp[i] = True
            x_of[i] = x
            flag[i] = nonempty and not eq and not (rep and x == N)
    # prefix operations over the segments in order
    last_exp_x, have_exp, last_wipe, pos = None, False, -1, 0
    for i in range(N):
        if wipe[i]:
            last_wipe = i
        if is_exp[i]:
            oor = have_exp and not (last_wipe > last_exp_x)
            nw[i] = min(b[i] - pos + 1, W) if oor else W
            have_exp, last_exp_x = True, x_of[i]
            if flag[i]:
                pos = bb[i + 1]
    return nw


def closed_form_compact(b, c, plen, W, rep):
    """the same on the VALID segments alone (what the device does): a valid segment carries its begin, its end (= the
    begin of the next segment of either kind) and its key; the begin of its predecessor of either kind follows from its
    valid predecessor (two skipped stretches are never adjacent); every index comparison of closed_form becomes a
    comparison of begins."""
    N = len(b)
    bb = b + [plen]
    key = (lambda v: 32767 - v) if rep else (lambda v: v)
    standin = (lambda v: max(W - v, 0)) if rep else (lambda v: v)
    V = [i for i in range(N) if c[i] >= 0]
    vb, ve, vk = [b[i] for i in V], [bb[i + 1] for i in V], [key(c[i]) for i in V]
    nv = len(V)
    b_last = b[N - 1] if N else 0
    NONE = -10 ** 9

    def pb(v):                                  # begin of the predecessor segment of either kind, NONE if there is none
        if v > 0:
            return vb[v - 1] if ve[v - 1] == vb[v] else ve[v - 1]
        return 0 if vb[0] > 0 else NONE
    out = [None] * N
    is_exp, flag, wipe, nwv = [False] * nv, [False] * nv, [False] * nv, [None] * nv
    for v in range(nv):
        g, eq, nonempty = -1, False, False
        j = v + 1
        while j < nv and pb(j) < vb[v] + W:      # j <= x(v)
            if vk[j] > vk[v]:
                g = j
                break
            nonempty = True
            if vk[j] == vk[v]:
                eq = True
            j += 1
        blocked, back_nonempty, p_b = False, False, None
        if pb(v) != NONE:
            lim = pb(v) - W
            j = v - 1
            while j >= 0 and vb[j] > lim:
                back_nonempty = True
                if vk[j] >= vk[v]:
                    blocked = True
                if vk[j] > vk[v]:
                    p_b = vb[j]
                    break
                j -= 1
        wipe[v] = back_nonempty and not blocked
        if g >= 0:
            nwv[v] = standin(c[V[v]]) if (p_b is not None and p_b > pb(g) - W) else min(vb[g] - vb[v], W)
        else:
            is_exp[v] = True
            endflush = b_last < vb[v] + W
            flag[v] = nonempty and not eq and not (rep and endflush)
    have, last_exp_b, last_wipe_pb, pos = False, 0, None, 0
    for v in range(nv):
        if wipe[v]:
            last_wipe_pb = pb(v)
        if is_exp[v]:
            oor = have and not (last_wipe_pb is not None and last_wipe_pb >= last_exp_b + W)
            nwv[v] = min(vb[v] - pos + 1, W) if oor else W
            have, last_exp_b = True, vb[v]
            if flag[v]:
                pos = ve[v]
    for v in range(nv):
        out[V[v]] = nwv[v]
    return out


def device_like(c):
    """the device's segment sequences never hold two skipped stretches in a row"""
    return all(not (c[i] < 0 and c[i + 1] < 0) for i in range(len(c) - 1))


def trial(rng, N, W, rep):
    b, p = [], 0
    for _ in range(N):
        b.append(p)
        p += rng.choice([1, 1, 2, 3, 5, 8, 13, W // 3 + 1, W, 2 * W + 3])
    plen = p
    hi = rng.choice([2, 3, 5, 40, 2000])
    c = [(-1 if rng.random() < rng.choice([0.0, 0.1, 0.5]) else rng.randrange(hi)) for _ in range(N)]
    if rng.random() < 0.7:                       # mostly sequences as the device makes them: no two skipped stretches in a row
        for i in range(1, N):
            if c[i] < 0 and c[i - 1] < 0:
                c[i] = rng.randrange(hi)
    a, z = simulate(b, c, plen, W, rep), closed_form(b, c, plen, W, rep)
    if a == z and device_like(c):
        z = closed_form_compact(b, c, plen, W, rep)
        trial.compact += 1
    if a != z:
        k = next(i for i in range(N) if a[i] != z[i])
        raise SystemExit("MISMATCH at segment %d: reference %r, closed form %r\n b=%r\n c=%r\n W=%d rep=%r" % (k, a[k], z[k], b, c, W, rep))


trial.compact = 0


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rng = random.Random(12345)
    for t in range(n):
        trial(rng, rng.choice([1, 2, 3, 5, 8, 13, 30, 80]), rng.choice([1, 2, 4, 10, 50, 200, 1000]), rng.random() < 0.5)
    print("closed form == reference loop on %d random segment sequences (%d of them also through the compacted form)" % (n, trial.compact))
