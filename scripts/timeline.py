#!/usr/bin/env python3
"""Reduce a rocprofv3 --kernel-trace of `bench.py` to a per-stream timeline of the classification pipeline.

    python scripts/timeline.py <..._kernel_trace.csv> [out.txt]

A sub-batch = one cp_classify_batch call = the kernels between a k_scan_candidates that is followed by k_count_caps
on the same stream and that call's k_paint_labels (the scan-only launches of bench.py's roofline loop are left out).
Reported, over the sub-batches of the timed steps (the first `--skip` per stream are warm-up and are dropped):
  * per kernel: start and end relative to the sub-batch's scan start, duration, idle gap on the stream before it
  * the serial head: scan start -> end of k_prefix_caps, the host round trip (end of k_prefix_caps -> start of the
    next kernel on the stream: D2H of the totals, host wake-up, scratch sizing, launch), scan start -> first wide
    kernel after it (k_wall_tasks)
  * sub-batch latency (scan start -> paint end) and period (distance between consecutive scan starts, all streams)
  * how many product kernels run at once: fraction of the timed span with 0, 1, 2, ... kernels in flight
"""
import collections
import csv
import sys


def short(name):
    n = name.split("(")[0]
    if n.startswith("void "):
        n = n[5:]
    return n.replace(" ", "")


def main():
    path = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else sys.stdout
    skip = 6
    for a in sys.argv[2:]:
        if a.startswith("--skip="):
            skip = int(a.split("=")[1])
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], short(r["Kernel_Name"])))
    rows.sort()
    by_stream = collections.defaultdict(list)
    for r in rows:
        by_stream[r[2]].append(r)

    def product(n):
        return n.startswith("k_") and not n.startswith("k_sg_") and not n.endswith("_table")

    # A call's head (scan, count, prefix sums) may run on the workspace's own high-priority stream and its body (ordering
    # ... paint) on the caller's: heads and bodies are collected per stream, a head stream is paired with the body stream
    # whose first k_wall_tasks starts soonest after its first head ends, and the two lists are zipped in order.
    HEAD = ("k_scan_candidates", "k_count_caps", "k_prefix_caps", "k_prefix_caps_mb")
    heads, bodies = collections.defaultdict(list), collections.defaultdict(list)
    for st, lst in by_stream.items():
        i = 0
        cur = []
        while i < len(lst):
            n = lst[i][3]
            if n == "k_scan_candidates":
                j = i + 1
                while j < len(lst) and not product(lst[j][3]):
                    j += 1
                if j < len(lst) and lst[j][3] == "k_count_caps":
                    k = j
                    while k < len(lst) and not lst[k][3].startswith("k_prefix_caps"):
                        k += 1
                    if k < len(lst):
                        heads[st].append([(a, b, n2) for a, b, _, n2 in lst[i:k + 1]])
                        i = k + 1
                        cur = []
                        continue
                i += 1                                      # a scan-only launch (bench.py's roofline loop)
                continue
            if cur or product(n) or n.startswith("__amd_rocclr"):
                cur.append((lst[i][0], lst[i][1], n))
            if n == "k_paint_labels":
                if any(x[2] == "k_wall_tasks" for x in cur):
                    while cur and not product(cur[0][2]) and cur[0][2] != "__amd_rocclr_copyBuffer":
                        cur.pop(0)
                    bodies[st].append(cur)
                cur = []
            i += 1
    subs = []                                             # (stream of the body, [(start,end,name), ...])
    used = set()
    for hs in sorted(heads, key=lambda h: heads[h][0][0][0]):
        h_end = heads[hs][0][-1][1]
        best = None
        for bs in bodies:
            if bs in used:
                continue
            wt = next(x[0] for x in bodies[bs][0] if x[2] == "k_wall_tasks")
            if wt >= h_end and (best is None or wt < best[0]):
                best = (wt, bs)
        if best is None:
            continue
        used.add(best[1])
        bl, bi = bodies[best[1]], 0
        for h in heads[hs]:                                 # a head's body: the next one whose k_wall_tasks starts after the head ended
            while bi < len(bl) and next(x[0] for x in bl[bi] if x[2] == "k_wall_tasks") < h[-1][1]:
                bi += 1
            if bi == len(bl):
                break
            subs.append((best[1], h + bl[bi]))
            bi += 1
    per_stream = collections.Counter()
    kept = []
    for st, ks in sorted(subs, key=lambda x: x[1][0][0]):
        per_stream[st] += 1
        if per_stream[st] > skip:
            kept.append((st, ks))
    w = out.write
    w("# %s\n# %d sub-batches found on %d streams, %d kept after dropping the first %d per stream (warm-up)\n"
      % (path, len(subs), len(per_stream), len(kept), skip))
    if not kept:
        return
    # ---- per kernel, relative to the scan start ----
    acc = collections.OrderedDict()
    head = collections.defaultdict(list)
    for st, ks in kept:
        t0 = ks[0][0]
        prev_end = None
        seen = collections.Counter()
        for a, b, n in ks:
            if not product(n):
                n = "(" + n[:28] + ")"
            seen[n] += 1
            key = n if seen[n] == 1 else "%s#%d" % (n, seen[n])
            acc.setdefault(key, []).append(((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, (a - prev_end) / 1e3 if prev_end else 0.0))
            prev_end = b if prev_end is None else max(prev_end, b)
        names = [n for _, _, n in ks]
        pc = next((x for x in ks if x[2].startswith("k_prefix_caps")), None)
        if pc:
            head["scan start -> k_prefix_caps end"].append((pc[1] - t0) / 1e3)
            nxt = next((x for x in ks if x[0] >= pc[1] and product(x[2]) and not x[2].startswith("k_prefix_caps")), None)
            if nxt:
                head["host round trip: k_prefix_caps end -> next kernel start (%s)" % nxt[2]].append((nxt[0] - pc[1]) / 1e3)
        wt = next((x for x in ks if x[2] == "k_wall_tasks"), None)
        if wt:
            head["scan start -> k_wall_tasks start (the serial head)"].append((wt[0] - t0) / 1e3)
        head["sub-batch latency: scan start -> k_paint_labels end"].append((ks[-1][1] - t0) / 1e3)
    w("\n%-44s %5s %10s %10s %10s %10s %10s\n" % ("kernel (in launch order)", "n", "start_us", "end_us", "dur_us", "dur_max", "gap_before"))
    for k, v in acc.items():
        n = len(v)
        w("%-44s %5d %10.1f %10.1f %10.1f %10.1f %10.1f\n" % (k[:44], n, sum(x[0] for x in v) / n, sum(x[1] for x in v) / n,
                                                               sum(x[2] for x in v) / n, max(x[2] for x in v), sum(x[3] for x in v) / n))
    w("\n%-88s %5s %10s %10s %10s\n" % ("interval", "n", "avg_us", "min_us", "max_us"))
    for k, v in head.items():
        w("%-88s %5d %10.1f %10.1f %10.1f\n" % (k, len(v), sum(v) / len(v), min(v), max(v)))
    starts = sorted(ks[0][0] for _, ks in kept)
    per = [(b - a) / 1e3 for a, b in zip(starts, starts[1:])]
    span0, span1 = starts[0], max(ks[-1][1] for _, ks in kept)
    w("%-88s %5d %10.1f %10.1f %10.1f\n" % ("period: distance between consecutive scan starts (all streams)", len(per), sum(per) / len(per), min(per), max(per)))
    w("timed span %.1f ms for %d sub-batches = %.3f ms per sub-batch\n" % ((span1 - span0) / 1e6, len(kept), (span1 - span0) / 1e6 / len(kept)))
    # ---- concurrency of product kernels over the span ----
    ev = []
    busy = collections.defaultdict(float)
    for a, b, st, n in rows:                              # every product kernel inside the span, the auxiliary streams' too
        if product(n) and a >= span0 and b <= span1:
            ev.append((a, 1))
            ev.append((b, -1))
            busy[n] += (b - a)
    ev.sort()
    hist = collections.defaultdict(float)
    cur, last = 0, span0
    for t, d in ev:
        hist[cur] += t - last
        last = t
        cur += d
    tot = sum(hist.values())
    w("\nproduct kernels in flight at once (fraction of the span):\n")
    for k in sorted(hist):
        w("  %d: %5.1f %%\n" % (k, 100.0 * hist[k] / tot))
    w("sum of kernel durations / span = %.2f (average number of kernels in flight)\n" % (sum(busy.values()) / tot))
    w("\nshare of the summed kernel time:\n")
    for n, v in sorted(busy.items(), key=lambda x: -x[1]):
        w("  %-44s %5.1f %%  (%.3f ms per sub-batch)\n" % (n[:44], 100.0 * v / sum(busy.values()), v / 1e6 / len(kept)))
    # ---- idle time per stream ----
    w("\nper stream: time with no product kernel of that stream running, as a fraction of the span\n")
    for st in sorted(by_stream):
        iv = sorted((a, b) for a, b, s2, n in rows if s2 == st and product(n) and a >= span0 and b <= span1)
        if not iv:
            continue
        covered, ce = 0, iv[0][0]
        for a, b in iv:
            if b > ce:
                covered += b - max(a, ce)
                ce = b
        w("  stream %s: %.1f %% idle\n" % (st, 100.0 * (1 - covered / (iv[-1][1] - iv[0][0]))))


if __name__ == "__main__":
    main()
