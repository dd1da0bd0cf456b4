#!/usr/bin/env python3
"""Copy the judged rocprofv3 summaries of a round from gpurun_out/prof_<tag>/ into profiles/.

    python scripts/summarize_profile.py r01a r01

writes profiles/<name>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the default bench.py run),
profiles/<name>_pmc.txt (FETCH_SIZE / WRITE_SIZE per kernel, separate --pmc passes),
profiles/<name>_bench.json and profiles/scan_pmc.json (HBM bytes per position of k_scan_candidates,
corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE is in KiB and counts 1/2 of the bytes of a
16-B-per-lane streaming read on gfx950 -> bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 as is).
"""
import collections
import csv
import glob


def newest(pattern):
    """gpurun merges every call's files into the same directory: take the latest run's"""
    return max(glob.glob(pattern), key=os.path.getmtime)

import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), os.path.join(dst, name + "_kernel_stats.csv"))
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, name + "_bench.json"), "w"), indent=1)
kmers = bench["roofline"]["algorithmic_bytes_per_launch"] / 2.0
sys.path.insert(0, ROOT)
import bench as _bench_mod
kernel_id = _bench_mod.scan_kernel_id()

agg = collections.defaultdict(list)
grid = {}
for kind in ("fetch", "write"):
    for r in csv.DictReader(open(newest(os.path.join(src, kind, "*", "*_counter_collection.csv")))):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("void "):                          # template kernels: "void k_classify_rel_grp<0, 128, 4>"
            k = k[5:]
        agg[(k.replace(" ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(os.path.join(dst, name + "_pmc.txt"), "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 1 --warmup 1 --no-cpu --no-extras`\n")
    f.write("# values in KiB per dispatch as reported; gfx950 correction for streaming reads: bytes = 2*FETCH_SIZE*1024\n")
    f.write("%-36s %-11s %6s %14s %14s %14s\n" % ("kernel", "counter", "n", "avg_KiB", "min_KiB", "max_KiB"))
    tot = collections.Counter()
    nsub = max(1, len(agg[("k_find_wall", "FETCH_SIZE")]))      # sub-batches in the run (the scan also runs in the roofline loop)
    # scan launches per sub-batch: a sub-batch is scanned in launches of at most 2^31 positions (capi.hip: launch_scan)
    sub_kmers = bench["roofline"]["working_set_bytes"] / 2.0 / max(1, bench["extras"].get("sub_batches_per_rank", 1))
    scan_per_sub = max(1, int(round(sub_kmers / kmers)))
    for (k, c), v in sorted(agg.items()):
        if k.startswith("k_") and "_table" not in k and not k.startswith("k_sg_"):
            f.write("%-36s %-11s %6d %14.1f %14.1f %14.1f\n" % (k, c, len(v), sum(v) / len(v), min(v), max(v)))
            tot[c] += scan_per_sub * sum(v) / len(v) if k == "k_scan_candidates" else sum(v) / nsub
    f.write("# sum over the product kernels per sub-batch (%d sub-batches in the run, %d scan launches each; only the scan's FETCH_SIZE is\n" % (nsub, scan_per_sub))
    f.write("# doubled: the x2 is calibrated for 16-B-per-lane streaming reads, the other kernels' access widths are not):\n")
    scan_f = scan_per_sub * sum(agg[("k_scan_candidates", "FETCH_SIZE")]) / len(agg[("k_scan_candidates", "FETCH_SIZE")])
    f.write("#   FETCH_SIZE %.0f KiB raw, %.0f KiB with the scan doubled; WRITE_SIZE %.0f KiB\n" % (tot["FETCH_SIZE"], tot["FETCH_SIZE"] + scan_f, tot["WRITE_SIZE"]))
    f.write("#   in bytes (x 1024): %.2f GB fetched + %.2f GB written = %.2f GB per sub-batch\n" % ((tot["FETCH_SIZE"] + scan_f) * 1024 / 1e9, tot["WRITE_SIZE"] * 1024 / 1e9, (tot["FETCH_SIZE"] + scan_f + tot["WRITE_SIZE"]) * 1024 / 1e9))
fs = agg[("k_scan_candidates", "FETCH_SIZE")]
wsz = agg[("k_scan_candidates", "WRITE_SIZE")]
fetch = 2.0 * 1024 * sum(fs) / len(fs)
write = 1024.0 * sum(wsz) / len(wsz)
json.dump({"kernel": "k_scan_candidates", "kernel_id": kernel_id, "positions_per_launch": kmers,
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 1 --warmup 1 --no-cpu --no-extras (scripts/profile_round.sh)",
           "fetch_bytes_per_launch_corrected": fetch, "write_bytes_per_launch": write,
           "hbm_bytes_per_position": (fetch + write) / kmers,
           "source": "profiles/%s_pmc.txt (FETCH_SIZE x2 per MI355X_MICROARCH.md HBM section)" % name},
          open(os.path.join(dst, "scan_pmc.json"), "w"), indent=1)
print(open(os.path.join(dst, name + "_pmc.txt")).read())
print(open(os.path.join(dst, "scan_pmc.json")).read())

# The default bench keeps two sub-batches in flight, so the scan launches of the pipeline overlap with other kernels; the
# launches of bench.py's roofline loop (the last ones, back-to-back on one stream) are the ones its HIP events time.
tr = newest(os.path.join(src, "stats", "*", "*_kernel_trace.csv"))
rows = [r for r in csv.DictReader(open(tr)) if r["Kernel_Name"].startswith("k_scan_candidates")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
nloop = bench["roofline"].get("launches_timed", 200)
loop = d[-nloop:]
with open(os.path.join(dst, name + "_scan_durations.txt"), "w") as f:
    f.write("# k_scan_candidates launch durations in the rocprofv3 --kernel-trace of the stats run; the last %d launches are bench.py's\n" % nloop)
    f.write("# roofline loop (back-to-back on one stream), the earlier ones belong to the pipeline and overlap with the other stream's kernels\n")
    f.write("all %d launches: average %.1f us (the figure in %s_kernel_stats.csv)\n" % (len(d), sum(d) / len(d), name))
    f.write("roofline-loop launches (%d): average %.1f us, min %.1f, max %.1f\n" % (len(loop), sum(loop) / len(loop), min(loop), max(loop)))
    f.write("bench.py HIP events, separate unprofiled run (%s_bench.json): %.1f us per launch, %.0f GB/s\n" % (name, bench["roofline"]["ms_per_launch"] * 1e3, bench["roofline"]["achieved"]))
    same = [json.loads(l) for l in open(os.path.join(src, "stats.log")) if l.startswith("{")]
    if same:                                   # the bench line printed by the traced run itself: events and trace of the SAME launches
        r = same[-1]["roofline"]
        f.write("bench.py HIP events of the traced run itself (stats.log): %.1f us per launch, %.0f GB/s, frac %.3f\n" % (r["ms_per_launch"] * 1e3, r["achieved"], r["frac"]))
print(open(os.path.join(dst, name + "_scan_durations.txt")).read())
