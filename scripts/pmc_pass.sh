#!/bin/bash
# Diagnostic: one rocprofv3 counter pass over the default bench (counters given as arguments), summed per kernel.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_tmp -- python bench.py --steps 1 --warmup 1 --no-cpu --no-extras --streams 1 > gpurun_out/pmc_tmp.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_tmp/**/*counter_collection.csv", recursive=True)
if not f:
    print(open("gpurun_out/pmc_tmp.log").read()[-2000:]); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k in acc:
    if not any(x in k for x in ("find_wall", "classify_rel", "classify_unrel", "find_rel", "scan_cand", "paint", "order_by")): continue
    print(k)
    for c, v in acc[k].items():
        print("    %-28s %16.0f per launch" % (c, v / n[(k, c)]))
PY
