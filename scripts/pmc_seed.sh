#!/bin/bash
# Diagnostic: one rocprofv3 counter pass over scripts/seed_bench.py (counters given as arguments), per launch of k_find_seeds.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_seed
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_seed -- python scripts/seed_bench.py > gpurun_out/pmc_seed.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_seed/**/*counter_collection.csv", recursive=True)
if not f:
    print(open("gpurun_out/pmc_seed.log").read()[-2000:]); raise SystemExit
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    if not r["Kernel_Name"].startswith("k_find_seeds"): continue
    acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for c, v in acc.items():
    print("    %-28s %16.0f per launch" % (c, v / n[c]))
PY
