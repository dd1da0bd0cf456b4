#!/bin/bash
# Diagnostic: rocprofv3 kernel stats of the default bench for the library given (or the product build).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-cur}
rm -rf gpurun_out/ks_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$TAG -- python bench.py --steps 5 --warmup 2 --no-cpu $KSTATS_ARGS > gpurun_out/ks_$TAG.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/ks_$TAG/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]: print("%-62s %4s %10.1f us" % (r["Name"][:62], r["Calls"], float(r["AverageNs"])/1e3))
PY
