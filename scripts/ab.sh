#!/bin/bash
# A/B of library variants on the default bench inside ONE gpurun call (box-to-box spread is larger than most effects):
#   bash scripts/ab.sh [rounds] lib...      ("default" = the in-tree library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=$1; shift
mkdir -p gpurun_out/ab
for i in $(seq 1 $R); do
  for l in "$@"; do
    if [ "$l" = default ]; then unset CLASSPRO_AMD_LIB; else export CLASSPRO_AMD_LIB=$GRAFT_REPO_ROOT/$l; fi
    python bench.py --no-cpu --no-extras --steps 10 --warmup 3 > gpurun_out/ab/out.json 2> gpurun_out/ab/err.txt
    python - "$l" <<'PY'
import json,sys
j=json.loads(open("gpurun_out/ab/out.json").read().strip().splitlines()[-1]); print("%-28s %9.1f Mbases/s  %7.3f ms/step  scan %.1f us frac %.3f" % (sys.argv[1], j["value"], j["ms_per_step"], j["roofline"]["ms_per_launch"]*1e3, j["roofline"]["frac"]))
PY
  done
done
