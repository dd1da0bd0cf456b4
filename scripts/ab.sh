#!/bin/bash
# A/B of library variants on the default bench inside ONE gpurun call (box-to-box spread is larger than most effects):
#   bash scripts/ab.sh [rounds] item...     item = lib | lib,ENV=VAL[,ENV=VAL...]    ("default" = the in-tree library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=$1; shift
mkdir -p gpurun_out/ab
for i in $(seq 1 $R); do
  for item in "$@"; do
    IFS=',' read -ra parts <<< "$item"
    l=${parts[0]}
    envs=("${parts[@]:1}")
    if [ "$l" = default ]; then libenv=(); else libenv=("CLASSPRO_AMD_LIB=$GRAFT_REPO_ROOT/$l"); fi
    env "${libenv[@]}" "${envs[@]}" python bench.py --no-cpu --no-extras --steps 10 --warmup 3 > gpurun_out/ab/out.json 2> gpurun_out/ab/err.txt
    python - "$item" <<'PY'
import json,sys
try:
    j=json.loads(open("gpurun_out/ab/out.json").read().strip().splitlines()[-1]); print("%-44s %9.1f Mbases/s  %7.3f ms/step  scan %.1f us frac %.3f" % (sys.argv[1], j["value"], j["ms_per_step"], j["roofline"]["ms_per_launch"]*1e3, j["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("gpurun_out/ab/err.txt").read()[-400:])
PY
  done
done
