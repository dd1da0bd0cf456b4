#!/bin/bash
# bench-argument A/B inside ONE gpurun call (sub-batch size and stream count at the end of round 5, after the kernels got faster):
#   bash scripts/r5_args_ab.sh "<args>" "<args>" ...      ("" = the defaults)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/args_ab
for r in 1 2; do
  for args in "$@"; do
    python bench.py --no-cpu --no-extras --steps 10 --warmup 3 $args > gpurun_out/args_ab/out.json 2> gpurun_out/args_ab/err.txt
    python - "$args" <<'PY'
import json,sys
try:
    j=json.loads(open("gpurun_out/args_ab/out.json").read().strip().splitlines()[-1]); print("%-44s %9.1f Mbases/s  %7.3f ms/step  sub-batches %d streams %d" % (sys.argv[1] or "(defaults)", j["value"], j["ms_per_step"], j["extras"]["sub_batches_per_rank"], j["extras"]["streams"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("gpurun_out/args_ab/err.txt").read()[-300:])
PY
  done
done
