#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_runs2; rm -rf $O; mkdir -p $O
for i in 1 2; do
python bench.py --steps 2 --warmup 1 --no-cpu --only-pcie > $O/p$i.json 2> $O/p$i.err; python - $i <<'PY'
import json,sys
try:
    j=json.loads(open("gpurun_out/r4_runs2/p%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); p=j["extras"].get("pcie") or j["extras"]
    print({k:v for k,v in p.items() if k!="label_runs"}); print("RUNS", p.get("label_runs"))
except Exception as e:
    print("failed", e, open("gpurun_out/r4_runs2/p%s.err"%sys.argv[1]).read()[-800:])
PY
done
