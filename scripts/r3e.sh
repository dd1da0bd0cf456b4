#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3e; rm -rf $O; mkdir -p $O
run() { name=$1; shift; python bench.py --no-cpu --no-extras --steps 10 --warmup 3 "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("%-12s %9.1f Mbases/s  %7.2f ms/step  scan frac %.3f ws %.1f GB" % (sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["frac"], j["extras"]["workspace_gb"]))
PY
}
run s2b1200
run s3b1200 --streams 3
run s3b800 --streams 3 --batch-mbases 800
run s2b800 --batch-mbases 800
run s2b1600 --batch-mbases 1600
run s4b600 --streams 4 --batch-mbases 600
for w in 4 5; do CLASSPRO_AMD_LIB=build_diag/lib_sw$w.so python scripts/seed_bench.py 2>&1 | tail -1; done
python scripts/seed_bench.py 2>&1 | tail -1
