#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3i; rm -rf $O; mkdir -p $O
run() { name=$1; shift; env "$@" python bench.py --no-cpu --no-extras --steps 3 --warmup 1 > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("%-12s %9.1f Mbases/s  scan %.1f GB/s frac %.3f" % (sys.argv[2], j["value"], j["roofline"]["achieved"], j["roofline"]["frac"]))
PY
}
run default X=1
for v in u8 u2 b16 b4 u8b4 u2b16; do run $v CLASSPRO_AMD_LIB=build_diag/lib_scan_$v.so; done
run default2 X=1
CLASSPRO_AMD_LIB=build_diag/lib_sw5.so python scripts/seed_bench.py 2>&1 | tail -1
python scripts/seed_bench.py 2>&1 | tail -1
