#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3h; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "not config3 and not config2_full and not seeds" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
run() { name=$1; shift; env "$@" python bench.py --no-cpu --no-extras --steps 10 --warmup 3 > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("%-12s %9.1f Mbases/s  %7.2f ms/step  scan frac %.3f" % (sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["frac"]))
PY
}
run default X=1
run default2 X=1
CLASSPRO_AMD_LIB=build_diag/lib_walk.so python scripts/walk_profile.py > $O/walk.txt 2>&1
cat $O/walk.txt
