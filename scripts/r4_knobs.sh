#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash scripts/kt.sh default build_diag/lib_fw6.so 2>&1 | grep "streams="
run() { python bench.py --no-cpu --no-extras --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %9.1f Mbases/s %7.3f ms/step' % (' '.join(sys.argv[1:]), j['value'], j['ms_per_step']))" "$@"; }
run --streams 2 --batch-mbases 1200
run --streams 3 --batch-mbases 1200
run --streams 2 --batch-mbases 1700
run --streams 2 --batch-mbases 800
run --streams 3 --batch-mbases 800
run --streams 4 --batch-mbases 600
