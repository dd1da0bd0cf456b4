#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu --no-extras --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %9.1f Mbases/s %7.3f ms/step  sub-batches %d  workspace %.1f GB' % (' '.join(sys.argv[1:]), j['value'], j['ms_per_step'], j['extras']['sub_batches_per_rank'], j['extras']['workspace_gb']))" "$@"; }
for i in 1 2; do
run --batch-mbases 2100
run --batch-mbases 2700 --streams 3
run --batch-mbases 2100
run --batch-mbases 4100
done
