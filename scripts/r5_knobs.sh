#!/bin/bash
# bench value for library variants, two runs each, inside one call
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for l in "$@"; do
  if [ "$l" = default ]; then unset CLASSPRO_AMD_LIB; else export CLASSPRO_AMD_LIB=$GRAFT_REPO_ROOT/$l; fi
  for i in 1 2; do
    python bench.py --steps 10 --warmup 3 --no-cpu --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-32s value %.1f Gb/s  step %.2f ms' % ('$l', d['value']/1e3, d['ms_per_step']))"
  done
done
