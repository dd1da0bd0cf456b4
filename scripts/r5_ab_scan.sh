#!/bin/bash
# scan-kernel A/B inside one call: the library at HEAD against a build with the round-4 store form (-DSCAN_BYTE_STORES)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stage_parity or scale_properties or edge" 2>&1 | tail -3
for l in default build_diag/lib_bytestores.so; do
  if [ "$l" = default ]; then unset CLASSPRO_AMD_LIB; else export CLASSPRO_AMD_LIB=$GRAFT_REPO_ROOT/$l; fi
  for i in 1 2; do
    python bench.py --steps 10 --warmup 3 --no-cpu --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$l', 'value %.1f Gb/s  step %.2f ms  scan %.1f us  frac %.3f' % (d['value']/1e3, d['ms_per_step'], r['ms_per_launch']*1e3, r['frac']))"
  done
done
