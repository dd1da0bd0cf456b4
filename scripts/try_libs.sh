#!/bin/bash
# Diagnostic: run the default bench with each diagnostic library given (paths relative to the repo root).
for lib in "$@"; do
  CLASSPRO_AMD_LIB=$lib timeout -k 10 200 python bench.py --no-cpu 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib', j['value'], j['ms_per_step'], 'scan GB/s', j['roofline']['achieved'])"
done
