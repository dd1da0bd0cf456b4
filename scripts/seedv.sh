#!/bin/bash
# rate of the seed path for each library variant given (no tests)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for l in "$@"; do echo "$l"; CLASSPRO_AMD_LIB=$l python scripts/seed_bench.py 2>&1 | tail -${SEEDV_TAIL:-1}; done
