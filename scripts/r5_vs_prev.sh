#!/bin/bash
# the library as built against build_diag/lib_prev.so (the build of the commit before, scripts/build_diag.sh lib_prev on a
# stashed tree): parity tests, bench A/B (two runs each, twice), per-kernel durations alone and in the pipeline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_neighbours.py -m gpu -x -q 2>&1 | tail -3 || exit 1
bash scripts/r5_knobs.sh default build_diag/lib_prev.so default build_diag/lib_prev.so
bash scripts/kt.sh default build_diag/lib_prev.so
