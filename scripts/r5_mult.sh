#!/bin/bash
# the multi-error search of k_find_wall with the O-only walls pre-filtered by their memo entries (default) against the build
# that takes every wall (build_diag/lib_nopre.so = -DCP_NO_MULT_PREFILTER): parity tests, bench A/B, per-kernel durations
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_neighbours.py -m gpu -x -q 2>&1 | tail -3 || exit 1
bash scripts/r5_knobs.sh default build_diag/lib_nopre.so default build_diag/lib_nopre.so
bash scripts/kt.sh default build_diag/lib_nopre.so
