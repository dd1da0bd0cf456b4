#!/bin/bash
# the off-list SELF walls of k_find_wall (FW_XCAP slots on chip): the parity / reference / neighbour tests against a build
# with ONE slot (build_diag.sh lib_xcap1 -DFW_XCAP=1), so that the full-list path -- the flags written out to the arrays
# after the replay -- runs on every read with two such walls; then the default build's counts from the diagnostic build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CLASSPRO_AMD_LIB=$GRAFT_REPO_ROOT/build_diag/lib_xcap1.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_neighbours.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -3
