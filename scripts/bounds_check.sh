#!/bin/bash
# The gpu suite's classification / seed / neighbour tests against a -DCP_BOUNDS build of the library (cp_bounds.h): every
# access of the per-read kernels to a read's counts and bases is checked against the read's length; the count of
# accesses outside a read must be 0 at the end.  Build first, on the build host:  bash scripts/build_diag.sh lib_bounds -DCP_BOUNDS
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bounds; rm -rf $O; mkdir -p $O
export CLASSPRO_AMD_LIB=$GRAFT_REPO_ROOT/build_diag/lib_bounds.so CP_BOUNDS_REPORT=$GRAFT_REPO_ROOT/$O/report.txt
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_neighbours.py tests/test_gpu_seeds.py tests/test_gpu_pack.py tests/test_gpu_reference.py -x -q -k "not cli and not config2_full" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log; cat $O/report.txt
