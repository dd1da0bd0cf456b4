#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_libm2; rm -rf $O; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_libm.py tests/test_gpu_parity.py tests/test_gpu_neighbours.py -x -q -k "not config2_full" > $O/pytest.log 2>&1; echo "parity rc=$?"; tail -3 $O/pytest.log
timeout -k 10 500 python scripts/fuzz_parity.py 324 24 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
bash scripts/ab.sh 2 default build_diag/lib_prev.so
bash scripts/kt.sh default 2>&1 | grep "streams="
