#!/bin/bash
# round 4: libm + neighbour tests, the failing fuzz read, a fuzz soak, the stage parity tests, A/B of the round-3 library
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_second; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_libm.py tests/test_gpu_neighbours.py -x -q > $O/new.log 2>&1; echo "libm+neighbours rc=$?"; tail -4 $O/new.log
timeout -k 10 100 python scripts/diag_read.py tests/golden/fuzz305_34.npz 60 120 > $O/diag.log 2>&1; echo "diag rc=$?"; head -3 $O/diag.log; tail -2 $O/diag.log
timeout -k 10 500 python scripts/fuzz_parity.py 300 24 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_seeds.py -x -q > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log
bash scripts/ab.sh 2 default build_diag/lib_r3.so
