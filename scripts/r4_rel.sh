#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_rel; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_neighbours.py -x -q -k "not config2_full" > $O/pytest.log 2>&1; echo "parity rc=$?"; tail -3 $O/pytest.log
bash scripts/ab.sh 2 "$@"
