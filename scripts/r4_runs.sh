#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_runs; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_pack.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -15 $O/pytest.log
