#!/bin/bash
# seed path while iterating: its gpu tests, then the rate on the 60x r=25000 set (scripts/seed_bench.py); extra libs to A/B as arguments
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/seedq
timeout -k 10 600 python -m pytest tests/test_gpu_seeds.py -x -q > gpurun_out/seedq/pytest.log 2>&1 || { tail -30 gpurun_out/seedq/pytest.log; exit 1; }
tail -1 gpurun_out/seedq/pytest.log
python scripts/seed_bench.py 2>&1 | tail -1
for l in "$@"; do echo "$l"; CLASSPRO_AMD_LIB=$l python scripts/seed_bench.py 2>&1 | tail -1; done
