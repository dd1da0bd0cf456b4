#!/bin/bash
# k_wall_tasks with FW_RPW reads per wave (default build 4; build_diag/lib_rpw2.so, lib_rpw8.so; lib_prev.so = one read per wave,
# the commit before): the whole gpu suite on the default build, bench A/B, per-kernel durations alone and in the pipeline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
bash scripts/r5_knobs.sh default build_diag/lib_prev.so build_diag/lib_rpw2.so build_diag/lib_rpw8.so default build_diag/lib_prev.so
bash scripts/kt.sh default build_diag/lib_prev.so
