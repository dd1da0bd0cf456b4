#!/bin/bash
# parity tests of the classification path + the default bench line (one gpurun call while iterating on a kernel)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/quick
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -k "not config2_full" > gpurun_out/quick/pytest.log 2>&1 || { tail -30 gpurun_out/quick/pytest.log; exit 1; }
tail -2 gpurun_out/quick/pytest.log
timeout -k 10 300 python bench.py --no-cpu --no-extras --steps 10 --warmup 3 "$@" > gpurun_out/quick/bench.json 2> gpurun_out/quick/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/quick/bench.json").read().strip().splitlines()[-1]); print("%9.1f Mbases/s  %7.3f ms/step  scan frac %.3f" % (j["value"], j["ms_per_step"], j["roofline"]["frac"]))
PY
