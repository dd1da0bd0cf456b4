#!/bin/bash
# full validation at HEAD: the whole gpu suite (incl. configs[3]), smoke, the default bench, a 2-rank rehearsal of the
# N > 1 path with resident windows (gloo: both ranks on the one GPU), a shard of configs[3]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/validate; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
python - <<'PY'
import json
j=json.loads(open("gpurun_out/validate/bench.json").read().strip().splitlines()[-1])
print("default:", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["cpu_baseline"]["value"], j["cpu_baseline"]["label_mismatches_vs_hip"])
print({k:v for k,v in j["extras"].items() if k not in ("cli_end_to_end",)})
PY
python bench.py --gpus 2 --backend gloo --genome 4e8 --window-gbases 4 --steps 3 --warmup 1 > $O/n2.json 2> $O/n2.err || { tail $O/n2.err; exit 1; }
python - <<'PY'
import json
j=json.loads(open("gpurun_out/validate/n2.json").read().strip().splitlines()[-1])
print("2 ranks (gloo, one GPU):", j["value"], j["n_gpus"], j["config"]["workload"][:200], j["extras"]["resident_windows_per_rank"])
PY
# two ranks with the default 30-Gbase windows on ONE card: the pre-flight check must stop all of them with exit code 2 and a sentence
set +e
python bench.py --gpus 2 --backend gloo --no-cpu --no-extras --steps 1 --warmup 1 > $O/n2_default.json 2> $O/n2_default.err; rc=$?
set -e
echo "2 ranks, default windows, one card: launcher exit code $rc (every rank exits 2: below)"; grep -o "exitcode  : [0-9]*" $O/n2_default.err | sort | uniq -c; grep "bench.py: rank" $O/n2_default.err | head -2
python bench.py --gpus 1 --backend nccl --force-dist --no-cpu --no-extras --steps 3 --warmup 1 > $O/nccl1.json 2> $O/nccl1.err && echo "nccl world 1 ok: $(cut -c1-120 $O/nccl1.json)"
