#!/bin/bash
# round 4: the default bench with the new extras.pcie, the nccl path on one device, the pre-flight message, configs[3] whole at N=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_bench; rm -rf $O; mkdir -p $O
timeout -k 10 400 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r4_bench/bench.json").read().strip().splitlines()[-1])
print("default:", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["cpu_baseline"]["value"], j["cpu_baseline"]["label_mismatches_vs_hip"])
print(json.dumps(j["extras"].get("pcie", j["extras"].get("pcie_error")), indent=1))
PY
timeout -k 10 200 python bench.py --gpus 1 --backend nccl --force-dist --no-cpu --no-extras --steps 5 --warmup 2 > $O/nccl1.json 2> $O/nccl1.err; echo "nccl world 1 rc=$?"; tail -2 $O/nccl1.err; cut -c1-300 $O/nccl1.json
timeout -k 10 200 python bench.py --gpus 2 --backend gloo --no-cpu --no-extras --steps 2 --warmup 1 > $O/n2.json 2> $O/n2.err; echo "2 ranks on one GPU, default windows rc=$? (2 expected)"; grep "bench.py:" $O/n2.err | head -3
timeout -k 10 500 python bench.py --genome 3e9 --steps 5 --warmup 2 --no-cpu --no-extras > $O/config3_n1.json 2> $O/config3_n1.err; echo "configs[3] whole at N=1 rc=$?"; tail -2 $O/config3_n1.err; cut -c1-1200 $O/config3_n1.json
