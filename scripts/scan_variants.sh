#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp classpro_amd/libclasspro_amd.so /tmp/lib_full.so
echo "base:"; python bench.py --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['achieved'], j['roofline']['ms_per_launch'])"
for v in 0 1 2 3 4 5; do
  cp build/lib_scan$v.so classpro_amd/libclasspro_amd.so
  echo "variant $v:"; python bench.py --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['achieved'], j['roofline']['ms_per_launch'])"
done
cp /tmp/lib_full.so classpro_amd/libclasspro_amd.so
