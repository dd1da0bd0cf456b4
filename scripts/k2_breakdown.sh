#!/bin/bash
# timing-only experiment: find_wall truncated after the walk (0: before broadcast, 1: after walk, 2: before emission)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp classpro_amd/libclasspro_amd.so /tmp/lib_full.so
for v in 0 1 2; do
  cp build/lib_stop$v.so classpro_amd/libclasspro_amd.so
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k2b_$v -- python bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/k2b_$v.log 2>&1
  grep k_find_wall gpurun_out/k2b_$v/*/*_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
done
cp /tmp/lib_full.so classpro_amd/libclasspro_amd.so
