#!/bin/bash
# PCIe pipeline variants inside one gpurun call: slots, pack threads, batch size, SDMA off
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_pcie; rm -rf $O; mkdir -p $O
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo; lscpu | grep -i "numa\|model name\|socket" 
run() { name=$1; shift; timeout -k 10 300 "$@" > $O/$name.json 2> $O/$name.err; python - "$O/$name.json" "$name" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=j["extras"].get("pcie") or j["extras"]
    print("%-22s %8.1f Mbases/s h2d %.1f d2h %.1f GB/s  peaks h2d %.1f d2h %.1f both ways in+out %.1f  pack %s batch %.0f" % (sys.argv[2], p["mbases_per_s"], p["h2d_gb_per_s"], p["d2h_gb_per_s"], p["pinned_h2d_peak_gb_per_s"], p["pinned_d2h_peak_gb_per_s"], p["pinned_both_ways_gb_per_s_in_plus_out"], p["host_pack_seconds_per_slot"], p["batch_mbases"]))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
B="python bench.py --steps 2 --warmup 1 --no-cpu --only-pcie"
run base $B
run base_again $B
run slots4 $B --pcie-slots 4
run slots6 $B --pcie-slots 6
run pack16 $B --pcie-pack-threads 16
run batch200 $B --pcie-batch-mbases 200
run batch800 $B --pcie-batch-mbases 800
HSA_ENABLE_SDMA=0 run nosdma $B
