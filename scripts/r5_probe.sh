#!/bin/bash
# what the GPU box gives a command: cores, cgroup quota, NUMA layout, memory
echo "nproc: $(nproc)"; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
echo "mem.max: $(cat /sys/fs/cgroup/memory.max 2>/dev/null)"
lscpu | grep -i "model name\|socket\|numa\|^CPU(s)"
free -g | head -2
cat /sys/kernel/mm/transparent_hugepage/enabled
