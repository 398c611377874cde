#!/bin/bash
# pcie_inclusive against the number of staging threads (TS_STAGE_THREADS); usage: bash profiles/stage_threads_sweep.sh <tag>
set -e
TAG=${1:-a}
make -s -C teloscope_amd/csrc && make -s -C oracle
mkdir -p gpurun_out
O=gpurun_out/stage_threads_$TAG.txt
echo "nproc $(nproc), cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null || echo n/a)" > $O
for T in 8 12 16 24; do
  echo "TS_STAGE_THREADS=$T" >> $O
  TS_STAGE_THREADS=$T timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-reads --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d['pcie_inclusive']
print({k:(v['gbases_per_s'] if isinstance(v,dict) and 'gbases_per_s' in v else None) for k,v in p.items() if isinstance(v,dict)})" >> $O
done
cat $O
