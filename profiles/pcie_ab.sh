#!/bin/bash
# pcie_inclusive legs of bench.py under an environment A/B; usage: bash profiles/pcie_ab.sh <tag> VAR=a VAR=b ...
set -e
TAG=$1; shift
make -s -C teloscope_amd/csrc && make -s -C oracle
mkdir -p gpurun_out
O=gpurun_out/pcie_ab_$TAG.txt
: > $O
for SETTING in "$@"; do
  echo "$SETTING" >> $O
  env $SETTING timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-reads --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d['pcie_inclusive']
print('  ', {k:(v['gbases_per_s'] if isinstance(v,dict) and 'gbases_per_s' in v else None) for k,v in p.items() if isinstance(v,dict)})" >> $O
done
cat $O
