#!/bin/bash
# plain build: the tile's staged 16-bit records copied four to a lane: parity, then plain / emit / reads new against old
cd "$(dirname "$0")/.."
set -o pipefail
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2 || exit 1
for i in 1 2 3; do
  LABEL=new python3 profiles/emit_time.py 2>/dev/null
  LABEL=old TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_prewin.so python3 profiles/emit_time.py 2>/dev/null
done
for v in new old new old; do
  if [ $v = old ]; then export TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_prewin.so; else unset TELOSCAN_LIB; fi
  timeout -k 10 300 python3 bench.py --reads --n-reads 1e6 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v reads 1e6: step', d['ms_per_step'], 'value', d['value'], 'scan kernel', d['roofline']['kernel_ms'])"
done
