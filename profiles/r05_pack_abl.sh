#!/bin/bash
cd "$(dirname "$0")/.."
set -o pipefail
for i in 1 2 3; do
LABEL=script STREAMS=probe EV=lib TIME_EVERY=0 python3 profiles/pack_abl_time.py 60 2>/dev/null
timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-e2e --no-reads 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['scan_plus_block_calling']
print('bench: plain', d['ms_per_step'], d['roofline']['kernel_ms'], 'sharded', s['ms_per_step'], round(s['ms_per_step']/d['ms_per_step'],3), 'emit alone', s['emitting_scan_alone_ms'], 'pack alone', s['block_calling_and_pack_alone_ms'])"
done
