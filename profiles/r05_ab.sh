#!/bin/bash
# A/B on one box: round 4's tree (ab_old/) against this tree — plain scan, emitting scan, scan_plus_block_calling; optional env for
# the new tree in $NEW_ENV (e.g. "TS_STAGE_U32=1"); $REPS repetitions (default 2), interleaved.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_ab
mkdir -p $OUT
set -o pipefail
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d.get('scan_plus_block_calling',{})
print('%-14s step %.4f ms kernel %.4f ms | sharded step %.4f  emit-scan alone %.4f  pack alone %.4f  (x%.3f of plain)' % (sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], s.get('ms_per_step',0), s.get('emitting_scan_alone_ms',0), s.get('block_calling_and_pack_alone_ms',0), s.get('ms_per_step',0)/d['ms_per_step']))" "$1"; }
ARGS="--no-cpu-baseline --no-e2e --no-reads"
for i in $(seq 1 ${REPS:-2}); do
  (cd ab_old && timeout -k 10 300 python3 bench.py $ARGS 2>/dev/null) | tee $OUT/old_$i.json | line r04 || exit 1
  timeout -k 10 300 python3 bench.py $ARGS 2>/dev/null | tee $OUT/new_$i.json | line r05 || exit 1
  for e in $NEW_ENVS; do
    env $e timeout -k 10 300 python3 bench.py $ARGS 2>/dev/null | tee $OUT/new_${e%%=*}_$i.json | line "r05 $e" || exit 1
  done
done
