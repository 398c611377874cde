#!/bin/bash
# 16-bit records in the scan's regions (the default) against 32-bit ones (TS_REC32=1): the whole GPU suite first, then plain / emitting scan
# and the sharded step, interleaved
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05_rec16
( time timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q ) > gpurun_out/r05_rec16/gputests.txt 2>&1; rc=$?
tail -5 gpurun_out/r05_rec16/gputests.txt
[ $rc -ne 0 ] && exit $rc
for i in 1 2 3; do
  LABEL=rec16 python3 profiles/emit_time.py 2>/dev/null
  LABEL=rec32 TS_REC32=1 python3 profiles/emit_time.py 2>/dev/null
done
for i in 1 2; do
  LABEL=rec16 STREAMS=probe EV=lib TIME_EVERY=0 python3 profiles/pack_abl_time.py 60 2>/dev/null
  LABEL=rec32 TS_REC32=1 STREAMS=probe EV=lib TIME_EVERY=0 python3 profiles/pack_abl_time.py 60 2>/dev/null
done
