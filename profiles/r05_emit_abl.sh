#!/bin/bash
# where the emitting build's time goes: library variants with pieces of finish_records compiled out (-DTS_EMIT_ABL: 1 rows not looked
# at, 2 no visible records, 4 no chain summary), the per-window accumulators (TS_ACC_PER_WINDOW=1), round 4's library; plain / emit ms
cd "$(dirname "$0")/.."
for i in 1 2; do
  (cd ab_old && LABEL=r04 python3 profiles/emit_time.py 2>/dev/null)
  LABEL=r05 python3 profiles/emit_time.py 2>/dev/null
  LABEL=r05-acc-per-window TS_ACC_PER_WINDOW=1 python3 profiles/emit_time.py 2>/dev/null
  for v in 1 2 4 6; do LABEL=r05-eabl$v TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_eabl$v.so python3 profiles/emit_time.py 2>/dev/null; done
done
