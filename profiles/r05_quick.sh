#!/bin/bash
# quick A/B of scan kernels on one box: round 4's library, this tree, and whatever variants / environments are listed in $VARIANTS
# (entries "label:ENV=val,ENV2=val2" with TELOSCAN_LIB allowed); plain and emitting scan of the 3 Gb assembly, ms per launch
cd "$(dirname "$0")/.."
for i in $(seq 1 ${REPS:-2}); do
  (cd ab_old && LABEL=r04 python3 profiles/emit_time.py ${LAUNCHES:-60} 2>/dev/null)
  LABEL=r05 python3 profiles/emit_time.py ${LAUNCHES:-60} 2>/dev/null
  for v in $VARIANTS; do
    lab=${v%%:*}; envs=${v#*:}
    env LABEL=$lab $(echo $envs | tr ';' ' ') python3 profiles/emit_time.py ${LAUNCHES:-60} 2>/dev/null
  done
done
