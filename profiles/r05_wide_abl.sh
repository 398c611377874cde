#!/bin/bash
# where the wide form's time goes: profiles/general_path_rate.py on the nine-length set with stages switched off (TS_GEN_ABL: 128 no
# matching, 256 no window records; results are then wrong), TS_TIMING=1 stage sums of the blocks-only call
cd "$(dirname "$0")/.."
export TS_GEN_ONLY=wide_9_lengths TS_TIMING=1
for abl in 0 128 256 384; do
  echo "== TS_GEN_ABL=$abl"
  TS_GEN_ABL=$abl timeout -k 10 300 python3 profiles/general_path_rate.py 3.0 2>&1 | grep -E "general path|kernels|wide_9" | tail -4 | cut -c1-300
done
