#!/bin/bash
cd "$(dirname "$0")/.."
set -o pipefail
export GBASES=0.375 CONTIGS=25
for i in 1 2; do
  LABEL=base python3 profiles/pack_abl_time.py 200 2>/dev/null
  LABEL=noevents EV=none PACK=0 python3 profiles/pack_abl_time.py 200 2>/dev/null
  LABEL=record-only PACK=0 EV=record-only python3 profiles/pack_abl_time.py 200 2>/dev/null
  LABEL=wait-only PACK=0 EV=wait-only python3 profiles/pack_abl_time.py 200 2>/dev/null
  LABEL=hip-events EV=hip python3 profiles/pack_abl_time.py 200 2>/dev/null
done
bash profiles/shard_step_timeline.sh small PACK=both
