#!/bin/bash
cd "$(dirname "$0")/.."
set -o pipefail
for i in 1 2; do
  LABEL=base python3 profiles/pack_abl_time.py 2>/dev/null
  LABEL=noevents EV=none PACK=0 python3 profiles/pack_abl_time.py 2>/dev/null
  LABEL=record-only PACK=0 EV=record-only python3 profiles/pack_abl_time.py 2>/dev/null
  LABEL=wait-only PACK=0 EV=wait-only python3 profiles/pack_abl_time.py 2>/dev/null
  LABEL=slots3 SLOTS=3 python3 profiles/pack_abl_time.py 2>/dev/null
  LABEL=slots2 SLOTS=2 python3 profiles/pack_abl_time.py 2>/dev/null
  LABEL=slots1-noevents SLOTS=1 EV=none PACK=0 python3 profiles/pack_abl_time.py 2>/dev/null
done
bash profiles/shard_step_timeline.sh noev PACK=0 EV=none
