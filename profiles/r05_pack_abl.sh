#!/bin/bash
cd "$(dirname "$0")/.."
set -o pipefail
LABEL=warm python3 profiles/emit_time.py 2>/dev/null
for i in 1 2 3; do
  LABEL=base python3 profiles/emit_time.py 2>/dev/null
  for d in 1 2 3; do LABEL=l2-prefetch-$d TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_pf$d.so python3 profiles/emit_time.py 2>/dev/null; done
done
