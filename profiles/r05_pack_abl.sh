#!/bin/bash
cd "$(dirname "$0")/.."
set -o pipefail

for i in 1 2 3; do
  LABEL=row-trim python3 profiles/emit_time.py 2>/dev/null
  LABEL=before TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_pretrim.so python3 profiles/emit_time.py 2>/dev/null
done
