#!/bin/bash
cd "$(dirname "$0")/.."
set -o pipefail
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_shard_results.py -x -q 2>&1 | tail -2 || exit 1
for i in 1 2 3; do
  LABEL=row-trim2 python3 profiles/emit_time.py 2>/dev/null
  LABEL=before TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_pretrim.so python3 profiles/emit_time.py 2>/dev/null
done
