#!/bin/bash
# Round 5, step 1 on one box: the shard / emit parity tests on the new emitting scan (records looked at as they leave the stage,
# window records packed by the scan, per-segment sums in the screen kernel), then an A/B of the default bench line against
# round 4's tree (ab_old/) — plain scan, emitting scan, scan_plus_block_calling — twice each, interleaved.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_step1
mkdir -p $OUT
set -o pipefail
timeout -k 10 900 python3 -m pytest tests/test_gpu_shard_results.py -x -q 2>&1 | tail -15 > $OUT/tests_shard.txt
rc=$?; cat $OUT/tests_shard.txt; [ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "block_calling or emit or manifest or val or device_block" 2>&1 | tail -8 > $OUT/tests_parity_subset.txt
rc=$?; cat $OUT/tests_parity_subset.txt; [ $rc -ne 0 ] && exit $rc
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d.get('scan_plus_block_calling',{})
print('%-10s step %.4f ms kernel %.4f ms | sharded step %.4f  emit-scan alone %.4f  pack alone %.4f  (x%.3f of plain)' % (sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], s.get('ms_per_step',0), s.get('emitting_scan_alone_ms',0), s.get('block_calling_and_pack_alone_ms',0), s.get('ms_per_step',0)/d['ms_per_step']))" "$1"; }
for i in 1 2; do
  (cd ab_old && timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-e2e --no-reads 2>/dev/null) | tee $OUT/bench_old_$i.json | line r04 || exit 1
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-e2e --no-reads 2>/dev/null | tee $OUT/bench_new_$i.json | line r05 || exit 1
  TS_SHARD_BIND=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-e2e --no-reads 2>/dev/null | tee $OUT/bench_new_nobind_$i.json | line r05-nobind || exit 1
done
# how long the fuzzes the suite now runs take (bar: 120 s)
( time timeout -k 10 600 python3 -m pytest tests/test_gpu_fuzz.py -x -q ) 2>&1 | tail -8 | tee $OUT/fuzz_tests.txt
