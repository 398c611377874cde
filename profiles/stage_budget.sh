#!/bin/bash
# Per-stage budget of ts_scan_tiles on configs[1] (round 4's VERDICT, item 6): library variants with one stage compiled out
# (-DTS_ABL=<mask>, kernels.hip; results are then wrong) -> kernel time (bench.py's HIP events) and SQ_INSTS_VALU / _SALU / _LDS per
# launch (one rocprofv3 --pmc pass each).  Variants are built here first:  for m in 1 2 4 8 16 32 64 128; do build_variant abl$m -DTS_ABL=$m; done
# usage (GPU box): bash profiles/stage_budget.sh > gpurun_out/stage_budget.txt
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/stage_budget
mkdir -p $OUT
export TS_BENCH_NO_BLOCKS_RECORD=1
ARGS="$REPO/bench.py --no-cpu-baseline --no-e2e --no-reads"
for m in 0 1 2 4 8 16 32 64 128; do
  if [ $m = 0 ]; then unset TELOSCAN_LIB; else export TELOSCAN_LIB=$REPO/teloscope_amd/libteloscan_abl$m.so; fi
  ms=$(cd $REPO && timeout -k 10 300 python3 $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.4f' % d['roofline']['kernel_ms'])")
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES \
      --output-format csv -d $OUT/m$m -- python3 $ARGS --steps 8 --warmup 2 > $OUT/m$m.log 2>&1)
  echo "mask $m kernel_ms $ms $(python3 $REPO/profiles/kernel_counters.py $OUT/m$m | grep '^scan' | awk '{printf "%s %s  ", $2, $NF}')"
done
