#!/bin/bash
# SQ counters of the scan kernel with and without the emitting build (TS_BENCH_EMIT=1), same box.
# usage: bash profiles/emit_counters.sh <tag> -> gpurun_out/emitpmc_<tag>/{e0,e1}_*/ + summary.txt
set -e
TAG=${1:-r04}
REPO=$(pwd)
make -s -C $REPO/teloscope_amd/csrc && make -s -C $REPO/oracle
OUT=$REPO/gpurun_out/emitpmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --no-cpu-baseline --no-e2e --no-reads --steps 10 --warmup 2"
for E in 0 1; do
  export TS_BENCH_EMIT=$E
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
      --output-format csv -d $OUT/e${E}_sq1 -- python3 $ARGS > $OUT/e${E}_sq1.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU \
      --output-format csv -d $OUT/e${E}_sq2 -- python3 $ARGS > $OUT/e${E}_sq2.log 2>&1 || true
  rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM \
      --output-format csv -d $OUT/e${E}_sq3 -- python3 $ARGS > $OUT/e${E}_sq3.log 2>&1 || true
done
for d in $OUT/e*_sq*/; do echo "== $d"; python3 $REPO/profiles/kernel_counters.py $d | grep scan; done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
