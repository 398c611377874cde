#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun).
# usage: bash profiles/run_profile.sh <round-tag>      -> gpurun_out/prof_<tag>/...
set -e
TAG=${1:-r04}
REPO=$(pwd)
# build first, outside the profiler: bench.py must never spawn make / hipcc from a profiled, GPU-initialised process
make -s -C $REPO/teloscope_amd/csrc && make -s -C $REPO/oracle
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --no-cpu-baseline --no-e2e --no-reads"        # bench.py defaults: --steps 50 --warmup 5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
    --output-format csv -d $OUT/pmc_sq1 -- python3 $ARGS > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR \
    --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
python3 $REPO/profiles/summarize.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
