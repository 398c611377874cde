#!/bin/bash
# Stage costs of ts_scan_tiles under real overlap: builds libteloscan with one stage compiled out
# (-DTS_ABL=<mask>, see kernels.hip) and times bench.py's workload with each.  Results are wrong by
# construction; only kernel_ms is read.  Build here (hipcc cross-compiles), run on the GPU box:
#   profiles/ablate.sh build && gpurun -- 'profiles/ablate.sh run > gpurun_out/ablate.txt'
set -e
cd "$(dirname "$0")/.."
MASKS="0 1 2 4 8 16 32 64 128"
if [ "$1" = build ]; then
    for m in $MASKS; do
        (cd teloscope_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 \
            -DTS_ABL=$m -x hip -shared -o ../../profiles/abl_$m.so kernels.hip predicate.hip generic.hip blockcall.hip exchange.hip capi.cpp \
            pipeline.cpp patterns.cpp blocks.cpp -lpthread 2>/dev/null) &
    done
    wait
else
    for m in $MASKS; do
        TELOSCAN_LIB=$PWD/profiles/abl_$m.so python3 bench.py --no-cpu-baseline --no-e2e $BENCH_FLAGS \
            | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mask %3d  kernel_ms %.4f  ms_per_step %.4f' % ($m, d['roofline']['kernel_ms'], d['ms_per_step']))"
    done
fi
