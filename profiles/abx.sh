#!/bin/bash
# A/B of kernel experiments in ONE gpurun call (box-to-box variance is larger than most kernel changes):
#   profiles/abx.sh build "0 1 3"       builds teloscope_amd/csrc with -DTS_EXP=<mask> into profiles/abx_<mask>.so
#   profiles/abx.sh run "0 1 3" [rounds]  on the GPU box: kernel_ms of the three reference configurations per variant
# profiles/abx_base.so, if present (e.g. the previous commit's build), is measured as "base".
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
    for m in $2; do
        # a variant is a TS_EXP mask, or pXYZ = wave priorities of the per-match pass / the queue append / the window phase
        # qN = a match queue of N entries (TS_LIST)
        case $m in p???) DEFS="-DTS_PRIO_PASS=${m:1:1} -DTS_PRIO_APPEND=${m:2:1} -DTS_PRIO_WINDOWS=${m:3:1}";; q*) DEFS="-DTS_LIST=${m:1}";; *) DEFS="-DTS_EXP=$m";; esac
        (cd teloscope_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 \
            $DEFS -x hip -shared -o ../../profiles/abx_$m.so kernels.hip predicate.hip generic.hip blockcall.hip exchange.hip capi.cpp \
            pipeline.cpp patterns.cpp blocks.cpp -lpthread 2>/dev/null) &
    done
    wait
    ls -la profiles/abx_*.so
    exit 0
fi
run() { TELOSCAN_LIB=$LIB timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-e2e "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-6s %-9s kernel %.4f ms  frac %.4f' % (sys.argv[1], sys.argv[2], d['roofline']['kernel_ms'], d['roofline']['frac']))" "$TAG" "$LABEL"; }
ROUNDS=${3:-2}
VARIANTS="$2"
[ -f profiles/abx_base.so ] && VARIANTS="base $VARIANTS"
for r in $(seq 1 $ROUNDS); do
  for m in $VARIANTS; do
    LIB=$PWD/profiles/abx_$m.so; TAG=exp$m
    LABEL=headline run
    LABEL=plant run --flags "-c CCCTAAA -w 2000 -s 1000 -r -g -e -m -i"
    LABEL=default run --flags "-c TTAGGG -r -g -e -m -i"
  done
done
