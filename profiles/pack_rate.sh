#!/bin/bash
# host-side packing rate (pack.cpp) against threads (the variants tried are recorded in profiles/r04/pack_rate.txt); usage: bash profiles/pack_rate.sh <tag>
set -e
TAG=${1:-a}
mkdir -p gpurun_out
g++ -O3 -std=c++17 -Iteloscope_amd/csrc -Iinclude profiles/micro/pack_rate.cpp teloscope_amd/csrc/pack.cpp -o /tmp/pack_rate -lpthread
O=gpurun_out/pack_rate_$TAG.txt
lscpu | grep "Model name" > $O
for T in 1 8 12; do /tmp/pack_rate $T | tail -2 >> $O; done
cat $O
