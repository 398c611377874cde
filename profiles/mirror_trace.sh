#!/bin/bash
# per-group stage intervals of scanFastaToFiles (TS_MIRROR_TRACE=1) on the writers_rate input
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
sed -n '1,/^PY$/p' profiles/writers_rate.sh | sed -n '/^python3 - /,/^PY$/p' > /tmp/gen.sh
NREC=60 MB=50 bash -c "$(sed -e 's/\$NREC/60/; s/\$MB/50/' /tmp/gen.sh)"
make -s -C teloscope_amd/csrc
g++ -std=c++17 -O2 -I include tests/cpp/manifest_cli.cpp -L teloscope_amd -lteloscan -Wl,-rpath,$PWD/teloscope_amd -pthread -lz -o /tmp/manifest_cli
FLAGS="-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 500 -r -g -e -i"
cat /tmp/writers_rate.fa > /dev/null
for run in 1 2; do
  TS_MIRROR_TRACE=1 /tmp/manifest_cli -f /tmp/writers_rate.fa --out-base /tmp/writers_rate $FLAGS 2>&1 >/dev/null | grep -E "manifest_cli|trace" > gpurun_out/mirror_trace_$run.txt
done
cat gpurun_out/mirror_trace_2.txt
rm -f /tmp/writers_rate*
