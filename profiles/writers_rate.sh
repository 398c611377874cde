#!/bin/bash
# Rows f2 + f3 at scale: FASTA text in, all window/block files out, for a synthetic assembly (default 3 Gb:
# 60 x 50 Mb records, 80-column lines) through the C++ mirror (tests/cpp/manifest_cli.cpp on
# include/teloscope_mi355x_io.hpp): the streaming pipeline (scanFastaToFiles: text pieces, lines never joined; and with
# lines joined on the host) and the three-phase path.
# Run on the GPU box:  profiles/writers_rate.sh [records] [Mb per record]
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
NREC=${1:-60}
MB=${2:-50}
python3 - $NREC $MB <<'PY'
import numpy as np, sys
nrec, mb = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(7)
lut = np.frombuffer(b'ACGT', dtype=np.uint8)
p = np.tile(np.frombuffer(b'CCCTAA', dtype=np.uint8), 2000)
q = np.tile(np.frombuffer(b'TTAGGG', dtype=np.uint8), 2000)
with open('/tmp/writers_rate.fa', 'wb') as fh:
    for i in range(nrec):
        a = lut[rng.integers(0, 4, size=mb * 1_000_000, dtype=np.uint8)]
        a[:len(p)] = p; a[-len(q):] = q
        for _ in range(10):                                            # a few interstitial blocks
            at = int(rng.integers(100_000, len(a) - 100_000)); a[at:at + 600] = np.tile(q[:6], 100)
        fh.write(b'>chr%d\n' % (i + 1))
        pad = (-len(a)) % 80                                           # 80-column lines, as assemblies ship
        lines = np.concatenate([a, np.full(pad, ord('A'), np.uint8)]).reshape(-1, 80)
        out = np.concatenate([lines, np.full((lines.shape[0], 1), 10, np.uint8)], axis=1).ravel()
        fh.write(out[:len(out) - pad - 1].tobytes() if pad else out.tobytes()[:-1]); fh.write(b'\n')
PY
g++ -std=c++17 -O2 -I include tests/cpp/manifest_cli.cpp -L teloscope_amd -lteloscan -Wl,-rpath,$PWD/teloscope_amd -pthread -lz -o /tmp/manifest_cli
FLAGS="-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 500 -r -g -e -i"
ls -la /tmp/writers_rate.fa | awk '{print "# input:", $5, "bytes"}'
cat /tmp/writers_rate.fa > /dev/null                                   # page cache warm, as after a download
for run in 1 2; do
  echo "# streaming (scanFastaToFiles), run $run"
  TS_TIMING=1 /tmp/manifest_cli -f /tmp/writers_rate.fa --out-base /tmp/writers_rate $FLAGS 2>&1 >/tmp/writers_rate.stdout | grep -E "manifest_cli|ts_scan_segments"
done
echo "# streaming, records joined on the host (--join-lines)"
TS_TIMING=1 /tmp/manifest_cli -f /tmp/writers_rate.fa --out-base /tmp/writers_rate_j --join-lines $FLAGS 2>&1 >/tmp/writers_rate_j.stdout | grep -E "manifest_cli"
for f in /tmp/writers_rate_j_*; do cmp $f /tmp/writers_rate_${f#/tmp/writers_rate_j_} || echo "DIFFERS: $f"; done
echo "# three phases (readFasta, walkPaths, writeBEDFiles)"
TS_TIMING=1 /tmp/manifest_cli -f /tmp/writers_rate.fa --out-base /tmp/writers_rate_3p --no-stream $FLAGS 2>&1 >/tmp/writers_rate_3p.stdout | grep -E "manifest_cli"
cmp /tmp/writers_rate.stdout /tmp/writers_rate_3p.stdout && echo "# console text identical"
for f in /tmp/writers_rate_3p_*; do cmp $f /tmp/writers_rate_${f#/tmp/writers_rate_3p_} || echo "DIFFERS: $f"; done
ls -la /tmp/writers_rate_* | awk '{s += $5} END {print "# output bytes (both runs):", s}'
echo "# with -m (match vectors cross PCIe, matchSeq per record), streaming"
TS_TIMING=1 /tmp/manifest_cli -f /tmp/writers_rate.fa --out-base /tmp/writers_rate_m $FLAGS -m 2>&1 >/dev/null | grep -E "manifest_cli"
echo "# the reference's default mode (no -r: tips-only scan of the contig ends, terminal blocks + summary), streaming, twice"
for run in 1 2; do
TS_TIMING=1 /tmp/manifest_cli -f /tmp/writers_rate.fa --out-base /tmp/writers_rate_t -c TTAGGG 2>&1 >/dev/null | grep -E "manifest_cli"
done
[ -n "$KEEP" ] || rm -f /tmp/writers_rate*
