#!/bin/bash
# Row f2 at scale: scan + write all window/block files for a synthetic 300 Mb assembly through the C++
# mirror (tests/cpp/manifest_cli.cpp on include/teloscope_mi355x_io.hpp).  Run on the GPU box.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from tests import seqgen
rng = np.random.default_rng(7)
with open('/tmp/writers_rate.fa', 'wb') as fh:
    for i in range(6):
        s = seqgen.chromosome(rng, 50_000_000, telo_repeats=2000, n_its=10)
        fh.write(b'>chr%d\n' % (i + 1))
        a = np.frombuffer(bytes(s), dtype=np.uint8)                    # 80-column lines, as assemblies ship
        pad = (-len(a)) % 80
        lines = np.concatenate([a, np.full(pad, ord('A'), np.uint8)]).reshape(-1, 80)
        out = np.concatenate([lines, np.full((lines.shape[0], 1), 10, np.uint8)], axis=1).ravel()
        fh.write(out[:len(out) - pad - 1].tobytes() if pad else out.tobytes()[:-1]); fh.write(b'\n')
PY
g++ -std=c++17 -O2 -I include tests/cpp/manifest_cli.cpp -L teloscope_amd -lteloscan -Wl,-rpath,$PWD/teloscope_amd -pthread -lz -o /tmp/manifest_cli
TS_TIMING=1 /tmp/manifest_cli -f /tmp/writers_rate.fa --out-base /tmp/writers_rate -c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 500 -r -g -e -i > /tmp/writers_rate.stdout
ls -la /tmp/writers_rate_* | awk '{print $5, $9}'
