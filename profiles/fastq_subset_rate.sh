#!/bin/bash
# Row f3 at scale: --fastq-subset -l 42 on synthetic HiFi-like reads (N(15 kb, 3 kb), 0.5 % telomeric)
# through the C++ mirror (tests/cpp/manifest_cli.cpp --fastq-subset).  Run on the GPU box.
set -e
cd "$(dirname "$0")/.."
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from tests import seqgen
rng = np.random.default_rng(43)
n = 100_000
lens = np.clip(rng.normal(15000, 3000, size=n), 1000, 40000).astype(np.int64)
pool = seqgen.random_dna(rng, int(lens.sum()))
offs = np.concatenate(([0], np.cumsum(lens)))
for i in np.flatnonzero(rng.random(n) < 0.005):
    ln = int(rng.integers(300, 8000))
    t = seqgen.mutate(rng, seqgen.repeat_array("TTAGGG", ln // 6 + 1), 0.01)[:min(ln, lens[i])]
    pool[offs[i]:offs[i] + len(t)] = t
buf = pool.tobytes()
with open('/tmp/reads.fq', 'wb') as fh:
    for i in range(n):
        s = buf[offs[i]:offs[i + 1]]
        fh.write(b'@r%d\n' % i); fh.write(s); fh.write(b'\n+\n'); fh.write(b'I' * len(s)); fh.write(b'\n')
print("reads", n, "bases", int(lens.sum()))
PY
g++ -std=c++17 -O2 -I include tests/cpp/manifest_cli.cpp -L teloscope_amd -lteloscan -Wl,-rpath,$PWD/teloscope_amd -pthread -lz -o /tmp/manifest_cli
/tmp/manifest_cli --fastq-subset -l 42 /tmp/reads.fq > /tmp/kept.fq     # warm-up (page cache, device init)
t0=$(date +%s%N); TS_TIMING=1 /tmp/manifest_cli --fastq-subset -l 42 /tmp/reads.fq > /tmp/kept.fq; t1=$(date +%s%N); echo "wall $(( (t1 - t0) / 1000000 )) ms"
ls -la /tmp/reads.fq /tmp/kept.fq | awk '{print $5, $9}'
echo "# the read shard: two read-filter contexts (here on one GPU), every batch cut into two consecutive shards"
t0=$(date +%s%N); TS_TIMING=1 /tmp/manifest_cli --fastq-subset -l 42 --read-devices 0,0 /tmp/reads.fq > /tmp/kept2.fq; t1=$(date +%s%N); echo "wall $(( (t1 - t0) / 1000000 )) ms"
cmp /tmp/kept.fq /tmp/kept2.fq && echo "# identical output"
