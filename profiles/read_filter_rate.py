#!/usr/bin/env python3
"""configs[3] scaled down: --fastq-subset -l 42 on synthetic HiFi reads (N(15 kb, 3 kb) clipped to
[1 kb, 40 kb], 0.5 % with a 300-8000 b terminal TTAGGG/CCCTAA tract at 1 % substitutions).
Times ts_filter_reads end to end (host reads in, one pass byte per read out: staging + H2D + tips
scan + on-device terminal-block predicate) and checks a sample against the oracle."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tests import harness as H
from tests import seqgen
from tests.backends import OracleReadFilter, ProductReadFilter

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
rng = np.random.default_rng(43)
lens = np.clip(rng.normal(15000, 3000, size=n_reads), 1000, 40000).astype(np.int64)
pool = seqgen.random_dna(rng, int(lens.sum()))
offs = np.concatenate(([0], np.cumsum(lens)))
tel = np.flatnonzero(rng.random(n_reads) < 0.005)
for i in tel:
    ln = int(rng.integers(300, 8000))
    unit = "TTAGGG" if rng.random() < 0.5 else "CCCTAA"
    t = seqgen.mutate(rng, seqgen.repeat_array(unit, ln // 6 + 1), 0.01)[:min(ln, lens[i])]
    if rng.random() < 0.5:
        pool[offs[i]:offs[i] + len(t)] = t
    else:
        pool[offs[i + 1] - len(t):offs[i + 1]] = t
buf = pool.tobytes()
reads = [buf[offs[i]:offs[i + 1]] for i in range(n_reads)]
opts = H.parse_cli(os.environ.get("TS_RF_FLAGS", "--fastq-subset -l 42"))
rf = ProductReadFilter(opts)
rf.filter(reads[:1000])                                   # warm-up
t0 = time.perf_counter()
got = rf.filter(reads)
dt = time.perf_counter() - t0
nb = int(lens.sum())
print("ts_filter_reads: %d reads, %.2f Gb in %.2f s = %.2f Gbases/s (%.0f reads/s), kept %d (planted %d)"
      % (n_reads, nb / 1e9, dt, nb / dt / 1e9, n_reads / dt, sum(got), len(tel)))
idx = sorted(set(list(tel[:500]) + list(rng.integers(0, n_reads, size=1500))))
exp = OracleReadFilter(opts).filter([reads[i] for i in idx])
assert [got[i] for i in idx] == exp, "read filter parity failed"
print("parity vs oracle on %d sampled reads (all planted ones among them): OK" % len(idx))
