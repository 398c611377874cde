#!/usr/bin/env python3
"""One-off extended fuzz of the read filter (not part of the test suite): random block-calling parameters x reads of
every kind the predicate kernels tell apart — short and long match lists, terminal tracts of both orientations and any
decay, tracts in the middle, soft-masked and IUPAC bytes, reads shorter than a pattern — against the oracle's
ReadTelomereFilter.   python3 profiles/fuzz_reads.py [iterations] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tests import harness as H
from tests import seqgen
from tests.backends import OracleReadFilter, ProductReadFilter

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
motifs = ["TTAGGG", "CCCTAA", "TTTAGGG", "TTAGG", "TTAGGGG"]
total = kept = 0
for it in range(iters):
    c = motifs[int(rng.integers(0, len(motifs)))]
    cli = "--fastq-subset -c %s -x %d -k %d -d %d -l %d -y %.2f" % (
        c, int(rng.integers(0, 3)), int(rng.choice([3, 10, 50, 200])), int(rng.choice([5, 100, 500, 3000])),
        int(rng.choice([6, 12, 42, 300, 2000])), float(rng.choice([0.1, 0.3, 0.5, 0.666, 0.9, 1.0])))
    opts = H.parse_cli(cli)
    prod, orac = ProductReadFilter(opts), OracleReadFilter(opts)
    reads = []
    for r in range(int(rng.integers(100, 400))):
        kind = int(rng.integers(0, 8))
        n = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(40, 3000)), int(rng.integers(3000, 45000))]))
        s = bytearray(seqgen.random_dna(rng, n).tobytes())
        if kind in (1, 2, 3, 4) and n > 12:
            unit = c if kind % 2 else c[::-1].translate(str.maketrans("ACGT", "TGCA"))          # either orientation
            reps = int(rng.integers(2, 1500))
            t = seqgen.mutate(rng, seqgen.repeat_array(unit, reps), float(rng.choice([0.0, 0.01, 0.05, 0.2]))).tobytes()[:n]
            at = 0 if kind in (1, 2) else (n - len(t) if kind == 3 else int(rng.integers(0, n - len(t) + 1)))
            s[at:at + len(t)] = t
        if kind == 5 and n > 100:                                   # several short tracts a few hundred bases apart
            for _ in range(int(rng.integers(2, 12))):
                t = seqgen.repeat_array(c, int(rng.integers(2, 6))).tobytes()
                at = int(rng.integers(0, max(1, n - len(t))))
                s[at:at + len(t)] = t[:n - at]
        if kind == 6 and n > 10:
            for _ in range(int(rng.integers(1, 6))):
                s[int(rng.integers(0, n))] = ord(rng.choice(list("NRYKMnacgt")))
        reads.append(bytes(s))
    got, exp = prod.filter(reads), orac.filter(reads)
    bad = [i for i, (a, b) in enumerate(zip(got, exp)) if bool(a) != bool(b)]
    assert not bad, ("read filter differs from the oracle", cli, bad[:5], [len(reads[i]) for i in bad[:5]])
    total += len(reads); kept += int(sum(bool(x) for x in exp))
    if (it + 1) % 20 == 0:
        print("fuzz_reads: %d parameter sets ok (%d reads, %d kept)" % (it + 1, total, kept), flush=True)
print("fuzz_reads: all %d parameter sets (%d reads, %d kept) equal the oracle" % (iters, total, kept))
