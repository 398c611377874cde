#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer API (ts_scan_segments: H2D + kernels + D2H + host
post-processing incl. block calling).  Reported in DESIGN.md; never the bench `value`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tests import harness as H
from tests import seqgen
from tests.backends import ProductBackend

opts = H.parse_cli("x -c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 500 -r -g -e -m -i")
be = ProductBackend(opts)
rng = np.random.default_rng(1)
segs = [(seqgen.chromosome(rng, 50_000_000, telo_repeats=2000, n_its=10), 0, False) for _ in range(6)]
be.teloscope.scanSegments(segs[:1])
t0 = time.perf_counter()
res = be.teloscope.scanSegments(segs)
dt = time.perf_counter() - t0
nb = sum(len(s[0]) for s in segs)
nm = sum(len(r.allMatches) for r in res)
print("host API (ts_scan_segments + numpy copies): %.0f Mb in %.2f s = %.3f Gbases/s, %d matches, %d windows"
      % (nb / 1e6, dt, nb / dt / 1e9, nm, sum(len(r.windows) for r in res)))
