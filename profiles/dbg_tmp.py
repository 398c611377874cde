import sys, numpy as np
sys.path.insert(0, '.')
from tests import harness as H
from tests.backends import OracleBackend, ProductBackend
from tests.seqgen import *
opts = H.parse_cli("x.fa -c TTAGGG -w 1000 -s 500 -r -g -e -m -i")
rng = np.random.default_rng(1)
seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=30000).tobytes())
seq = seq[:5000] + b"TTAGGG"*300 + seq[5000:]
for tips in (False, True):
    o = OracleBackend(opts).scan_segment(seq, 0, tips)
    g = ProductBackend(opts).scan_segment(seq, 0, tips)
    for name in ("fwd_matches", "rev_matches"):
        eo = set(int(m["position"]) for m in o[name]); eg = set(int(m["position"]) for m in g[name])
        miss = sorted(eo - eg); extra = sorted(eg - eo)
        print(tips, name, len(eo), len(eg), "missing", miss[:20], "extra", extra[:20])
        print("  missing mod 1008:", sorted(set(p % 1008 for p in miss))[:40])
        print("  missing mod 16:", sorted(set(p % 16 for p in miss)))
