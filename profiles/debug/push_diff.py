# debugging aid: first differing interstitial block of a push-ordered set, with the oracle's stream around it
import sys, numpy as np
sys.path.insert(0, ".")
from tests import harness as H, seqgen
from tests.backends import OracleBackend, ProductBackend, BLOCK_FIELDS
import tests.test_gpu_parity as T
which = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cli = T.PUSH_ORDER_GRID[which]
opts = H.parse_cli("x.fa " + cli)
prod, orac = ProductBackend(opts), OracleBackend(opts)
rng = np.random.default_rng(len(cli) * 7 + 1)
segs = []
for i, n in enumerate([5, 70, 999, 4095, 4096, 4097, 4200, 8192, 12290, 40000, 131072 + 17]):
    s = bytearray(seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev, telo_repeats=min(300, max(1, n // 30)), tvr_rate=0.02, n_its=4, iupac=0))
    for at in (4096, 8192, 12288, 36864):
        if n > at + 200:
            s[at - 150:at + 150] = (b"TTAGGG" * 50)[:300]
    segs.append((bytes(s), int(rng.integers(0, 10 ** 6)), False))
got = prod.scan_segments(segs)
for (s, ap, tips), g in zip(segs, got):
    e = orac.scan_segment(s, ap, tips)
    gb, eb = g["interstitial_blocks"], e["interstitial_blocks"]
    print("len", len(s), "its", len(gb), len(eb), "terminal", [(int(a), int(b), str(c)) for a, b, c in zip(e["terminal_blocks"]["start"], e["terminal_blocks"]["block_len"], e["terminal_blocks"]["block_label"])])
    n = min(len(gb), len(eb))
    bad = [i for i in range(n) if any(gb[f][i] != eb[f][i] for f in BLOCK_FIELDS)]
    if len(gb) != len(eb) or bad:
        i = bad[0] if bad else n
        for j in range(max(0, i - 2), min(n, i + 3)):
            print("  blk", j, "got", {f: int(gb[f][j]) for f in ("start", "block_len", "block_counts", "canonical_count")}, "exp", {f: int(eb[f][j]) for f in ("start", "block_len", "block_counts", "canonical_count")})
        am = e["all_matches"]
        st = int(eb["start"][i]) if i < len(eb) else int(gb["start"][i])
        idx = np.nonzero((am["position"] >= st - 40) & (am["position"] <= st + int(eb["block_len"][min(i, len(eb) - 1)]) + 60))[0]
        lo, hi = idx.min(), idx.max()
        print("  oracle stream [%d..%d]:" % (lo, hi), [(int(am["position"][q]), int(am["match_size"][q]), int(am["is_canonical"][q])) for q in range(lo, hi + 1)][:120])
        break
