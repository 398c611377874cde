"""What the device's push-ordered stream relies on, checked against the oracle (CPU only).

`scanSegment` pushes a match from the window whose own scan sees it (src/teloscope.cpp:485-509), so with pattern lengths two or
more apart under w > s `allMatches` is not quite in position order.  generic.hip (ts_general_compact_push) writes the stream in
that order from a closed form: the pushing window of a match that ends at e is key(e) = 0 for e < w - s, else (e - (w - s)) / s (one
window, key 0, when the segment is shorter than w) — monotone in the end position — and push order is the position-ordered stream
(position, then length) STABLY sorted by key.  Here the oracle's allMatches is compared with exactly that, and the two facts the
kernel's neighbour search uses are checked: records are only ever out of position order when they start fewer than
(longest - shortest) bases apart, and within one key the stream is in position order."""
import numpy as np
import pytest

from tests import harness as H
from tests import seqgen
from tests.backends import OracleBackend

GRID = [
    "-p TTAGGG,TTAGGGTTAGGG,TTAGGGTTAGGGTTAGGG -x 0 -w 1000 -s 500 -r -g -i",
    "-p TTAGGG,TTAGGGTTAGGG,TTAGGGTTAGGGTTAGGG -x 0 -w 100 -s 7 -r -g -i -k 12",
    "-p TTAGGG,TTAGGGTTAGGGTTAGGGTTAGGGTTAGGG -x 0 -w 64 -s 3 -g -i",
    "-p TTAGGG,TTAGGGTT,TTAGGGTTAGGGT -x 0 -w 200 -s 100 -g -i",
    "-x 0 -p " + ",".join(("TTAGGG" * 11)[:k] for k in (6, 9, 12, 15, 18, 24, 30, 36, 48, 63)) + " -w 300 -s 20 -g -i",
    "-p TTAGGG,TTTAGGGTTTAGGG -x 0 -w 1000 -s 997 -r -g -i",
]


def push_key(e, w, s, n):
    if n < w:
        return 0
    ov = w - s
    return 0 if e < ov else (e - ov) // s


@pytest.mark.parametrize("cli", GRID)
def test_all_matches_is_the_position_ordered_stream_stably_sorted_by_pushing_window(cli):
    opts = H.parse_cli("x.fa " + cli)
    orac = OracleBackend(opts)
    w, s = int(opts.window_size), int(opts.step)
    rng = np.random.default_rng(len(cli))
    n_inv = 0
    for n in (50, 999, 5000, 20011):
        seq = seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev, telo_repeats=min(200, max(1, n // 30)), tvr_rate=0.02, n_its=3)
        e = orac.scan_segment(seq, 0, False)
        am = e["all_matches"]
        pos = am["position"].astype(np.int64)
        ln = am["match_size"].astype(np.int64)
        if len(pos) == 0:
            continue
        spread = int(ln.max() - ln.min())
        key = np.array([push_key(int(p + l - 1), w, s, n) for p, l in zip(pos, ln)], dtype=np.int64)
        # the stream as pushed: keys never step back; inside a key, (position, length) ascends
        assert np.all(np.diff(key) >= 0)
        same = np.diff(key) == 0
        dp, dl = np.diff(pos), np.diff(ln)
        assert np.all((dp[same] > 0) | ((dp[same] == 0) & (dl[same] > 0)))
        # ... which is the (position, length)-ordered stream stably sorted by key
        order = np.lexsort((ln, pos))
        by_pos = np.stack([pos[order], ln[order], key[order]], axis=1)
        stable = by_pos[np.argsort(by_pos[:, 2], kind="stable")]
        assert np.array_equal(stable[:, 0], pos) and np.array_equal(stable[:, 1], ln)
        # a record that lies ahead of an earlier one in the stream starts fewer than `spread` bases before it
        back = dp < 0
        n_inv += int(back.sum())
        assert np.all(-dp[back] < max(spread, 1))
    if "-s 997" not in cli:
        assert n_inv > 0, "the set was meant to produce a stream that is not in position order"
