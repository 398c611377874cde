"""Scan backends for tests/harness.py: the CPU oracle and the HIP product, presenting the
same interface so every parity test reads `product == oracle` on identical inputs."""
import numpy as np


class OracleBackend:
    """oracle/teloscope_oracle.c through ctypes (the checker)."""

    def __init__(self, opts, patterns=None):
        """patterns: optional explicit [(seq, is_forward, is_canonical)] (what the Teloscope ctor is
        given); default = the oracle's own expandPatternsWithOrientation."""
        from oracle import pyoracle as po
        self.po = po
        self.opts = opts
        self.patterns = patterns if patterns is not None else \
            po.expand_patterns(opts.raw_patterns, opts.edit_distance, opts.canonical_fwd)
        self.ambiguous = any(len(p) > 3 and p[3] for p in self.patterns)
        self.oracle = po.Oracle(opts.params(), self.patterns)

    def scan_segment(self, seq, abs_pos, tips_only):
        return self.oracle.scan_segment(seq, abs_pos, tips_only)

    def with_ambiguous_orientation_from(self, product_patterns):
        """Some pattern sets produce the same expanded k-mer from both strands (a k-mer that is its own reverse
        complement, or k = 5 with two mismatches): which copy the reference keeps depends on how std::sort orders equal
        keys (src/tools.cpp:275-280), i.e. is unspecified.  The oracle flags exactly those entries (`ambiguous`).  This
        returns an oracle whose pattern list is ITS OWN expansion, with the orientation flag of the ambiguous entries — and
        of those only — taken from the product; every unambiguous entry must agree between the two expansions, or the
        call fails.  (Rounds 1-2 handed the oracle the product's whole list, which compared the product's orientation
        flags with themselves.)"""
        prod = {p: (f, c) for p, f, c in product_patterns}
        assert sorted(prod) == sorted(p[0] for p in self.patterns), "product and oracle expand to different pattern sets"
        merged = []
        for p, f, c, amb in self.patterns:
            pf, pc = prod[p]
            assert pc == c, "canonical flag of %s differs" % p
            if amb:
                merged.append((p, pf, c))
            else:
                assert pf == f, "orientation of the unambiguous pattern %s differs: product %s, oracle %s" % (p, pf, f)
                merged.append((p, f, c))
        return OracleBackend(self.opts, patterns=merged)

    def empty_blocks(self):
        return np.zeros(0, dtype=self.po.BLOCK_DT)

    def label_terminal_blocks(self, blocks, gaps, path_size, terminal_limit):
        return self.po.label_terminal_blocks(blocks, gaps, path_size, terminal_limit)


class OracleReadFilter:
    """ReadTelomereFilter on the oracle."""

    def __init__(self, opts):
        from oracle import pyoracle as po
        self.patterns = po.expand_patterns(opts.raw_patterns, opts.edit_distance, opts.canonical_fwd)
        rp = po.read_filter_params(opts.params(), opts.min_block_len_set)
        rp.pop("reserved", None)
        self.oracle = po.Oracle(rp, self.patterns)

    def filter(self, seqs):
        return [self.oracle.read_filter_matches(s) for s in seqs]


def _user_input(opts):
    from teloscope_amd.cli import user_input
    return user_input(opts)


class ProductBackend:
    """libteloscan.so (HIP, gfx950) through teloscope_amd: the thing under test."""

    def __init__(self, opts):
        import teloscope_amd as ta
        self.ta = ta
        self.ui = _user_input(opts)
        self.patterns = [(p, f, p in (opts.canonical_fwd, opts.canonical_rev)) for p, f in self.ui.patternInfo]
        self.teloscope = ta.Teloscope(self.ui)

    def scan_segment(self, seq, abs_pos, tips_only):
        return segment_as_dict(self.teloscope.scanSegment(seq, abs_pos, tips_only))

    def scan_segments(self, segs):
        return [segment_as_dict(s) for s in self.teloscope.scanSegments(segs)]

    def empty_blocks(self):
        from teloscope_amd import _capi
        return np.zeros(0, dtype=_capi.BLOCK_DT)

    def label_terminal_blocks(self, blocks, gaps, path_size, terminal_limit):
        return self.ta.Teloscope.labelTerminalBlocks(blocks, gaps, path_size, terminal_limit)


class ProductReadFilter:
    def __init__(self, opts):
        import teloscope_amd as ta
        self.rf = ta.ReadTelomereFilter(_user_input(opts))

    def filter(self, seqs):
        return self.rf.matchesBatch(seqs)


def segment_as_dict(sd):
    return dict(windows=sd.windows, terminal_blocks=sd.terminalBlocks,
                interstitial_blocks=sd.interstitialBlocks, canonical_matches=sd.canonicalMatches,
                non_canonical_matches=sd.nonCanonicalMatches, fwd_matches=sd.fwdMatches,
                rev_matches=sd.revMatches, all_matches=sd.allMatches)


MATCH_FIELDS = ("position", "match_size")
WINDOW_FIELDS = ("window_start", "current_window_size", "nucleotide_counts", "canonical_covered",
                 "non_canonical_covered", "fwd_covered", "rev_covered")
BLOCK_FIELDS = ("start", "block_len", "block_counts", "forward_count", "reverse_count",
                "canonical_count", "non_canonical_count", "total_covered", "fwd_covered",
                "can_covered", "has_valid_or", "block_label")


def assert_segment_equal(got, exp, tips_only, float_fields=True, ctx=""):
    """Bit-exact comparison of a product SegmentData (dict) with the oracle's."""
    def cmp_matches(name):
        g, e = got[name], exp[name]
        assert len(g) == len(e), "%s %s: %d vs %d matches" % (ctx, name, len(g), len(e))
        for f in MATCH_FIELDS:
            assert np.array_equal(g[f], e[f]), "%s %s.%s differs" % (ctx, name, f)
        gf = (g["flags"] & 1) != 0
        gc = (g["flags"] & 2) != 0
        assert np.array_equal(gf, e["is_forward"] != 0), "%s %s.is_forward differs" % (ctx, name)
        assert np.array_equal(gc, e["is_canonical"] != 0), "%s %s.is_canonical differs" % (ctx, name)

    for name in ("fwd_matches", "rev_matches", "all_matches", "canonical_matches", "non_canonical_matches"):
        cmp_matches(name)
    gw, ew = got["windows"], exp["windows"]
    assert len(gw) == len(ew), "%s windows: %d vs %d" % (ctx, len(gw), len(ew))
    for f in WINDOW_FIELDS:
        assert np.array_equal(gw[f], ew[f]), "%s windows.%s differs" % (ctx, f)
    if float_fields and len(gw):
        # float32 metrics are evaluated on the host from identical integers: expected bit-equal,
        # north_star tolerance is 1e-6
        assert np.max(np.abs(gw["gc_content"].astype(np.float64) - ew["gc_content"])) <= 1e-6, ctx
        assert np.max(np.abs(gw["shannon_entropy"].astype(np.float64) - ew["shannon_entropy"])) <= 1e-6, ctx
        # (the same float32 expressions on the same integers: in fact bit-equal — the product looks the entropy terms of
        # full-size windows up in a table built by those expressions)
        assert np.array_equal(gw["gc_content"].astype(np.float32), np.asarray(ew["gc_content"], dtype=np.float32)), ctx + " gc_content bits"
        assert np.array_equal(gw["shannon_entropy"].astype(np.float32), np.asarray(ew["shannon_entropy"], dtype=np.float32)), ctx + " shannon_entropy bits"
    for name in ("terminal_blocks", "interstitial_blocks"):
        g, e = got[name], exp[name]
        assert len(g) == len(e), "%s %s: %d vs %d" % (ctx, name, len(g), len(e))
        for f in BLOCK_FIELDS:
            assert np.array_equal(g[f], e[f]), "%s %s.%s differs" % (ctx, name, f)


def assert_visible_view_equal(got, exp, tips, cnt, ctx):
    """got: teloscope_amd.SegmentData built from a visible-view ts_segment_out; exp: the oracle's dict."""
    gw, ew = got.windows, exp["windows"]
    assert len(gw) == len(ew), "%s windows: %d vs %d" % (ctx, len(gw), len(ew))
    for f in WINDOW_FIELDS:
        assert np.array_equal(gw[f], ew[f]), "%s windows.%s differs" % (ctx, f)
    if len(gw):
        assert np.max(np.abs(gw["gc_content"].astype(np.float64) - ew["gc_content"])) <= 1e-6, ctx
        assert np.max(np.abs(gw["shannon_entropy"].astype(np.float64) - ew["shannon_entropy"])) <= 1e-6, ctx
    for name, g in (("terminal_blocks", got.terminalBlocks), ("interstitial_blocks", got.interstitialBlocks)):
        e = exp[name]
        assert len(g) == len(e), "%s %s: %d vs %d" % (ctx, name, len(g), len(e))
        for f in BLOCK_FIELDS:
            assert np.array_equal(g[f], e[f]), "%s %s.%s differs" % (ctx, name, f)
    for name, g in (("canonical_matches", got.canonicalMatches), ("non_canonical_matches", got.nonCanonicalMatches)):
        e = exp[name]
        assert len(g) == len(e), "%s %s: %d vs %d" % (ctx, name, len(g), len(e))
        for f in MATCH_FIELDS:
            assert np.array_equal(g[f], e[f]), "%s %s.%s differs" % (ctx, name, f)
        assert np.array_equal((g["flags"] & 1) != 0, e["is_forward"] != 0), "%s %s.is_forward" % (ctx, name)
    if not tips:
        assert len(got._m) == len(exp["canonical_matches"]) + len(exp["non_canonical_matches"]), ctx
    if cnt is not None:
        if tips:
            assert (cnt.n_windows, cnt.n_matches, cnt.n_forward) == (0, len(exp["fwd_matches"]) + len(exp["rev_matches"]), len(exp["fwd_matches"])), ctx
        else:
            assert (cnt.n_windows, cnt.n_matches, cnt.n_canonical, cnt.n_forward) == \
                (len(ew), len(exp["all_matches"]), len(exp["canonical_matches"]), len(exp["fwd_matches"])), ctx
