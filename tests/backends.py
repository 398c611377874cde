"""Scan backends for tests/harness.py: the CPU oracle and the HIP product, presenting the
same interface so every parity test reads `product == oracle` on identical inputs."""
import numpy as np


class OracleBackend:
    """oracle/teloscope_oracle.c through ctypes (the checker)."""

    def __init__(self, opts):
        from oracle import pyoracle as po
        self.po = po
        self.opts = opts
        self.patterns = po.expand_patterns(opts.raw_patterns, opts.edit_distance, opts.canonical_fwd)
        self.oracle = po.Oracle(opts.params(), self.patterns)

    def scan_segment(self, seq, abs_pos, tips_only):
        return self.oracle.scan_segment(seq, abs_pos, tips_only)

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
