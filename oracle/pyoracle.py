"""ctypes binding of the CPU oracle (oracle/teloscope_oracle.c).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by anything under teloscope_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libteloscope_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "teloscope_oracle.c")
    hdr = os.path.join(_HERE, "teloscope_oracle.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


class Params(C.Structure):
    _fields_ = [
        ("window_size", C.c_uint32), ("step", C.c_uint32), ("terminal_limit", C.c_uint32),
        ("max_match_dist", C.c_uint16), ("min_block_len", C.c_uint16),
        ("max_block_dist", C.c_uint16), ("min_block_counts", C.c_uint16),
        ("min_block_density", C.c_float), ("canonical_size", C.c_uint16),
        ("out_gc", C.c_uint8), ("out_entropy", C.c_uint8), ("out_matches", C.c_uint8),
        ("reserved", C.c_uint8),
    ]


class Pattern(C.Structure):
    _fields_ = [("seq", C.c_char * 64), ("len", C.c_uint8), ("is_forward", C.c_uint8),
                ("is_canonical", C.c_uint8), ("ambiguous", C.c_uint8)]


class Match(C.Structure):
    _fields_ = [("position", C.c_uint64), ("match_size", C.c_uint16),
                ("is_forward", C.c_uint8), ("is_canonical", C.c_uint8), ("pad", C.c_uint32)]


class Window(C.Structure):
    _fields_ = [("window_start", C.c_uint64), ("current_window_size", C.c_uint32),
                ("nucleotide_counts", C.c_uint32 * 4), ("gc_content", C.c_float),
                ("shannon_entropy", C.c_float),
                ("canonical_counts", C.c_uint16), ("non_canonical_counts", C.c_uint16),
                ("fwd_counts", C.c_uint16), ("rev_counts", C.c_uint16),
                ("canonical_covered", C.c_uint32), ("non_canonical_covered", C.c_uint32),
                ("fwd_covered", C.c_uint32), ("rev_covered", C.c_uint32),
                ("has_can_dimer", C.c_uint8), ("pad", C.c_uint8 * 3)]


class Block(C.Structure):
    _fields_ = [("start", C.c_uint64), ("block_len", C.c_uint32), ("block_counts", C.c_uint32),
                ("forward_count", C.c_uint32), ("reverse_count", C.c_uint32),
                ("canonical_count", C.c_uint32), ("non_canonical_count", C.c_uint32),
                ("total_covered", C.c_uint32), ("fwd_covered", C.c_uint32),
                ("can_covered", C.c_uint32), ("has_valid_or", C.c_uint8),
                ("is_longest", C.c_uint8), ("block_label", C.c_char), ("pad", C.c_uint8)]


class Segment(C.Structure):
    _fields_ = [("windows", C.POINTER(Window)), ("n_windows", C.c_size_t),
                ("terminal_blocks", C.POINTER(Block)), ("n_terminal_blocks", C.c_size_t),
                ("interstitial_blocks", C.POINTER(Block)), ("n_interstitial_blocks", C.c_size_t),
                ("canonical_matches", C.POINTER(Match)), ("n_canonical_matches", C.c_size_t),
                ("non_canonical_matches", C.POINTER(Match)), ("n_non_canonical_matches", C.c_size_t),
                ("fwd_matches", C.POINTER(Match)), ("n_fwd_matches", C.c_size_t),
                ("rev_matches", C.POINTER(Match)), ("n_rev_matches", C.c_size_t),
                ("all_matches", C.POINTER(Match)), ("n_all_matches", C.c_size_t)]


MATCH_DT = np.dtype(Match)
WINDOW_DT = np.dtype(Window)
BLOCK_DT = np.dtype(Block)

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.tso_revcom.argtypes = [C.c_char_p, C.c_char_p]
        L.tso_expand_patterns.restype = C.POINTER(Pattern)
        L.tso_expand_patterns.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.POINTER(C.c_size_t)]
        L.tso_create.restype = C.c_void_p
        L.tso_create.argtypes = [C.POINTER(Params), C.POINTER(Pattern), C.c_size_t]
        L.tso_destroy.argtypes = [C.c_void_p]
        L.tso_longest_pattern.restype = C.c_uint16
        L.tso_longest_pattern.argtypes = [C.c_void_p]
        L.tso_scan_segment.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_int,
                                       C.POINTER(Segment)]
        L.tso_free_segment.argtypes = [C.POINTER(Segment)]
        L.tso_label_terminal_blocks.argtypes = [C.POINTER(Block), C.c_size_t, C.c_uint16,
                                                C.c_uint64, C.c_uint32, C.c_char_p]
        L.tso_read_filter_params.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(Params)]
        L.tso_read_filter_matches.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        L.tso_gc_content.restype = C.c_float
        L.tso_gc_content.argtypes = [C.POINTER(C.c_uint32), C.c_uint32]
        L.tso_shannon_entropy.restype = C.c_float
        L.tso_shannon_entropy.argtypes = [C.POINTER(C.c_uint32), C.c_uint32]
        L.tso_bench_scan.restype = C.c_uint64
        L.tso_bench_scan.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64,
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.free = C.CDLL(None).free
        L.free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def revcom(s):
    out = C.create_string_buffer(len(s) + 1)
    lib().tso_revcom(s.encode(), out)
    return out.value.decode()


def expand_patterns(raw_patterns, edit_distance, canonical_fwd):
    """-> list of (seq, is_forward, is_canonical, ambiguous)"""
    n = C.c_size_t(0)
    csv = ",".join(raw_patterns).encode()
    arr = lib().tso_expand_patterns(csv, int(edit_distance), canonical_fwd.encode(), C.byref(n))
    out = [(arr[i].seq.decode(), bool(arr[i].is_forward), bool(arr[i].is_canonical),
            bool(arr[i].ambiguous)) for i in range(n.value)]
    lib().free(arr)
    return out


def _copy(ptr, n, dt):
    if n == 0:
        return np.zeros(0, dtype=dt)
    buf = C.string_at(C.cast(ptr, C.c_void_p), n * dt.itemsize)
    return np.frombuffer(buf, dtype=dt).copy()


class Oracle:
    """One Teloscope instance (params + trie) of the CPU oracle."""

    def __init__(self, params, patterns):
        """params: dict of tso_params fields; patterns: list of (seq, is_forward, is_canonical[, _])"""
        self.p = Params(**params)
        arr = (Pattern * max(1, len(patterns)))()
        for i, pt in enumerate(patterns):
            arr[i].seq = pt[0].encode()
            arr[i].len = len(pt[0])
            arr[i].is_forward = int(pt[1])
            arr[i].is_canonical = int(pt[2])
        self._pats = arr
        self.n_patterns = len(patterns)
        self.ctx = lib().tso_create(C.byref(self.p), arr, len(patterns))

    def close(self):
        if self.ctx:
            lib().tso_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def longest(self):
        return lib().tso_longest_pattern(self.ctx)

    def scan_segment(self, seq, abs_pos=0, tips_only=False):
        if isinstance(seq, str):
            seq = seq.encode()
        s = Segment()
        lib().tso_scan_segment(self.ctx, seq, len(seq), abs_pos, int(tips_only), C.byref(s))
        out = {
            "windows": _copy(s.windows, s.n_windows, WINDOW_DT),
            "terminal_blocks": _copy(s.terminal_blocks, s.n_terminal_blocks, BLOCK_DT),
            "interstitial_blocks": _copy(s.interstitial_blocks, s.n_interstitial_blocks, BLOCK_DT),
            "canonical_matches": _copy(s.canonical_matches, s.n_canonical_matches, MATCH_DT),
            "non_canonical_matches": _copy(s.non_canonical_matches, s.n_non_canonical_matches, MATCH_DT),
            "fwd_matches": _copy(s.fwd_matches, s.n_fwd_matches, MATCH_DT),
            "rev_matches": _copy(s.rev_matches, s.n_rev_matches, MATCH_DT),
            "all_matches": _copy(s.all_matches, s.n_all_matches, MATCH_DT),
        }
        lib().tso_free_segment(C.byref(s))
        return out

    def read_filter_matches(self, seq):
        if isinstance(seq, str):
            seq = seq.encode()
        return bool(lib().tso_read_filter_matches(self.ctx, seq, len(seq)))

    def bench_scan(self, seq):
        nw, nm = C.c_uint64(0), C.c_uint64(0)
        h = lib().tso_bench_scan(self.ctx, seq, len(seq), C.byref(nw), C.byref(nm))
        return h, nw.value, nm.value


def read_filter_params(params, min_block_len_set):
    pin, pout = Params(**params), Params()
    lib().tso_read_filter_params(C.byref(pin), int(min_block_len_set), C.byref(pout))
    return {f[0]: getattr(pout, f[0]) for f in Params._fields_}


def label_terminal_blocks(blocks, gaps, path_size, terminal_limit):
    """blocks: np array BLOCK_DT -> (sorted blocks, label str, scaffold type int)"""
    n = len(blocks)
    arr = (Block * max(1, n))()
    if n:
        C.memmove(arr, np.ascontiguousarray(blocks).ctypes.data, n * BLOCK_DT.itemsize)
    lab = C.create_string_buffer(2 * n + 2)
    t = lib().tso_label_terminal_blocks(arr, n, gaps, path_size, terminal_limit, lab)
    out = np.frombuffer(C.string_at(arr, n * BLOCK_DT.itemsize), dtype=BLOCK_DT).copy() if n else \
        np.zeros(0, BLOCK_DT)
    return out, lab.value.decode(), t


def gc_content(counts, size):
    a = (C.c_uint32 * 4)(*[int(x) for x in counts])
    return np.float32(lib().tso_gc_content(a, int(size)))


def shannon_entropy(counts, size):
    a = (C.c_uint32 * 4)(*[int(x) for x in counts])
    return np.float32(lib().tso_shannon_entropy(a, int(size)))
