"""ctypes binding of libteloscan.so (the C-ABI declared in include/teloscan.h).

The library is the product; this module only maps its structs and entry points.  It fails
loudly (ImportError) when the shared object is missing — there is no Python or CPU
fallback for the scan path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TELOSCAN_LIB") or os.path.join(_HERE, "libteloscan.so")

TS_OK = 0
TS_ERR_INVALID_ARG, TS_ERR_NO_DEVICE, TS_ERR_HIP, TS_ERR_ALLOC, TS_ERR_UNSUPPORTED, TS_ERR_STATE = \
    -1, -2, -3, -4, -5, -6
MATCH_FORWARD, MATCH_CANONICAL, MATCH_TERMINAL = 1, 2, 4


class Pattern(C.Structure):
    _fields_ = [("seq", C.c_char * 64), ("len", C.c_uint8), ("is_forward", C.c_uint8),
                ("is_canonical", C.c_uint8), ("reserved", C.c_uint8)]


class Params(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("window_size", C.c_uint32), ("step", C.c_uint32),
                ("terminal_limit", C.c_uint32), ("max_match_dist", C.c_uint16),
                ("min_block_len", C.c_uint16), ("max_block_dist", C.c_uint16),
                ("min_block_counts", C.c_uint16), ("min_block_density", C.c_float),
                ("canonical_size", C.c_uint16), ("out_gc", C.c_uint8), ("out_entropy", C.c_uint8),
                ("out_matches", C.c_uint8), ("out_its", C.c_uint8), ("fold_case", C.c_uint8),
                ("reserved0", C.c_uint8), ("device", C.c_int32), ("reserved1", C.c_uint32)]


class Match(C.Structure):
    _fields_ = [("position", C.c_uint64), ("match_size", C.c_uint16), ("flags", C.c_uint8),
                ("reserved", C.c_uint8 * 5)]


class Window(C.Structure):
    _fields_ = [("window_start", C.c_uint64), ("current_window_size", C.c_uint32),
                ("nucleotide_counts", C.c_uint32 * 4), ("gc_content", C.c_float),
                ("shannon_entropy", C.c_float), ("canonical_covered", C.c_uint32),
                ("non_canonical_covered", C.c_uint32), ("fwd_covered", C.c_uint32),
                ("rev_covered", C.c_uint32), ("reserved", C.c_uint32)]


class Block(C.Structure):
    _fields_ = [("start", C.c_uint64), ("block_len", C.c_uint32), ("block_counts", C.c_uint32),
                ("forward_count", C.c_uint32), ("reverse_count", C.c_uint32),
                ("canonical_count", C.c_uint32), ("non_canonical_count", C.c_uint32),
                ("total_covered", C.c_uint32), ("fwd_covered", C.c_uint32),
                ("can_covered", C.c_uint32), ("has_valid_or", C.c_uint8),
                ("is_longest", C.c_uint8), ("block_label", C.c_char), ("reserved", C.c_uint8)]


class SegmentIn(C.Structure):
    _fields_ = [("seq", C.c_char_p), ("len", C.c_uint64), ("abs_pos", C.c_uint64),
                ("tips_only", C.c_uint8), ("input_format", C.c_uint8), ("reserved", C.c_uint8 * 2), ("n_pieces", C.c_uint32)]


TS_INPUT_BASES, TS_INPUT_TEXT_PIECES, TS_INPUT_PACKED2 = 0, 1, 2


class PackedRun(C.Structure):
    _fields_ = [("start", C.c_uint64), ("len", C.c_uint64)]


class PackedSeq(C.Structure):
    _fields_ = [("codes", C.c_void_p), ("runs", C.c_void_p), ("n_runs", C.c_uint64)]


def pack_sequence(seq, fold_case):
    """bytes -> (PackedSeq, keep-alive objects): 2-bit codes + invalid runs through ts_pack_bases (pieces of < 2^32 bases,
    the run positions made segment-relative), the TS_INPUT_PACKED2 form of a segment."""
    n = len(seq)
    codes = np.zeros((n + 3) // 4 + 64, dtype=np.uint8)
    runs = []
    piece = 1 << 28                                                  # a multiple of 4: pieces start on byte boundaries
    buf = C.create_string_buffer(seq, n) if not isinstance(seq, (bytes, bytearray)) else seq
    # the bases are handed over where they lie (no copy per piece)
    if isinstance(buf, bytearray):
        view = (C.c_char * n).from_buffer(buf)
        base = C.addressof(view)
    elif isinstance(buf, bytes):
        base = C.cast(C.c_char_p(buf), C.c_void_p).value or 0
    else:
        base = C.addressof(buf)
    for a in range(0, n, piece):
        m = min(piece, n - a)
        cap = 1 << 16                                                # runs are rare: start small, grow to what the call reports
        while True:
            rr = np.empty((cap, 2), dtype=np.uint32)
            nr = C.c_uint64(0)
            rc = lib().ts_pack_bases(C.cast(C.c_void_p(base + a), C.c_char_p), m, int(bool(fold_case)),
                                     C.c_void_p(codes.ctypes.data + a // 4), C.c_void_p(rr.ctypes.data), cap, C.byref(nr))
            if rc == TS_OK:
                break
            if int(nr.value) <= cap:                                 # (more runs than `cap` is the one failure worth a retry)
                raise TeloscanError(rc, "ts_pack_bases failed")
            cap = int(nr.value) + 16
        for s0, ln in rr[:int(nr.value)]:
            if runs and runs[-1][0] + runs[-1][1] == a + int(s0):
                runs[-1][1] += int(ln)
            else:
                runs.append([a + int(s0), int(ln)])
    arr = (PackedRun * max(1, len(runs)))()
    for i, (s0, ln) in enumerate(runs):
        arr[i].start, arr[i].len = s0, ln
    ps = PackedSeq(C.c_void_p(codes.ctypes.data), C.cast(arr, C.c_void_p), len(runs))
    return ps, (codes, arr)


class TextPiece(C.Structure):
    _fields_ = [("text", C.c_char_p), ("text_len", C.c_uint64), ("n_bases", C.c_uint64)]


class SegmentOut(C.Structure):
    _fields_ = [("windows", C.POINTER(Window)), ("n_windows", C.c_uint64),
                ("matches", C.POINTER(Match)), ("n_matches", C.c_uint64),
                ("terminal_blocks", C.POINTER(Block)), ("n_terminal_blocks", C.c_uint64),
                ("interstitial_blocks", C.POINTER(Block)), ("n_interstitial_blocks", C.c_uint64)]


class SegmentCounts(C.Structure):
    _fields_ = [("n_windows", C.c_uint64), ("n_matches", C.c_uint64), ("n_canonical", C.c_uint64),
                ("n_forward", C.c_uint64)]


class BatchInfo(C.Structure):
    _fields_ = [("n_segments", C.c_uint64), ("total_bases", C.c_uint64), ("input_bytes", C.c_uint64),
                ("n_windows", C.c_uint64), ("n_tiles", C.c_uint64), ("match_capacity", C.c_uint64),
                ("n_matches", C.c_uint64), ("algorithmic_bytes", C.c_uint64),
                ("last_kernel_ms", C.c_double), ("avg_kernel_ms", C.c_double),
                ("kernel_launches", C.c_uint64)]


class TileInfo(C.Structure):
    _fields_ = [("seg_index", C.c_uint64), ("seg_offset", C.c_uint64), ("first_window", C.c_uint64),
                ("n_windows", C.c_uint32), ("owned_bases", C.c_uint32)]


class RangeInfo(C.Structure):
    _fields_ = [("tile_begin", C.c_uint64), ("tile_end", C.c_uint64), ("window_begin", C.c_uint64),
                ("window_end", C.c_uint64), ("input_begin", C.c_uint64), ("input_end", C.c_uint64),
                ("bases", C.c_uint64)]


class ShardInfo(C.Structure):
    _fields_ = [("n_parts", C.c_uint32), ("part", C.c_uint32), ("own_begin", C.c_uint64), ("own_end", C.c_uint64),
                ("ext_begin", C.c_uint64), ("ext_end", C.c_uint64), ("window_begin", C.c_uint64),
                ("window_end", C.c_uint64), ("input_begin", C.c_uint64), ("input_end", C.c_uint64),
                ("bases", C.c_uint64), ("seg_begin", C.c_uint64), ("seg_end", C.c_uint64), ("msg_bytes", C.c_uint64),
                ("visible_capacity", C.c_uint64), ("block_capacity", C.c_uint32), ("window_bytes", C.c_uint32),
                ("visible_bytes", C.c_uint32), ("context_tiles", C.c_uint32)]


class ShardStatus(C.Structure):
    _fields_ = [("part", C.c_uint32), ("n_parts", C.c_uint32), ("flags", C.c_uint32), ("n_blocks", C.c_uint32),
                ("n_visible", C.c_uint64), ("visible_capacity", C.c_uint64), ("block_capacity", C.c_uint32),
                ("scale_factor_needed", C.c_uint32), ("msg_bytes", C.c_uint64)]


SHARD_RETRY_SYNC, SHARD_RETRY_GROW, SHARD_NEED_FULL = 1, 2, 3
SHARD_OVERFLOW_VISIBLE, SHARD_OVERFLOW_BLOCKS, SHARD_OVERFLOW_SCAN, SHARD_OUT_OF_CONTEXT = 1, 2, 4, 8
DEVICE_NONE = -2            # TS_DEVICE_NONE: planning-only context
MATCH_DT = np.dtype(Match)
TILE_DT = np.dtype(TileInfo)
WINDOW_DT = np.dtype(Window)
BLOCK_DT = np.dtype(Block)

# every symbol include/teloscan.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "ts_abi_version", "ts_last_error", "ts_device_count", "ts_canonical_orientation",
    "ts_expand_patterns", "ts_free_patterns", "ts_create", "ts_destroy", "ts_uses_fast_path",
    "ts_scan_segments", "ts_scan_segments_blocks", "ts_free_segments", "ts_create_read_filter", "ts_filter_reads",
    "ts_label_terminal_blocks", "ts_gc_content", "ts_shannon_entropy", "ts_batch_create",
    "ts_batch_destroy", "ts_batch_segment_offset", "ts_batch_input_ptr", "ts_batch_upload",
    "ts_batch_scan", "ts_batch_sync", "ts_batch_get_info", "ts_batch_windows_ptr",
    "ts_batch_matches_ptr", "ts_batch_download", "ts_batch_download_blocks", "ts_batch_segment_summary",
    "ts_batch_get_tiles", "ts_batch_range_info", "ts_batch_partition", "ts_batch_restrict", "ts_batch_bind_results",
    "ts_batch_export", "ts_batch_adopt", "ts_batch_tile_stats_ptr", "ts_filter_reads_multi", "ts_batch_read_pass",
    "ts_batch_wire16_ok", "ts_wire_widen_u16", "ts_takes_text_input", "ts_bind_thread_to_device",
    "ts_batch_shard_info", "ts_batch_restrict_shard", "ts_batch_set_shard_scale", "ts_batch_pack_shard",
    "ts_shard_peek", "ts_shards_finalize", "ts_scan_segments_multi", "ts_batch_read_pass_status", "ts_pack_bases",
    "ts_batch_set_emit", "ts_exchange_unique_id", "ts_exchange_last_error", "ts_exchange_create", "ts_exchange_destroy",
    "ts_exchange_gather", "ts_box_probe", "ts_batch_bind_shard_message", "ts_refresh_env", "ts_streams_concurrent", "ts_batch_wait_scan", "ts_batch_set_timing", "ts_batch_set_record_bits",
]


def build(force=False):
    """Compiles libteloscan.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    inc = os.path.join(os.path.dirname(_HERE), "include", "teloscan.h")
    deps = [os.path.join(src_dir, f) for f in os.listdir(src_dir)
            if f.endswith((".hip", ".cpp", ".h", ".hpp"))] + [inc]
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(d) > os.path.getmtime(LIB_PATH) for d in deps)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", src_dir])
    return LIB_PATH


_lib = None


def _share_torch_hip_runtime():
    """One process, one HIP runtime.  libteloscan.so needs libamdhip64.so.7; PyTorch-ROCm ships its own copy under
    torch/lib with the same SONAME, and whichever is loaded first serves both.  When torch comes second it finds a
    runtime it was not built against and reports "No HIP GPUs are available" — so, if PyTorch is installed but not
    imported yet, its runtime libraries are loaded first (what `import torch` would have done); without PyTorch the
    system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libteloscan.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C teloscope_amd/csrc`; teloscope_amd has no CPU fallback." % LIB_PATH)
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.ts_abi_version.restype = C.c_int
    L.ts_last_error.restype = C.c_char_p
    L.ts_last_error.argtypes = [C.c_void_p]
    L.ts_device_count.restype = C.c_int
    L.ts_canonical_orientation.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    L.ts_expand_patterns.argtypes = [C.c_char_p, C.c_int, C.c_char_p,
                                     C.POINTER(C.POINTER(Pattern)), C.POINTER(C.c_size_t)]
    L.ts_free_patterns.argtypes = [C.POINTER(Pattern)]
    L.ts_create.restype = C.c_void_p
    L.ts_create.argtypes = [C.POINTER(Params), C.POINTER(Pattern), C.c_size_t]
    L.ts_create_read_filter.restype = C.c_void_p
    L.ts_create_read_filter.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(Pattern), C.c_size_t]
    L.ts_destroy.argtypes = [C.c_void_p]
    L.ts_uses_fast_path.argtypes = [C.c_void_p]
    L.ts_takes_text_input.argtypes = [C.c_void_p, C.c_int]
    L.ts_bind_thread_to_device.argtypes = [C.c_void_p]
    L.ts_pack_bases.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.ts_pack_bases.restype = C.c_int
    L.ts_scan_segments.argtypes = [C.c_void_p, C.POINTER(SegmentIn), C.c_size_t, C.POINTER(SegmentOut)]
    L.ts_scan_segments_blocks.argtypes = [C.c_void_p, C.POINTER(SegmentIn), C.c_size_t, C.POINTER(SegmentOut),
                                          C.POINTER(SegmentCounts)]
    L.ts_free_segments.argtypes = [C.POINTER(SegmentOut), C.c_size_t]
    L.ts_filter_reads.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64),
                                  C.c_size_t, C.POINTER(C.c_uint8)]
    L.ts_label_terminal_blocks.argtypes = [C.POINTER(Block), C.c_size_t, C.c_uint16, C.c_uint64,
                                           C.c_uint32, C.c_char_p, C.POINTER(C.c_int)]
    L.ts_gc_content.restype = C.c_float
    L.ts_gc_content.argtypes = [C.POINTER(C.c_uint32), C.c_uint32]
    L.ts_shannon_entropy.restype = C.c_float
    L.ts_shannon_entropy.argtypes = [C.POINTER(C.c_uint32), C.c_uint32]
    L.ts_batch_create.restype = C.c_void_p
    L.ts_batch_create.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                  C.c_size_t, C.c_int, C.c_uint64]
    L.ts_batch_destroy.argtypes = [C.c_void_p]
    L.ts_batch_segment_offset.restype = C.c_uint64
    L.ts_batch_segment_offset.argtypes = [C.c_void_p, C.c_size_t]
    L.ts_batch_input_ptr.restype = C.c_void_p
    L.ts_batch_input_ptr.argtypes = [C.c_void_p]
    L.ts_batch_upload.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p]
    L.ts_batch_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ts_batch_sync.argtypes = [C.c_void_p]
    L.ts_batch_get_info.argtypes = [C.c_void_p, C.POINTER(BatchInfo)]
    L.ts_batch_windows_ptr.restype = C.c_void_p
    L.ts_batch_windows_ptr.argtypes = [C.c_void_p]
    L.ts_batch_matches_ptr.restype = C.c_void_p
    L.ts_batch_matches_ptr.argtypes = [C.c_void_p]
    L.ts_batch_download.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(SegmentOut)]
    L.ts_batch_download_blocks.argtypes = [C.c_void_p, C.POINTER(SegmentOut)]
    L.ts_batch_segment_summary.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ts_batch_get_tiles.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    L.ts_batch_range_info.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(RangeInfo)]
    L.ts_batch_partition.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.ts_batch_restrict.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    L.ts_batch_bind_results.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    if hasattr(L, "ts_batch_set_emit"):                # (absent from a round-3 build loaded through TELOSCAN_LIB for an A/B run)
        L.ts_batch_set_emit.argtypes = [C.c_void_p, C.c_int]
    L.ts_batch_export.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.ts_batch_adopt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.ts_batch_tile_stats_ptr.restype = C.c_void_p
    L.ts_batch_tile_stats_ptr.argtypes = [C.c_void_p]
    L.ts_batch_wire16_ok.argtypes = [C.c_void_p]
    L.ts_wire_widen_u16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.ts_batch_read_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ts_batch_read_pass_status.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.ts_filter_reads_multi.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64),
                                        C.c_size_t, C.POINTER(C.c_uint8)]
    L.ts_batch_shard_info.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(ShardInfo)]
    L.ts_batch_restrict_shard.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    L.ts_batch_set_shard_scale.argtypes = [C.c_void_p, C.c_uint32]
    L.ts_batch_pack_shard.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.ts_batch_bind_shard_message.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.ts_refresh_env.argtypes = [C.c_void_p]
    L.ts_streams_concurrent.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ts_batch_wait_scan.argtypes = [C.c_void_p, C.c_void_p]
    L.ts_batch_set_timing.argtypes = [C.c_void_p, C.c_uint32]
    L.ts_batch_set_record_bits.argtypes = [C.c_void_p, C.c_int]
    L.ts_shard_peek.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(ShardStatus)]
    L.ts_shards_finalize.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_uint32,
                                     C.POINTER(SegmentOut), C.POINTER(SegmentCounts)]
    L.ts_scan_segments_multi.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(SegmentIn), C.c_size_t,
                                         C.POINTER(SegmentOut), C.POINTER(SegmentCounts)]
    L.ts_box_probe.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ts_exchange_unique_id.argtypes = [C.c_void_p]
    L.ts_exchange_last_error.restype = C.c_char_p
    L.ts_exchange_create.restype = C.c_void_p
    L.ts_exchange_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.ts_exchange_destroy.argtypes = [C.c_void_p]
    L.ts_exchange_gather.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p]
    _lib = L
    return L


class TeloscanError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libteloscan error %d: %s" % (code, msg))
        self.code = code


def copy_array(ptr, n, dt):
    if n == 0:
        return np.zeros(0, dtype=dt)
    buf = C.string_at(C.cast(ptr, C.c_void_p), n * dt.itemsize)
    return np.frombuffer(buf, dtype=dt).copy()
