"""Host-side mirror of the reference's interface for the scan path, above the C-ABI.

Same names and argument meaning as the reference (include/teloscope.h, include/input.h,
include/read-filter.h, src/tools.cpp) so parity tests read like reference call sites:

    ui = UserInputTeloscope(windowSize=1000, step=500, outGC=True)
    ui.patternInfo = expandPatternsWithOrientation(ui.rawPatterns, ui.editDistance, ui.canonicalFwd)
    t = Teloscope(ui)
    seg = t.scanSegment(sequence, absPos, tipsOnly)      # -> SegmentData
    ReadTelomereFilter(ui).matches(read)                 # -> bool

Every call goes through libteloscan.so (HIP, gfx950).  Nothing here computes matches.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from . import _capi as K


def revCom(seq: str) -> str:
    return seq.translate(str.maketrans("ACGTacgt", "TGCAtgca"))[::-1]


def canonicalOrientation(canonical: str) -> Tuple[str, str]:
    """src/main.cpp:287-296: lexicographically smaller of (pattern, revcomp) is 'forward'."""
    f, r = C.create_string_buffer(64), C.create_string_buffer(64)
    rc = K.lib().ts_canonical_orientation(canonical.encode(), f, r)
    if rc != K.TS_OK:
        raise K.TeloscanError(rc, "bad canonical pattern")
    return f.value.decode(), r.value.decode()


def expandPatternsWithOrientation(rawPatterns, editDistance, canonicalFwd):
    """src/tools.cpp:201-283 -> [(pattern, isForward)] sorted and de-duplicated."""
    arr = C.POINTER(K.Pattern)()
    n = C.c_size_t(0)
    rc = K.lib().ts_expand_patterns(",".join(rawPatterns).encode(), int(editDistance),
                                    canonicalFwd.encode(), C.byref(arr), C.byref(n))
    if rc != K.TS_OK:
        raise K.TeloscanError(rc, "ts_expand_patterns failed")
    out = [(arr[i].seq.decode(), bool(arr[i].is_forward)) for i in range(n.value)]
    K.lib().ts_free_patterns(arr)
    return out


@dataclass
class UserInputTeloscope:
    """include/input.h:15-64 (fields the scan path reads)."""
    canonicalFwd: str = "CCCTAA"
    canonicalRev: str = "TTAGGG"
    canonicalSize: int = 6
    rawPatterns: List[str] = field(default_factory=lambda: ["TTAGGG", "CCCTAA"])
    patternInfo: List[Tuple[str, bool]] = field(default_factory=list)
    windowSize: int = 1000
    step: int = 1000
    terminalLimit: int = 50000
    editDistance: int = 1
    maxMatchDist: int = 50
    minBlockLen: int = 300
    minBlockLenSet: bool = False
    maxBlockDist: int = 500
    minBlockCounts: int = 2
    minBlockDensity: float = 0.5
    outGC: bool = False
    outEntropy: bool = False
    outMatches: bool = False
    outITS: bool = False
    outWinRepeats: bool = False
    ultraFastMode: bool = True
    foldCase: bool = True
    device: int = -1

    def _params(self):
        return K.Params(struct_size=C.sizeof(K.Params), window_size=self.windowSize, step=self.step,
                        terminal_limit=self.terminalLimit, max_match_dist=self.maxMatchDist,
                        min_block_len=self.minBlockLen, max_block_dist=self.maxBlockDist,
                        min_block_counts=self.minBlockCounts,
                        min_block_density=float(self.minBlockDensity),
                        canonical_size=self.canonicalSize, out_gc=int(self.outGC),
                        out_entropy=int(self.outEntropy), out_matches=int(self.outMatches),
                        out_its=int(self.outITS), fold_case=int(self.foldCase), device=self.device)

    def _patterns(self):
        if not self.patternInfo:
            self.patternInfo = expandPatternsWithOrientation(self.rawPatterns, self.editDistance,
                                                             self.canonicalFwd)
        arr = (K.Pattern * max(1, len(self.patternInfo)))()
        for i, (p, fwd) in enumerate(self.patternInfo):
            arr[i].seq = p.encode()
            arr[i].len = len(p)
            arr[i].is_forward = int(fwd)
            arr[i].is_canonical = int(p == self.canonicalFwd or p == self.canonicalRev)
        return arr, len(self.patternInfo)


class SegmentData:
    """include/teloscope.h:139-148 as numpy structured arrays."""

    def __init__(self, out: K.SegmentOut, tipsOnly: bool):
        self.windows = K.copy_array(out.windows, out.n_windows, K.WINDOW_DT)
        self.terminalBlocks = K.copy_array(out.terminal_blocks, out.n_terminal_blocks, K.BLOCK_DT)
        self.interstitialBlocks = K.copy_array(out.interstitial_blocks, out.n_interstitial_blocks,
                                               K.BLOCK_DT)
        self._m = K.copy_array(out.matches, out.n_matches, K.MATCH_DT)
        self._tips = tipsOnly
        self._split = {}

    # The five match vectors of SegmentData are views/selections of one record array; the selections
    # are made on first use (a 250 Mb contig has ~8 M matches: five eager copies cost more than the scan).
    def _sel(self, name):
        if name not in self._split:
            m = self._m
            f = m["flags"]
            if name == "fwd":
                v = m[(f & K.MATCH_FORWARD) != 0]
            elif name == "rev":
                v = m[(f & K.MATCH_FORWARD) == 0]
            elif self._tips:                   # src/teloscope.cpp:566-570 fills fwd/rev only
                v = m[:0]
            elif name == "can":
                v = m[(f & K.MATCH_CANONICAL) != 0]
            else:
                v = m[((f & K.MATCH_CANONICAL) == 0) & ((f & K.MATCH_TERMINAL) != 0)]
            self._split[name] = v
        return self._split[name]

    @property
    def allMatches(self):
        return self._m[:0] if self._tips else self._m

    @property
    def fwdMatches(self):
        return self._sel("fwd")

    @property
    def revMatches(self):
        return self._sel("rev")

    @property
    def canonicalMatches(self):
        return self._sel("can")

    @property
    def nonCanonicalMatches(self):
        return self._sel("noncan")


class _Ctx:
    def __init__(self, ptr):
        self.ptr = ptr

    def close(self):
        if self.ptr:
            K.lib().ts_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def error(self):
        return (K.lib().ts_last_error(self.ptr) or b"").decode()

    def refresh_env(self):
        """ts_refresh_env: the context reads its measurement / test knobs (TS_TIMING, TS_PACKED_UPLOAD, ...) from the
        environment when it is made; this reads them again."""
        rc = K.lib().ts_refresh_env(self.ptr)
        if rc != K.TS_OK:
            raise K.TeloscanError(rc, "ts_refresh_env failed")


class Teloscope:
    """include/teloscope.h:166-300: ctor builds the match structure from userInput.patternInfo."""

    def __init__(self, userInput: UserInputTeloscope):
        self.userInput = userInput
        params = userInput._params()
        pats, n = userInput._patterns()
        ptr = K.lib().ts_create(C.byref(params), pats, n)
        if not ptr:
            raise K.TeloscanError(K.TS_ERR_NO_DEVICE, (K.lib().ts_last_error(None) or b"").decode())
        self._ctx = _Ctx(ptr)

    def close(self):
        self._ctx.close()

    def usesFastPath(self):
        return bool(K.lib().ts_uses_fast_path(self._ctx.ptr))

    def scanSegments(self, segments, packed=False):
        """Batched scanSegment: segments = [(sequence, absPos, tipsOnly)] -> [SegmentData].
        packed=True hands the bases over as TS_INPUT_PACKED2 (2-bit codes + invalid runs, packed here with ts_pack_bases and
        this context's case folding): what a front end that packs while it parses would pass."""
        n = len(segments)
        seg_in = (K.SegmentIn * max(1, n))()
        keep = []
        for i, (seq, abs_pos, tips) in enumerate(segments):
            if isinstance(seq, str):
                seq = seq.encode()
            keep.append(seq)
            if packed:
                ps, alive = K.pack_sequence(seq, self.userInput.foldCase)
                keep.append((ps, alive))
                seg_in[i].seq = C.cast(C.pointer(ps), C.c_char_p)
                seg_in[i].input_format = K.TS_INPUT_PACKED2
            else:
                seg_in[i].seq = seq
            seg_in[i].len = len(seq)
            seg_in[i].abs_pos = abs_pos
            seg_in[i].tips_only = int(bool(tips))
        out = (K.SegmentOut * max(1, n))()
        rc = K.lib().ts_scan_segments(self._ctx.ptr, seg_in, n, out)
        if rc != K.TS_OK:
            raise K.TeloscanError(rc, self._ctx.error())
        res = [SegmentData(out[i], bool(segments[i][2])) for i in range(n)]
        K.lib().ts_free_segments(out, n)
        return res

    def scanSegmentsBlocksOnly(self, segments, tipsOnly=False, with_counts=False):
        """scanSegment for callers that do not read the match vectors (ts_scan_segments_blocks): scan +
        block calling ON THE DEVICE; returns SegmentData whose match vectors are empty (windows and
        blocks only).  segments = [(sequence, absPos)].  with_counts=True also returns, per segment,
        the sizes the match vectors would have had: (n_windows, n_matches, n_canonical, n_forward)."""
        n = len(segments)
        seqs = [s.encode() if isinstance(s, str) else bytes(s) for s, _ in segments]
        arr = (K.SegmentIn * max(1, n))()
        for i, (s, (_, a)) in enumerate(zip(seqs, segments)):
            arr[i].seq = s
            arr[i].len = len(s)
            arr[i].abs_pos = int(a)
            arr[i].tips_only = 1 if tipsOnly else 0
        out = (K.SegmentOut * max(1, n))()
        cnt = (K.SegmentCounts * max(1, n))()
        rc = K.lib().ts_scan_segments_blocks(self._ctx.ptr, arr, n, out, cnt)
        if rc != K.TS_OK:
            raise K.TeloscanError(rc, self._ctx.error())
        res = [SegmentData(out[i], bool(tipsOnly)) for i in range(n)]
        K.lib().ts_free_segments(out, n)
        if with_counts:
            return res, [(int(c.n_windows), int(c.n_matches), int(c.n_canonical), int(c.n_forward)) for c in cnt[:n]]
        return res

    def scanSegment(self, sequence, absPos=0, tipsOnly=False):
        """SegmentData Teloscope::scanSegment(std::string&, uint64_t absPos, bool tipsOnly)."""
        return self.scanSegments([(sequence, absPos, tipsOnly)])[0]

    @staticmethod
    def labelTerminalBlocks(blocks, gaps, pathSize, terminalLimit):
        """src/teloscope.cpp:259-383 -> (sorted blocks with isLongest, terminalLabel, scaffoldType)."""
        n = len(blocks)
        arr = (K.Block * max(1, n))()
        if n:
            C.memmove(arr, np.ascontiguousarray(blocks).ctypes.data, n * K.BLOCK_DT.itemsize)
        lab = C.create_string_buffer(2 * n + 2)
        st = C.c_int(0)
        rc = K.lib().ts_label_terminal_blocks(arr, n, gaps, pathSize, terminalLimit, lab, C.byref(st))
        if rc != K.TS_OK:
            raise K.TeloscanError(rc, "ts_label_terminal_blocks failed")
        out = np.frombuffer(C.string_at(arr, n * K.BLOCK_DT.itemsize), dtype=K.BLOCK_DT).copy() \
            if n else np.zeros(0, K.BLOCK_DT)
        return out, lab.value.decode(), st.value


class ReadTelomereFilter:
    """include/read-filter.h:10-18 / src/read-filter.cpp."""

    def __init__(self, userInput: UserInputTeloscope):
        params = userInput._params()
        pats, n = userInput._patterns()
        ptr = K.lib().ts_create_read_filter(C.byref(params), int(userInput.minBlockLenSet), pats, n)
        if not ptr:
            raise K.TeloscanError(K.TS_ERR_NO_DEVICE, (K.lib().ts_last_error(None) or b"").decode())
        self._ctx = _Ctx(ptr)

    def close(self):
        self._ctx.close()

    def matchesBatch(self, sequences):
        n = len(sequences)
        seqs = [s.encode() if isinstance(s, str) else bytes(s) for s in sequences]
        arr = (C.c_char_p * max(1, n))(*seqs) if n else (C.c_char_p * 1)()
        lens = (C.c_uint64 * max(1, n))(*[len(s) for s in seqs])
        out = (C.c_uint8 * max(1, n))()
        rc = K.lib().ts_filter_reads(self._ctx.ptr, arr, lens, n, out)
        if rc != K.TS_OK:
            raise K.TeloscanError(rc, self._ctx.error())
        return [bool(out[i]) for i in range(n)]

    def matches(self, sequence):
        """bool ReadTelomereFilter::matches(std::string sequence)."""
        return self.matchesBatch([sequence])[0]


def getGCContent(nucleotideCounts, windowSize):
    a = (C.c_uint32 * 4)(*[int(x) for x in nucleotideCounts])
    return np.float32(K.lib().ts_gc_content(a, int(windowSize)))


def getShannonEntropy(nucleotideCounts, windowSize):
    a = (C.c_uint32 * 4)(*[int(x) for x in nucleotideCounts])
    return np.float32(K.lib().ts_shannon_entropy(a, int(windowSize)))
