"""Option parsing of the reference's command line for the flags the scan path reads
(getopt loop of src/main.cpp:186-565, defaults of include/input.h:15-64) and the
UserInputTeloscope it yields.  bench.py and the parity tests describe their workloads as
Teloscope flag strings ("-c TTAGGG -w 1000 -s 500 -r -g ...") and turn them into the
library's parameters here, so that both mean exactly what the reference's CLI means by them.
"""
import shlex

import numpy as np

class Options:
    """UserInputTeloscope as main() leaves it (include/input.h:15-64)."""

    def __init__(self):
        self.input = None
        self.canonical_fwd = "CCCTAA"
        self.canonical_rev = "TTAGGG"
        self.canonical_size = 6
        self.raw_patterns = None
        self.window_size = 1000
        self.step = 1000
        self.terminal_limit = 50000
        self.edit_distance = 1
        self.max_match_dist = 50
        self.min_block_len = 300
        self.min_block_len_set = False
        self.max_block_dist = 500
        self.min_block_counts = 2
        self.min_block_density = np.float32(0.5)
        self.out_win_repeats = False
        self.out_gc = False
        self.out_entropy = False
        self.out_matches = False
        self.out_its = False
        self.ultra_fast = True
        self.manual_curation = False
        self.fastq_subset = False
        self.stdin_redirect = None

    def params(self):
        """fields shared by tso_params / ts_params"""
        return dict(window_size=self.window_size, step=self.step,
                    terminal_limit=self.terminal_limit, max_match_dist=self.max_match_dist,
                    min_block_len=self.min_block_len, max_block_dist=self.max_block_dist,
                    min_block_counts=self.min_block_counts,
                    min_block_density=float(self.min_block_density),
                    canonical_size=self.canonical_size, out_gc=int(self.out_gc),
                    out_entropy=int(self.out_entropy), out_matches=int(self.out_matches))


def _revcom(s):
    return s.translate(str.maketrans("ACGTacgt", "TGCAtgca"))[::-1]


def parse_cli(command):
    """getopt_long loop of src/main.cpp:186-565 for the options used by the manifests."""
    o = Options()
    toks = shlex.split(command.replace("<", " < "))
    i = 0
    needs_arg = {"-f", "-o", "-j", "-p", "-s", "-w", "-c", "-t", "-k", "-d", "-l", "-y", "-x"}
    while i < len(toks):
        t = toks[i]
        if t == "<":
            o.stdin_redirect = toks[i + 1]
            i += 2
            continue
        if t in needs_arg:
            a = toks[i + 1]
            i += 2
            if t == "-f":
                o.input = a
            elif t == "-c":
                c = a.upper()
                rc = _revcom(c)
                o.canonical_size = len(c)
                if c <= rc:                       # lex-smaller = Fwd (src/main.cpp:287-296)
                    o.canonical_fwd, o.canonical_rev = c, rc
                else:
                    o.canonical_fwd, o.canonical_rev = rc, c
            elif t == "-p":
                o.raw_patterns = [p.upper() for p in a.split(",") if p]
            elif t == "-w":
                o.window_size = int(a)
            elif t == "-s":
                o.step = int(a)
            elif t == "-t":
                o.terminal_limit = int(a)
            elif t == "-k":
                o.max_match_dist = int(a)
            elif t == "-d":
                o.max_block_dist = int(a)
            elif t == "-l":
                o.min_block_len = int(a)
                o.min_block_len_set = True
            elif t == "-y":
                o.min_block_density = np.float32(a)
            elif t == "-x":
                o.edit_distance = int(a)
            continue
        i += 1
        if t == "--fastq-subset":
            o.fastq_subset = True
        elif t in ("-r", "-g", "-e", "-m", "-i", "-a"):
            o.ultra_fast = False
            if t == "-r":
                o.out_win_repeats = True
            elif t == "-g":
                o.out_gc = True
            elif t == "-e":
                o.out_entropy = True
            elif t == "-m":
                o.out_matches = True
            elif t == "-i":
                o.out_its = True
        elif t == "-u":
            if o.out_win_repeats or o.out_gc or o.out_entropy or o.out_its or o.out_matches:
                o.ultra_fast = False
            else:
                o.ultra_fast = True
        elif t == "-n":
            o.manual_curation = True
        elif t.startswith("-"):
            pass                                   # --cmd, --verbose ...
        elif o.input is None:
            o.input = t                            # first positional
    if o.raw_patterns is None or not o.raw_patterns:
        o.raw_patterns = [o.canonical_fwd, o.canonical_rev]   # src/main.cpp:626-633
    return o




def user_input(opts, device=-1):
    """UserInputTeloscope of parsed options, with patternInfo expanded as main() does
    (src/main.cpp:636, expandPatternsWithOrientation)."""
    from .teloscope import UserInputTeloscope, expandPatternsWithOrientation
    ui = UserInputTeloscope(
        canonicalFwd=opts.canonical_fwd, canonicalRev=opts.canonical_rev,
        canonicalSize=opts.canonical_size, rawPatterns=list(opts.raw_patterns),
        windowSize=opts.window_size, step=opts.step, terminalLimit=opts.terminal_limit,
        editDistance=opts.edit_distance, maxMatchDist=opts.max_match_dist,
        minBlockLen=opts.min_block_len, minBlockLenSet=opts.min_block_len_set,
        maxBlockDist=opts.max_block_dist, minBlockCounts=opts.min_block_counts,
        minBlockDensity=float(opts.min_block_density), outGC=opts.out_gc,
        outEntropy=opts.out_entropy, outMatches=opts.out_matches, outITS=opts.out_its,
        outWinRepeats=opts.out_win_repeats, ultraFastMode=opts.ultra_fast, device=device)
    ui.patternInfo = expandPatternsWithOrientation(ui.rawPatterns, ui.editDistance, ui.canonicalFwd)
    return ui
