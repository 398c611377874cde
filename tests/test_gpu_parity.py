"""Parity of the HIP path (through the C-ABI) with the CPU oracle, on a real MI355X.

  * every legacy `.tst` manifest: the CLI stdout the reference expects, produced from the
    GPU scan, plus segment-level bit-exact equality with the oracle (windows, all five match
    lists, terminal and interstitial blocks);
  * FASTQ-subset manifests and read-filter known answers;
  * seeded random assemblies over the window/step/pattern grid incl. the BASELINE flag sets,
    with N runs, IUPAC codes, soft-masked bases and lengths straddling w, s, t, 2t and the tile;
  * size-independent properties at multi-megabase scale.
"""
import glob
import gzip
import os

import numpy as np
import pytest

from tests import harness as H
from tests import seqgen
from tests.backends import (OracleBackend, OracleReadFilter, ProductBackend, ProductReadFilter,
                            assert_segment_equal, segment_as_dict)
from tests.test_oracle_read_filter_kats import KATS, RANDOM_OPTION_SETS, random_read_set

pytestmark = pytest.mark.gpu

MANIFESTS = sorted(glob.glob(os.path.join(H.GOLDEN, "validateFiles", "*.tst")))
LEGACY = [m for m in MANIFESTS if H.load_manifest(m)["mode"] == "embedded"]
FASTQ = [m for m in MANIFESTS if os.path.basename(m).startswith("fastq_subset")]


class BothBackends:
    """Runs the product and checks every scanSegment result against the oracle on the fly."""

    def __init__(self, opts):
        self.prod = ProductBackend(opts)
        self.orac = OracleBackend(opts)
        assert [(p, f) for p, f, _ in self.prod.patterns] == [(p, f) for p, f, _, _ in self.orac.patterns]

    def scan_segment(self, seq, abs_pos, tips_only):
        g = self.prod.scan_segment(seq, abs_pos, tips_only)
        e = self.orac.scan_segment(seq, abs_pos, tips_only)
        assert_segment_equal(g, e, tips_only, ctx="abs_pos=%d len=%d tips=%s" % (abs_pos, len(seq), tips_only))
        return g

    def empty_blocks(self):
        return self.prod.empty_blocks()

    def label_terminal_blocks(self, *a):
        return self.prod.label_terminal_blocks(*a)


@pytest.mark.parametrize("path", LEGACY, ids=[os.path.basename(p) for p in LEGACY])
def test_legacy_manifest_on_gpu(path):
    m = H.load_manifest(path)
    opts = H.parse_cli(m["command"])
    stdout, _ = H.run_assembly(BothBackends(opts), opts, H.golden_path(opts.input))
    assert stdout.split("\n") == m["expected"].split("\n")


from tests import val_answers as _V


@pytest.mark.parametrize("args,must,sub", _V.cases(), ids=["%s | %s" % (a, s.replace("\t", " ")) for a, _, s in _V.cases()])
def test_val_known_answer_on_gpu(args, must, sub):
    """The known answers of the reference's CI script (val.sh:107-196, tests/golden/val_known_answers.tsv) through the HIP
    path, every scanSegment result compared with the oracle's on the way."""
    _V.check(BothBackends, args, must, sub)


@pytest.mark.parametrize("path", FASTQ, ids=[os.path.basename(p) for p in FASTQ])
def test_fastq_manifest_on_gpu(path):
    m = H.load_manifest(path)
    d = {}
    for k, v in m["directives"]:
        d.setdefault(k, []).append(v)
    opts = H.parse_cli(m["command"])
    src = H.golden_path(opts.input or opts.stdin_redirect)
    data = gzip.open(src, "rb").read() if src.endswith(".gz") else open(src, "rb").read()
    try:
        out, kept, total = H.run_fastq_subset(ProductReadFilter(opts), data)
        code = 0
    except ValueError:
        code = 1
    assert code == int(d["expect_exit"][0])
    if code == 0:
        so = d.get("expect_stdout", ["ignore"])[0]
        if " -o " in m["command"]:
            so = "testFiles/expected/fastq_subset.fq"
        if so != "ignore":
            assert out == open(H.golden_path(so), "rb").read()
        for sub in d.get("expect_stderr_substr", []):
            if sub.startswith("FASTQ subset: kept"):
                assert sub == "FASTQ subset: kept %d of %d reads." % (kept, total)


from tests.test_oracle_gfa_pins import GFA, gfa_case  # noqa: E402


@pytest.mark.parametrize("name,command,expect", GFA, ids=[g[0] for g in GFA])
def test_gfa_telomere_lengths_on_gpu(name, command, expect):
    """The reference's GFA manifests: per segment end the tips-only scan's terminal block length (tl_bp of
    testFiles/expected/gfa/*.tsv), from the HIP path."""
    opts = H.parse_cli(gfa_case(command))
    got = H.gfa_annotations(ProductBackend(opts), opts, H.golden_path(opts.input))
    assert got == H.read_gfa_expectation(H.golden_path(expect))


@pytest.mark.parametrize("case", range(len(KATS)))
def test_read_filter_kat_on_gpu(case):
    cli, seqs, expected = KATS[case]
    opts = H.parse_cli("--fastq-subset " + cli)
    got = [n for n, ok in zip(seqs, ProductReadFilter(opts).filter([s.encode() for s in seqs.values()])) if ok]
    assert got == expected


@pytest.mark.parametrize("cli", RANDOM_OPTION_SETS)
def test_read_filter_random_reads_on_gpu(cli):
    """the 240 seeded reads of scripts/test_bam_subset.py:384-404, product vs oracle"""
    seqs = [s.encode() for s in random_read_set().values()]
    opts = H.parse_cli("--fastq-subset " + cli)
    assert ProductReadFilter(opts).filter(seqs) == OracleReadFilter(opts).filter(seqs)


GRID = [
    # cli, note
    "-r -g -e -m -i",                                           # default w = s = 1000 (straddle loss)
    "-w 1000 -s 500 -r -g -e -m -i",
    "-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 500 -r -g -e -m -i",   # BASELINE cfg 2/3
    "-c CCCTAAA -w 2000 -s 1000 -r -g -e -m -i",                # BASELINE cfg 5 (k = 7)
    "-w 500 -s 250 -r -g -e -i",
    "-w 200 -s 200 -r -g",
    "-w 1000 -s 100 -g -e -i",                                  # q = 10
    "-w 1000 -s 300 -r -g -e -i",                               # w not a multiple of s
    "-w 777 -s 333 -r -g -e -m -i -t 700",
    "-w 64 -s 16 -g -i -t 100",
    "-x 0 -w 1000 -s 500 -g -e -i",
    "-x 2 -w 1000 -s 500 -r -i -k 20 -d 100",
    "-p TTAGGN -x 0 -w 300 -s 150 -g -i",
    "-c TTAGG -w 100 -s 50 -g -e -i -l 50",                     # k = 5
    "-c TTAG -x 1 -w 200 -s 100 -r -g -e -i -l 40",             # k = 4: every sixth position matches
    "-c TTA -x 0 -w 100 -s 50 -g -i -l 30 -k 10",               # k = 3, the shortest the tiled kernel takes
    "-c TTTTAGGG -x 1 -w 1000 -s 1000 -r -g -e -i",             # k = 8
    "-c TTTTTAGGG -x 1 -w 400 -s 200 -r -g",                    # k = 9
    "",                                                         # ultra-fast (tips only), t = 50000
    "-t 300",
    "-t 1000 -l 60 -k 10 -d 30 -y 0.8",
    "-c CCCTAAA -t 2500",
]
LENGTHS = [1, 5, 6, 7, 15, 16, 17, 63, 100, 250, 499, 500, 501, 999, 1000, 1001, 1999, 2000, 2001,
           4095, 32499, 32500, 32501, 33007, 65000, 70001, 99999, 100000, 100001, 131072, 250003]


@pytest.mark.parametrize("cli", GRID)
def test_random_segments_match_oracle(cli):
    opts = H.parse_cli("x.fa " + cli)
    prod, orac = ProductBackend(opts), OracleBackend(opts)
    if orac.ambiguous:
        # a pattern that is its own reverse complement comes out of the expansion from both strands (TTAA and CTAG from
        # TTAG with one mismatch): which copy std::sort leaves first is unspecified in the reference; the
        # oracle takes the product's orientation flag for THOSE entries only, every other entry has to agree
        orac = orac.with_ambiguous_orientation_from(prod.patterns)
    rng = np.random.default_rng(abs(hash(cli)) % (2 ** 32) if False else len(cli) * 7919 + 13)
    unit_f, unit_r = opts.canonical_fwd, opts.canonical_rev
    segs = []
    for i, n in enumerate(LENGTHS):
        s = seqgen.chromosome(rng, n, unit_f, unit_r, telo_repeats=min(150, max(1, n // 40)),
                              tvr_rate=0.03, n_its=3, iupac=(n // 5000) * (i % 2),
                              lower=0.0, n_runs=0)
        segs.append((s, int(rng.integers(0, 10 ** 7)), opts.ultra_fast))
    segs.append((b"", 5, opts.ultra_fast))
    # one call, many segments: result order = input order
    got = prod.scan_segments(segs)
    for (s, ap, tips), g in zip(segs, got):
        e = orac.scan_segment(s, ap, tips)
        assert_segment_equal(g, e, tips, ctx="cli=%r len=%d" % (cli, len(s)))


PACKED_ROUTE_GRID = ["-w 1000 -s 500 -r -g -e -m -i", "-c CCCTAAA -w 2000 -s 1000 -r -g -e -m -i", "-w 777 -s 333 -r -g -e -m -i -t 700",
                     "-t 300", "-p TTAGGG,TTAGG -w 1000 -s 500 -r -g -e -m -i", "-c TTAGG -w 100 -s 50 -g -e -i -l 50"]


@pytest.mark.parametrize("fold", [True, False])
@pytest.mark.parametrize("cli", PACKED_ROUTE_GRID)
def test_packed_upload_route_equals_plain_route_and_oracle(cli, fold, monkeypatch):
    """The host entry points upload bases as 2-bit codes + invalid runs (pack.cpp / unpack.hip) when a call is large; here
    the threshold is taken away so that the segments of the parity grid go that way: IUPAC codes, soft-masked stretches
    (valid when the context folds case, invalid otherwise), N runs, runs that cross the 16 k blocks and 4 k ranges the
    packing threads work in, every byte value once.  Route on == route off == oracle, record for record."""
    import teloscope_amd as ta
    from teloscope_amd.cli import user_input
    opts = H.parse_cli("x.fa " + cli)
    ui = user_input(opts)
    ui.foldCase = fold
    tel = ta.Teloscope(ui)
    orac = OracleBackend(opts)
    rng = np.random.default_rng(len(cli) * 3 + int(fold))
    segs = []
    for i, n in enumerate([1, 3, 4, 5, 63, 64, 65, 4095, 4096, 4097, 16383, 16385, 70001, 250003, 1_100_000]):
        s = bytearray(seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev, telo_repeats=min(150, max(1, n // 40)), tvr_rate=0.03,
                                        n_its=3, iupac=(n // 3000), lower=0.1 if i % 3 == 0 else 0.0, n_runs=(n // 20000)))
        if n > 70000:
            s[4090:4100] = b"N" * 10                       # across a range boundary
            s[16380:16390] = b"nnnnnRYKMS"                 # across a block boundary
            s[30000:30256] = bytes(range(256))             # every byte value
        segs.append((bytes(s), int(rng.integers(0, 10 ** 7)), opts.ultra_fast))
    exp = None
    # ... and a third way in: the caller hands the bases over ALREADY packed (TS_INPUT_PACKED2: codes + invalid runs, here
    # made by ts_pack_bases with the context's case folding) — the staging threads then copy codes, whole bytes where the
    # phases agree (full scans) and shifted where they do not (the second region of a tips-only scan starts anywhere)
    from teloscope_amd import _capi as K
    assert K.lib().ts_takes_text_input(tel._ctx.ptr, int(opts.ultra_fast)) == 1       # (tiled and general parameter sets alike)
    for route in ("1", "0", "packed-in"):
        monkeypatch.setenv("TS_PACKED_UPLOAD", "1" if route == "packed-in" else route)
        monkeypatch.setenv("TS_PACKED_MIN_BYTES", "0")
        tel._ctx.refresh_env()                              # (the context read its knobs when it was made)
        got = [segment_as_dict(s) for s in tel.scanSegments(segs, packed=(route == "packed-in"))]
        if exp is None:
            # the oracle is strict scanSegment (lower case = non-ACGT): a context that folds case sees what the reference's
            # callers hand over after unmaskSequence
            exp = [orac.scan_segment(s.upper() if fold else s, ap, tips) for s, ap, tips in segs]
        for (s, ap, tips), g, e in zip(segs, got, exp):
            assert_segment_equal(g, e, tips, ctx="cli=%r fold=%r route=%s len=%d" % (cli, fold, route, len(s)))


def test_packed_upload_falls_back_for_data_that_is_not_sequence(monkeypatch):
    """A chunk with more invalid runs than the packed route's run list holds (2^18) goes as ASCII: 1.2 M isolated 'N's in a
    2.4 Mb segment, next to an ordinary one; == oracle either way."""
    import teloscope_amd as ta
    from teloscope_amd.cli import user_input
    opts = H.parse_cli("x.fa -w 1000 -s 500 -r -g -e -m -i")
    tel = ta.Teloscope(user_input(opts))
    orac = OracleBackend(opts)
    rng = np.random.default_rng(12)
    junk = bytearray(seqgen.chromosome(rng, 2_400_000, opts.canonical_fwd, opts.canonical_rev, n_its=4))
    junk[0::2] = b"N" * len(junk[0::2])
    good = seqgen.chromosome(rng, 300_000, opts.canonical_fwd, opts.canonical_rev, n_its=4, iupac=9)
    segs = [(good, 7, False), (bytes(junk), 11, False), (good[::-1], 13, False)]
    monkeypatch.setenv("TS_PACKED_MIN_BYTES", "0")
    tel._ctx.refresh_env()
    got = [segment_as_dict(s) for s in tel.scanSegments(segs)]
    for (s, ap, tips), g in zip(segs, got):
        assert_segment_equal(g, orac.scan_segment(s, ap, tips), tips, ctx="len=%d" % len(s))


def test_general_path_dense_mixed_lengths_overflow_their_tile_slots():
    """The fused general kernel gives a tile a slot of one record per position; a mixed-length set on a homopolymer puts
    two records on every position: the group must run again with larger slots (and say nothing wrong in between)."""
    opts = H.parse_cli("x.fa -p AAAAAA,AAAAA -x 0 -w 1000 -s 500 -r -g -e -m -i")
    prod, orac = ProductBackend(opts), OracleBackend(opts)
    assert not prod.teloscope.usesFastPath()
    if orac.ambiguous:
        orac = orac.with_ambiguous_orientation_from(prod.patterns)
    rng = np.random.default_rng(4)
    segs = []
    for n, runs in [(20000, [(100, 19000)]), (9000, [(0, 9000)]), (70000, [(5000, 30000), (40000, 4097)])]:
        s = bytearray(seqgen.random_dna(rng, n).tobytes())
        for a, ln in runs:
            s[a:a + ln] = b"A" * ln
        segs.append((bytes(s), int(rng.integers(0, 10 ** 6)), False))
    segs.append((seqgen.chromosome(rng, 50000, opts.canonical_fwd, opts.canonical_rev, n_its=2), 5, False))
    got = prod.scan_segments(segs)
    for (s, ap, tips), g in zip(segs, got):
        assert_segment_equal(g, orac.scan_segment(s, ap, tips), tips, ctx="dense mixed len=%d" % len(s))


@pytest.mark.parametrize("cli", ["-w 1000 -s 500 -r -g -e -m -i", "-p TTAGGG,TTAGG -w 1000 -s 500 -r -g -e -m -i",
                                 "-c TTTAGGGTTTAGGG -x 1 -w 700 -s 700 -r -g -e -m -i"])
def test_text_pieces_are_bounded_by_their_count(cli):
    """TS_INPUT_TEXT_PIECES: the piece array carries its length (ts_segment_in.n_pieces); pieces that hold fewer bases
    than the segment declares — or a count that is not set — are TS_ERR_INVALID_ARG, never a read past the array; the
    same text through pieces and as joined bases gives the same segment.  (Tiled and general parameter sets.)"""
    import ctypes as C
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.cli import user_input
    opts = H.parse_cli("x.fa " + cli)
    tel = ta.Teloscope(user_input(opts))
    L = K.lib()
    rng = np.random.default_rng(3)
    seq = seqgen.chromosome(rng, 30_011, opts.canonical_fwd, opts.canonical_rev, n_its=2, iupac=3)
    text = b"\n".join(seq[i:i + 70] for i in range(0, len(seq), 70)) + b"\n"
    cuts = [0, 7100, 7171 * 2, len(text)]                                  # piece ends on line ends and inside lines
    blobs = [text[a:b] for a, b in zip(cuts, cuts[1:])]
    pieces = (K.TextPiece * len(blobs))()
    for i, bl in enumerate(blobs):
        pieces[i].text = bl
        pieces[i].text_len = len(bl)
        pieces[i].n_bases = len(bl.replace(b"\n", b""))
    assert sum(p.n_bases for p in pieces) == len(seq)

    def scan(n_pieces, length):
        seg = (K.SegmentIn * 1)()
        seg[0].seq = C.cast(pieces, C.c_char_p)
        seg[0].len = length
        seg[0].input_format = K.TS_INPUT_TEXT_PIECES
        seg[0].n_pieces = n_pieces
        out = (K.SegmentOut * 1)()
        rc = L.ts_scan_segments(tel._ctx.ptr, seg, 1, out)
        return rc, out

    rc, out = scan(len(blobs), len(seq))
    assert rc == 0, tel._ctx.error()
    got = segment_as_dict(ta.SegmentData(out[0], False))
    assert_segment_equal(got, OracleBackend(opts).scan_segment(seq, 0, False), False, ctx="text pieces")
    L.ts_free_segments(out, 1)
    for n_pieces, length in ((len(blobs) - 1, len(seq)), (0, len(seq)), (len(blobs), len(seq) + 5)):
        rc, out = scan(n_pieces, length)
        assert rc == K.TS_ERR_INVALID_ARG, (n_pieces, length, rc)


DENSE_GRID = [
    # every position of a homopolymer run is a match: the per-chunk match list, the staging flushes and
    # the packed 16-bit window counters of the tiled kernel at their limits
    ("-c AAAAAA -x 0 -w 1000 -s 500 -r -g -e -m -i", b"A"),
    ("-c AAAAAA -x 0 -r -g -e -m -i", b"A"),                       # w = s = 1000
    ("-c AAAAAA -x 1 -w 20000 -s 10000 -g -e -m -i", b"A"),        # windows of 20 kb: ~20 k matches each
    ("-c AAAAAA -x 0 -w 40000 -s 20000 -g -i", b"A"),              # > 32768: general kernels
    ("-c ACACAC -x 1 -w 3000 -s 1000 -g -e -m -i", b"AC"),
    ("-c AAAAAA -x 0 -t 6000", b"A"),                              # tips only
]


@pytest.mark.parametrize("cli,unit", DENSE_GRID)
def test_dense_repeats_and_large_windows(cli, unit):
    opts = H.parse_cli("x.fa " + cli)
    prod, orac = ProductBackend(opts), OracleBackend(opts)
    rng = np.random.default_rng(len(cli))
    segs = []
    for n, runs in [(9000, [(100, 8200)]), (70000, [(0, 30000), (41000, 29000)]), (150000, [(5, 149990)]),
                    (33000, [(1000, 2100), (16000, 2017), (30000, 3000)])]:
        s = bytearray(seqgen.random_dna(rng, n).tobytes())
        for a, ln in runs:
            s[a:a + ln] = (unit * (ln // len(unit) + 1))[:ln]
        s[n // 2] = ord("N")                                       # one invalid base inside a dense run or not
        segs.append((bytes(s), int(rng.integers(0, 10 ** 6)), opts.ultra_fast))
    # match density around the match queue's capacity (512 of a chunk's 2016 positions): runs of random length
    # alternate with random sequence, so some chunks overflow the queue (appended in lane groups) and their
    # neighbours do not; plus single dense lane groups inside sparse chunks
    for n, frac in [(60000, 0.2), (60000, 0.3), (90000, 0.5), (40000, 0.05)]:
        s = bytearray(seqgen.random_dna(rng, n).tobytes())
        at = 0
        while at < n:
            ln = int(rng.integers(20, 700))
            if rng.random() < frac:
                s[at:at + ln] = (unit * (ln // len(unit) + 1))[:ln][:n - at]
            at += ln
        segs.append((bytes(s), int(rng.integers(0, 10 ** 6)), opts.ultra_fast))
    got = prod.scan_segments(segs)
    for (s, ap, tips), g in zip(segs, got):
        assert_segment_equal(g, orac.scan_segment(s, ap, tips), tips, ctx="cli=%r len=%d" % (cli, len(s)))


def test_non_acgt_and_case_handling():
    """IUPAC codes / N inside a segment kill every k-mer that touches them; lower case follows
    fold_case (default: folded, like unmaskSequence + scanSegment; 0: strict scanSegment)."""
    import teloscope_amd as ta
    opts = H.parse_cli("x.fa -w 1000 -s 500 -r -g -e -m -i")
    rng = np.random.default_rng(5)
    s = bytearray(seqgen.chromosome(rng, 40000, n_its=4, iupac=40, lower=0.05, n_runs=6))
    prod, orac = ProductBackend(opts), OracleBackend(opts)
    g = prod.scan_segment(bytes(s), 123, False)
    assert_segment_equal(g, orac.scan_segment(bytes(s).upper(), 123, False), False, ctx="folded")
    # strict: lower-case letters are non-ACGT, exactly what scanSegment does with them
    ui = prod.ui
    ui.foldCase = False
    strict = ta.Teloscope(ui)
    from tests.backends import segment_as_dict
    g2 = segment_as_dict(strict.scanSegment(bytes(s), 123, False))
    assert_segment_equal(g2, orac.scan_segment(bytes(s), 123, False), False, ctx="strict")


GENERIC_GRID = [
    # parameter sets outside the tiled kernel's closed form -> general kernels (generic.hip)
    "-w 10 -s 5 -g -i",                                   # L > step and L > overlap: both start indices wrap
    "-w 1000 -s 997 -r -g -e -i",                         # L > overlap only
    "-w 12 -s 8 -g -e -m -i -t 50",                       # L > overlap, tiny windows
    "-w 1000 -s 998 -r -g",
    "-p TTAGGG,TTAGG -w 1000 -s 500 -r -g -e -m -i",      # mixed lengths 5/6 (several matches per position)
    "-p TTAGGG,TTAGG,TTTAGGGTTTAGGG -x 1 -w 500 -s 500 -g -i",
    "-p TTAGGG,TTAGG -t 400",                             # mixed lengths, tips only
    "-c TTTAGGGTTTAGGG -x 1 -w 1000 -s 500 -r -g -i",     # k = 14 (no LDS table)
    "-c TTAGGGTTAGGG -x 0",                               # k = 12, tips only
    "-c TTAGGGTTAGGGTTAGGGTTAGGGTTAGGGTT -x 0 -w 1000 -s 500 -r -g -e -i",     # k = 32, the longest pattern taken
    "-c TTAGGGTTAGGGTTAGGGTTAGGGTTAGGGT -x 1 -t 2000",    # k = 31 with one mismatch (94 patterns), tips only
    "-p TTAGGG,TTAGGGTTAGGGTTAGGGTTAGGGTTAGGGTT -x 0 -w 2000 -s 1000 -r -g -i",   # lengths 6 and 32 in one set
    "-c TTAGGGTTAGGGTTAGGGTT -x 2 -w 1000 -s 500 -g -i",  # k = 20 with two mismatches: 3 500 patterns in LDS (list form)
    "-c TTAGGGTTAGGGTTAGGGTTAGGG -x 2 -w 1000 -s 500 -g -i",   # k = 24: 5 100 patterns, more than the lists' LDS holds (strided form, lists in device memory)
]


@pytest.mark.parametrize("cli", GENERIC_GRID)
def test_general_kernels_match_oracle(cli):
    opts = H.parse_cli("x.fa " + cli)
    prod, orac = ProductBackend(opts), OracleBackend(opts)
    assert not prod.teloscope.usesFastPath() or opts.ultra_fast
    rng = np.random.default_rng(len(cli) * 31 + 7)
    segs = []
    for i, n in enumerate([1, 4, 5, 9, 10, 11, 23, 100, 997, 998, 1000, 1994, 2001, 5000, 33333, 70001]):
        s = seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev,
                              telo_repeats=min(100, max(1, n // 40)), tvr_rate=0.05, n_its=3,
                              iupac=(n // 3000) * (i % 2))
        segs.append((s, int(rng.integers(0, 10 ** 6)), opts.ultra_fast))
    got = prod.scan_segments(segs)
    for (s, ap, tips), g in zip(segs, got):
        assert_segment_equal(g, orac.scan_segment(s, ap, tips), tips, ctx="cli=%r len=%d" % (cli, len(s)))


@pytest.mark.parametrize("cli", ["-p TTAGGG,TTAGG -w 1000 -s 500 -r -g -e -m -i", "-c TTTAGGGTTTAGGG -x 1 -w 1000 -s 500 -r -g -i",
                                 "-w 1000 -s 997 -r -g -e -i", "-p TTAGGG,TTAGG,TTTAGGGTTTAGGG -x 1 -w 500 -s 500 -g -i"])
def test_general_kernels_strided_form_matches_oracle(cli, monkeypatch):
    """The fused general kernel has two forms (generic.hip): per-candidate work on compacted lists — the default where a
    tile adds to few enough window records — and the position-strided form, which other parameter sets and dense groups
    fall back to.  Here the strided form is pinned (TS_GEN_LIST=0) on parameter sets that normally take the list form."""
    monkeypatch.setenv("TS_GEN_LIST", "0")
    test_general_kernels_match_oracle(cli)


WIDE_GRID = [
    # pattern sets beyond the table forms of the general kernels (more than 8 distinct lengths, or a pattern above 32 bases):
    # the wide form (generic.hip: ts_general_wide — 128-bit codes, a u64 of matched lengths per position)
    "-p " + "TTAGGG" * 6 + " -x 0 -w 1000 -s 500 -g -r -i",                          # one 36-base pattern (+ its reverse complement)
    "-x 0 -p " + ",".join("TTAGGG"[:3] + "A" * i for i in range(9)) + " -w 1000 -s 500 -g -r -e -i",   # nine lengths 3..11
    "-x 1 -p " + ",".join("TTAGGG"[:3] + "A" * i for i in range(9)) + " -t 600",     # the same with one mismatch, tips only
    "-c TTAGGG -p TTAGGG,TTAGG,TTTAGGG,TTTTAGGG,TTAGGGG,TTAGGGTTAGGG,TTAGGGTTAGGGTTAGGG,TTAG,TTAGGGT," + "TTAGGG" * 7 + " -x 0 -w 500 -s 500 -g -i",
    "-c TTAGGG -p TTAGGG," + ("TTAGGG" * 11)[:63] + " -x 0 -w 2000 -s 1000 -r -g -m -i",   # lengths 6 and 63: the longest a ts_pattern holds
    "-p " + ("TTAGGG" * 7)[:40] + " -x 1 -w 1000 -s 990 -g -i",                      # 40 bases, one mismatch, L > overlap (wrapped start index)
    "-x 0 -p A,AA,AAA,AAAA,AAAAA,AAAAAA,AAAAAAA,AAAAAAAA,AAAAAAAAA,AAAAAAAAAA -w 100 -s 50 -g -i -l 20",   # every length matches inside a run
    "-x 0 -p " + ",".join(("TTAGGG" * 7)[:k] for k in range(4, 21)) + " -w 1000 -s 500 -g -r -i",     # 17 lengths: the 32-bit masks
    "-x 0 -p " + ",".join(("TTAGGG" * 7)[:k] for k in range(3, 36)) + " -w 600 -s 600 -g -i",          # 33 lengths: the 64-bit masks
    "-p " + ("TTAGGG" * 7)[:40] + " -x 2 -t 700",                                    # 14 k patterns of 40 bases: the lists stay in device memory
]


@pytest.mark.parametrize("cli", WIDE_GRID)
def test_wide_pattern_sets_match_oracle(cli):
    opts = H.parse_cli("x.fa " + cli)
    prod, orac = ProductBackend(opts), OracleBackend(opts)
    if orac.ambiguous:
        orac = orac.with_ambiguous_orientation_from(prod.patterns)
    assert not prod.teloscope.usesFastPath() or opts.ultra_fast
    rng = np.random.default_rng(len(cli) * 13 + 5)
    segs = []
    for i, n in enumerate([1, 5, 9, 36, 37, 63, 64, 65, 100, 997, 1000, 2001, 4095, 4096, 4097, 4160, 8200, 33333, 70001]):
        s = bytearray(seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev,
                                        telo_repeats=min(100, max(1, n // 40)), tvr_rate=0.03, n_its=3, iupac=(n // 3000) * (i % 2)))
        if n > 8000:
            s[4000:4300] = (b"TTAGGG" * 50)[:300]            # a repeat across a tile boundary
            s[6000:6200] = b"A" * 200                         # a homopolymer run
            s[4090] = ord("N")
        segs.append((bytes(s), int(rng.integers(0, 10 ** 6)), opts.ultra_fast))
    got = prod.scan_segments(segs)
    for (s, ap, tips), g in zip(segs, got):
        assert_segment_equal(g, orac.scan_segment(s, ap, tips), tips, ctx="cli=%r len=%d" % (cli, len(s)))


def test_what_a_pattern_cannot_hold_is_refused_loudly():
    """A pattern longer than the 63 bases a ts_pattern holds is refused when the context is made (never truncated, never routed
    to a CPU path); everything a ts_pattern[] can express is scanned (test_wide_pattern_sets_match_oracle)."""
    import ctypes as C
    from teloscope_amd import _capi as K
    L = K.lib()
    arr = (K.Pattern * 1)()
    arr[0].seq = b"TTAGGG" * 10
    arr[0].len = 64
    prm = K.Params()
    prm.struct_size = C.sizeof(K.Params)
    prm.window_size, prm.step = 1000, 500
    assert not L.ts_create(C.byref(prm), arr, 1)
    assert b"longer than the 63 bases" in L.ts_last_error(None)
    out, n = C.POINTER(K.Pattern)(), C.c_size_t(0)
    assert L.ts_expand_patterns(("TTAGGG" * 11).encode(), 0, b"CCCTAA", C.byref(out), C.byref(n)) == K.TS_ERR_UNSUPPORTED


def test_real_chromosome_matches_oracle():
    """testFiles/bTaeGut7_chr33_mat.fa.gz (4.2 Mb zebra finch chr33), BASELINE flag set."""
    opts = H.parse_cli("x -c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 500 -r -g -e -m -i")
    recs = H.read_fasta(H.golden_path("testFiles/bTaeGut7_chr33_mat.fa.gz"))
    both = BothBackends(opts)
    for i, (h, s) in enumerate(recs):
        H.walk_path(both, opts, i, h, s)


def test_large_scan_properties():
    """64 Mb in 5 segments at the BASELINE geometry: window count, exact nucleotide totals
    (checksum of checksums against numpy), sorted match stream, covered = k x count, and
    full equality with the oracle on a 3 Mb slice."""
    opts = H.parse_cli("x -c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 1000 -r -g -e -m -i")
    prod, orac = ProductBackend(opts), OracleBackend(opts)
    rng = np.random.default_rng(99)
    lens = [20_000_003, 3_000_000, 25_000_000, 999, 16_000_001]
    segs = [(seqgen.chromosome(rng, n, telo_repeats=500, n_its=20), 1000 * i, False) for i, n in enumerate(lens)]
    got = prod.scan_segments(segs)
    for (s, ap, _), g, n in zip(segs, got, lens):
        w = g["windows"]
        assert len(w) == -(-n // 1000)
        arr = np.frombuffer(s, dtype=np.uint8)
        tot = [int((arr == c).sum()) for c in b"ACGT"]
        assert w["nucleotide_counts"].sum(axis=0).tolist() == tot          # w == s: windows tile the segment
        assert int(w["current_window_size"].sum()) == n
        pos = g["all_matches"]["position"]
        assert np.all(np.diff(pos.astype(np.int64)) > 0)
        assert pos.min() >= ap and pos.max() + 6 <= ap + n
        f = (g["all_matches"]["flags"] & 1) != 0
        assert int(w["fwd_covered"].sum()) == 6 * int(f.sum())
        assert int(w["rev_covered"].sum()) == 6 * int((~f).sum())
    assert_segment_equal(got[1], orac.scan_segment(segs[1][0], segs[1][1], False), False, ctx="3Mb slice")


def _hip():
    import ctypes as C
    for name in ("libamdhip64.so", "libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so"):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    pytest.skip("libamdhip64 not loadable")


def test_batch_api_summary_and_overflow_rescan():
    """Device-resident batch API: a deliberately tiny match capacity forces the overflow ->
    grow -> rescan path; the per-segment summary kernel (the buffer ranks gather) must agree
    with the downloaded results and with the oracle."""
    import ctypes as C
    from teloscope_amd import _capi as K
    opts = H.parse_cli("x -w 1000 -s 500 -r -g -e -m -i")
    prod, orac = ProductBackend(opts), OracleBackend(opts)
    L = K.lib()
    rng = np.random.default_rng(4)
    seqs = [seqgen.chromosome(rng, n, telo_repeats=400, n_its=6) for n in (300_000, 50_000, 1, 0, 1_000_000)]
    n = len(seqs)
    lens = (C.c_uint64 * n)(*[len(s) for s in seqs])
    absp = (C.c_uint64 * n)(*[1000 * i for i in range(n)])
    ctx = prod.teloscope._ctx.ptr
    b = L.ts_batch_create(ctx, lens, absp, n, 0, 64)            # 64 records: certain to overflow
    assert b, L.ts_last_error(ctx)
    for i, s in enumerate(seqs):
        assert L.ts_batch_upload(b, i, s) == 0
    assert L.ts_batch_scan(b, None, None) == 0
    assert L.ts_batch_sync(b) == 0, L.ts_last_error(ctx)
    info = K.BatchInfo()
    L.ts_batch_get_info(b, C.byref(info))
    out = (K.SegmentOut * n)()
    assert L.ts_batch_download(b, None, out) == 0, L.ts_last_error(ctx)
    hip = _hip()
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), C.c_size_t(32 * n)) == 0
    assert L.ts_batch_segment_summary(b, dptr, None) == 0
    assert hip.hipDeviceSynchronize() == 0
    summ = np.zeros((n, 4), dtype=np.uint64)
    assert hip.hipMemcpy(summ.ctypes.data_as(C.c_void_p), dptr, C.c_size_t(32 * n), 2) == 0
    hip.hipFree(dptr)
    total = 0
    for i, s in enumerate(seqs):
        e = orac.scan_segment(s, 1000 * i, False)
        m = e["all_matches"]
        assert summ[i].tolist() == [len(e["windows"]), len(m), int(m["is_canonical"].sum()),
                                    int(m["is_forward"].sum())]
        assert out[i].n_matches == len(m) and out[i].n_windows == len(e["windows"])
        total += len(m)
    assert info.n_matches == total
    assert info.algorithmic_bytes == sum(len(s) for s in seqs) + 32 * info.n_windows + 4 * total
    L.ts_free_segments(out, n)
    L.ts_batch_destroy(b)


@pytest.mark.parametrize("cli", ["-l 42", "-x 0 -l 18 -y 0.8 -k 10 -d 10", "-c CCCTAAA -l 60 -k 20 -d 200 -y 0.6"])
def test_read_filter_device_predicate_long_reads(cli):
    """HiFi-like reads (1-40 kb, several tiles each) with terminal / internal telomeric tracts,
    1 % substitutions, N's and soft-masking: pass bits from the on-device terminal-block predicate
    must equal ReadTelomereFilter::matches as restated by the oracle, in input order."""
    opts = H.parse_cli("--fastq-subset " + cli)
    rng = np.random.default_rng(43)
    unit = opts.canonical_rev
    reads = []
    for i in range(1500):
        n = int(np.clip(rng.normal(15000, 8000), 50, 40000))
        s = bytearray(seqgen.random_dna(rng, n).tobytes())
        kind = i % 5
        if kind in (0, 1):                                     # tract at an end / in the middle
            ln = int(rng.integers(1, 120)) * len(unit)
            u = unit if rng.random() < 0.5 else opts.canonical_fwd
            t = seqgen.mutate(rng, seqgen.repeat_array(u, ln // len(u)), 0.01).tobytes()
            at = 0 if kind == 0 else int(rng.integers(0, max(1, n - len(t))))
            if rng.random() < 0.5 and kind == 0:
                at = max(0, n - len(t))
            s[at:at + len(t)] = t[:max(0, n - at)]
        if i % 7 == 0:
            for _ in range(3):
                at = int(rng.integers(0, n))
                s[at:at + int(rng.integers(1, 5))] = b"N" * 4
        if i % 9 == 0:
            s = bytearray(bytes(s).lower())
        if i % 11 == 0:
            s += b"\r"
        reads.append(bytes(s[:40001]))
    got = ProductReadFilter(opts).filter(reads)
    exp = OracleReadFilter(opts).filter(reads)
    assert got == exp
    assert 0 < sum(got) < len(got)


def test_bench_full_size_properties_small():
    """bench.py --verify at 0.2 Gb (the same untimed check the round-end profile runs at 3 Gb):
    per-contig match / canonical / forward counts against an independent torch k-mer lookup and
    A/C/G/T totals against even-window sums."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gbases", "0.2", "--contigs", "12",
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--verify"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["verify"]["contigs_checked"] == 12 and out["verify"]["matches_checked"] > 1_000_000


def test_bench_line_contract_and_host_entry_points_small():
    """The bench line's contract fields, and the PCIe-inclusive (default at N = 1) / --blocks legs (the host-buffer entry points and the
    device block calling on the bench workload): both entry points must see the matches the resident scan counted."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gbases", "0.1", "--contigs", "9",
                        "--steps", "2", "--warmup", "1", "--cpu-sample-mb", "8", "--blocks"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["scaling"] == "strong" and out["vs_baseline"] is None
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1 and out["roofline"]["launches_timed"] == 2
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0
    n = out["config"]["matches"]
    assert out["pcie_inclusive"]["blocks_windows_counts"]["matches"] == n
    assert out["pcie_inclusive"]["with_match_vectors"]["matches"] == n
    assert out["device_block_calling"]["terminal_blocks"] >= 9


BLOCKCALL_GRID = ["-r -g -e -m -i", "-w 1000 -s 500 -r -g -e -m -i",
                  "-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 500 -r -g -e -m -i",
                  "-c CCCTAAA -w 2000 -s 1000 -r -g -e -m -i", "-x 2 -w 1000 -s 500 -r -i -k 20 -d 100",
                  "-w 1000 -s 500 -i -t 700 -l 60 -k 10 -d 30 -y 0.8", "", "-t 300", "-t 1000 -l 60 -k 10 -d 30 -y 0.8"]


# ... and through the general kernels (generic.hip): mixed pattern lengths — the match stream carries a length per record —
# where the blocks are called on the device too (lengths at most one apart, w == s, tips-only), and a set with lengths 6 and
# 14 under w > s, whose stream is not in the reference's calling order and keeps the host path
GENERIC_BLOCKCALL_GRID = ["-p TTAGGG,TTAGG -w 1000 -s 500 -r -g -e -m -i", "-p TTAGGG,TTAGG,TTTAGGGTTTAGGG -x 1 -w 500 -s 500 -g -i",
                          "-p TTAGGG,TTAGG -t 400", "-c TTTAGGGTTTAGGG -x 1 -w 1000 -s 500 -r -g -i -k 80",
                          "-p TTAGGG,TTTAGGGTTTAGGG -x 0 -w 1000 -s 500 -r -g -i", "-w 1000 -s 997 -r -g -e -i"]


@pytest.mark.parametrize("host_blocks", [False, True])
@pytest.mark.parametrize("cli", GENERIC_BLOCKCALL_GRID)
def test_general_path_block_calling_matches_oracle(cli, host_blocks, monkeypatch):
    """The same through ts_scan_segments_blocks on parameter sets the general kernels take; host_blocks pins the host's block
    calling (TS_GEN_HOST_BLOCKS=1) so that both ways are compared with the oracle on the same input."""
    if host_blocks:
        monkeypatch.setenv("TS_GEN_HOST_BLOCKS", "1")
    test_device_block_calling_matches_oracle(cli)


# Streams that are NOT in position order: pattern lengths two or more apart under w > s — a long match across a window's end
# is pushed by the next window, behind shorter matches that begin after it (src/teloscope.cpp:485-509) — and the reference
# walks the stream as it lies.  The device writes the dense stream in push order (generic.hip: ts_general_compact_push) and
# calls blocks over it (blockcall.hip, MODE 1).  Nested telomeric patterns make every array a run of inversions at every
# window end; small steps (below the length spread) put several window ends into one inversion zone; wide sets take the
# wide form's records.
PUSH_ORDER_GRID = [
    "-p TTAGGG,TTAGGGTTAGGG,TTAGGGTTAGGGTTAGGG -x 0 -w 1000 -s 500 -r -g -i",
    "-p TTAGGG,TTAGGGTTAGGG,TTAGGGTTAGGGTTAGGG -x 0 -w 100 -s 7 -r -g -i -k 12",
    "-p TTAGGG,TTAGGGTT,TTAGGGTTAGGGT -x 0 -w 200 -s 100 -g -i -k 5 -d 40 -l 30",
    "-p TTAGGG,TTAGGGTTAGGGTTAGGGTTAGGGTTAGGG -x 0 -w 64 -s 3 -g -i",
    "-c TTAGGG -p TTAGGG,TTAGGGTTAGGGTTA -x 1 -w 512 -s 256 -r -g -e -i -t 3000",
    "-x 0 -p " + ",".join(("TTAGGG" * 7)[:k] for k in range(4, 21)) + " -w 1000 -s 500 -g -r -i",     # wide: 17 lengths
    "-x 0 -p " + ",".join(("TTAGGG" * 11)[:k] for k in (6, 9, 12, 15, 18, 24, 30, 36, 48, 63)) + " -w 300 -s 20 -g -i",   # wide, step < spread
]


@pytest.mark.parametrize("host_blocks", [False, True])
@pytest.mark.parametrize("cli", PUSH_ORDER_GRID)
def test_push_ordered_streams_blocks_match_oracle(cli, host_blocks, monkeypatch):
    if host_blocks:
        monkeypatch.setenv("TS_GEN_HOST_BLOCKS", "1")
    test_device_block_calling_matches_oracle(cli)


def test_push_ordered_sets_take_the_device_route(monkeypatch, capfd):
    """No parameter set is sent to the host's block calling any more unless TS_GEN_HOST_BLOCKS=1 asks for it (the library says
    which route a general-path call took under TS_TIMING=1)."""
    monkeypatch.setenv("TS_TIMING", "1")
    for cli, host in ((PUSH_ORDER_GRID[0], False), (PUSH_ORDER_GRID[5], False), (WIDE_GRID[3], False), (PUSH_ORDER_GRID[0], True)):
        if host:
            monkeypatch.setenv("TS_GEN_HOST_BLOCKS", "1")
        opts = H.parse_cli("x.fa " + cli)
        prod = ProductBackend(opts)
        rng = np.random.default_rng(3)
        s = seqgen.chromosome(rng, 30000, opts.canonical_fwd, opts.canonical_rev, telo_repeats=200, n_its=2)
        capfd.readouterr()
        prod.teloscope.scanSegmentsBlocksOnly([(s, 0)], tipsOnly=False, with_counts=True)
        err = capfd.readouterr().err
        assert ("blocks called on the host" in err) == host and ("blocks called on the device" in err) != host, err
        if not host and cli in PUSH_ORDER_GRID:
            assert "written in push order by the device" in err, err


@pytest.mark.parametrize("cli", PUSH_ORDER_GRID)
def test_push_ordered_streams_match_vectors_match_oracle(cli):
    """The same sets with the match vectors downloaded: allMatches, fwdMatches, ... in the reference's push order, straight
    from the device's stream (the host no longer orders anything), on segments whose arrays cross window ends and tile borders."""
    opts = H.parse_cli("x.fa " + cli)
    prod, orac = ProductBackend(opts), OracleBackend(opts)
    if orac.ambiguous:
        orac = orac.with_ambiguous_orientation_from(prod.patterns)
    rng = np.random.default_rng(len(cli) * 7 + 1)
    segs = []
    for i, n in enumerate([5, 70, 999, 4095, 4096, 4097, 4200, 8192, 12290, 40000, 131072 + 17]):
        s = bytearray(seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev,
                                        telo_repeats=min(300, max(1, n // 30)), tvr_rate=0.02, n_its=4, iupac=0))
        for at in (4096, 8192, 12288, 36864):                 # arrays across tile borders (4096 positions per tile)
            if n > at + 200:
                s[at - 150:at + 150] = (b"TTAGGG" * 50)[:300]
        # (absolute positions: the last two segments lie beyond 2^32 — block starts and match positions are 64-bit sums)
        ap = int(rng.integers(0, 10 ** 6)) + (5_000_000_000 if n >= 40000 else 0)
        segs.append((bytes(s), ap, False))
    got = prod.scan_segments(segs)
    for (s, ap, tips), g in zip(segs, got):
        assert_segment_equal(g, orac.scan_segment(s, ap, tips), tips, ctx="cli=%r len=%d" % (cli, len(s)))


@pytest.mark.parametrize("cli", BLOCKCALL_GRID)
def test_device_block_calling_matches_oracle(cli):
    """getTerminalBlocks / getInterstitialBlocks on the device (ts_batch_download_blocks): blocks
    and windows equal the oracle's, with no match record leaving the GPU."""
    from tests.backends import BLOCK_FIELDS, WINDOW_FIELDS
    opts = H.parse_cli("x.fa " + cli)
    prod, orac = ProductBackend(opts), OracleBackend(opts)
    rng = np.random.default_rng(len(cli) * 131 + 5)
    segs = []
    for i, n in enumerate([7, 600, 1999, 8000, 8001, 16500, 70001, 250003, 1_000_000]):
        s = seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev,
                              telo_repeats=min(400, max(1, n // 30)), tvr_rate=0.04, n_its=8,
                              iupac=(n // 5000) * (i % 2))
        if i % 3 == 0 and n > 3000:                          # an inverted telomere and a mid-contig tract
            s = s[:n // 2] + seqgen.repeat_array(opts.canonical_fwd, 40).tobytes() + s[n // 2 + 240:]
        segs.append((s, int(rng.integers(0, 10 ** 6))))
    got, counts = prod.teloscope.scanSegmentsBlocksOnly(segs, tipsOnly=opts.ultra_fast, with_counts=True)
    n_its = 0
    for (s, ap), g, cnt in zip(segs, got, counts):
        e = orac.scan_segment(s, ap, opts.ultra_fast)
        # the sizes the match vectors would have had, counted on the device
        nfwd, nrev = len(e["fwd_matches"]), len(e["rev_matches"])
        assert cnt[1] == nfwd + nrev and cnt[3] == nfwd, "%s counts len=%d" % (cli, len(s))
        if not opts.ultra_fast:
            assert cnt[0] == len(e["windows"]) and cnt[2] == len(e["canonical_matches"])
        for name, gb in (("terminal_blocks", g.terminalBlocks), ("interstitial_blocks", g.interstitialBlocks)):
            assert len(gb) == len(e[name]), "%s %s count len=%d" % (cli, name, len(s))
            for f in BLOCK_FIELDS:
                assert np.array_equal(gb[f], e[name][f]), "%s %s.%s len=%d" % (cli, name, f, len(s))
        for f in WINDOW_FIELDS:
            assert np.array_equal(g.windows[f], e["windows"][f])
        assert len(g.allMatches) == 0
        n_its += len(e["interstitial_blocks"])
    if not opts.ultra_fast:
        assert n_its > 0


def test_fuzz_parameters_and_degenerate_segments():
    """Seeded fuzz over the parameter space (canonical motif, edit distance, window/step, block
    thresholds, scan mode) with degenerate segments mixed in (empty, shorter than k, all N, all one
    base, IUPAC only, lower case only): whichever device path the library picks must equal the oracle."""
    rng = np.random.default_rng(2024)
    motifs = ["TTAGGG", "TTAGG", "CCCTAAA", "TTTTAGGG", "TTAGGGG", "TCAGG"]
    for it in range(16):
        c = motifs[int(rng.integers(0, len(motifs)))]
        w = int(rng.choice([len(c), 17, 64, 100, 333, 1000, 2000, 5000]))
        s = int(rng.integers(max(1, w // 40), w + 1)) if rng.random() < 0.5 else w
        tips = rng.random() < 0.25
        cli = "-c %s -x %d -w %d -s %d -t %d -k %d -d %d -l %d -y %.2f" % (
            c, int(rng.integers(0, 3)), w, s, int(rng.choice([50, 300, 5000, 50000])),
            int(rng.choice([5, 20, 50])), int(rng.choice([10, 100, 500])), int(rng.choice([12, 60, 300])),
            float(rng.choice([0.3, 0.5, 0.9])))
        if not tips:
            cli += " " + " ".join(rng.choice(["-r", "-g", "-e", "-m", "-i"], size=3, replace=False)) + " -g"
        opts = H.parse_cli("x.fa " + cli)
        if len(c) > w:
            continue
        prod, orac = ProductBackend(opts), OracleBackend(opts)
        if orac.ambiguous:
            # the same k-mer is reachable from both orientations (e.g. k = 5, -x 2): the reference's
            # flag then depends on std::sort's order among equal keys; give the oracle
            # the pattern list the product built so that the scan itself is what is compared
            assert [p for p, _, _ in prod.patterns] == [p for p, _, _, _ in orac.patterns]
            orac = orac.with_ambiguous_orientation_from(prod.patterns)
        else:
            assert [(p, f) for p, f, _ in prod.patterns] == [(p, f) for p, f, _, _ in orac.patterns]
        segs = [(b"", 0), (c[:-1].encode(), 3), (b"N" * 500, 0), (b"A" * 3000, 9), (b"RYKMSWBDHV" * 40, 1),
                (seqgen.chromosome(rng, 2500).lower(), 77), (c.encode() * 300, 5),
                (seqgen.chromosome(rng, int(rng.integers(1, 40000)), opts.canonical_fwd, opts.canonical_rev,
                                   telo_repeats=60, n_its=3, iupac=5, n_runs=2), int(rng.integers(0, 10 ** 9))),
                (seqgen.chromosome(rng, int(rng.integers(8000, 90000)), opts.canonical_fwd, opts.canonical_rev,
                                   telo_repeats=300, n_its=6), 0)]
        got = prod.scan_segments([(q, a, tips) for q, a in segs])
        for (q, a), g in zip(segs, got):
            seq_up = q.upper() if True else q                     # fold_case = 1 == unmaskSequence + scanSegment
            assert_segment_equal(g, orac.scan_segment(seq_up, a, tips), tips, ctx="fuzz %d cli=%r len=%d" % (it, cli, len(q)))


def test_concurrent_calls_on_one_context():
    """The reference calls scanSegment concurrently from its thread-pool workers on one shared Teloscope
    (src/input.cpp:719-724, 977); concurrent calls on one ts_ctx are coalesced inside the library and must give what
    sequential calls give."""
    import threading
    opts = H.parse_cli("x.fa -w 1000 -s 500 -r -g -e -m -i")
    prod, orac = ProductBackend(opts), OracleBackend(opts)
    rng = np.random.default_rng(77)
    jobs = [[(seqgen.chromosome(rng, int(rng.integers(2000, 300000)), telo_repeats=200, n_its=3), int(rng.integers(0, 10 ** 6)), False)
             for _ in range(3)] for _ in range(8)]
    results, errors = [None] * len(jobs), []

    def work(i):
        try:
            for _ in range(3):
                results[i] = prod.scan_segments(jobs[i])
        except Exception as e:                                  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for job, res in zip(jobs, results):
        for (s, ap, tips), g in zip(job, res):
            assert_segment_equal(g, orac.scan_segment(s, ap, tips), tips, ctx="concurrent len=%d" % len(s))


def test_sixty_four_concurrent_callers_cost_about_one_batched_call():
    """64 threads, one segment each, on ONE context (the literal drop-in of the reference's pool workers):
    every caller gets the oracle's result for its own segment, full scans, tips-only calls and read-filter calls mixed; and
    the 64 one-segment calls together take about as long as one call with the 64 segments (serialised one behind the
    other, as in round 2, they took 64 launches + 64 downloads)."""
    import threading
    import time
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.cli import user_input
    opts = H.parse_cli("x.fa -c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 500 -r -g -e -i")
    tel = ta.Teloscope(user_input(opts, device=0))
    orac = OracleBackend(opts)
    rng = np.random.default_rng(1234)
    segs = [(seqgen.chromosome(rng, int(rng.integers(150_000, 260_000)), opts.canonical_fwd, opts.canonical_rev, n_its=2), 1000 * i, i % 8 == 7)
            for i in range(64)]
    exp = [orac.scan_segment(s, a, t) for s, a, t in segs]
    tel.scanSegmentsBlocksOnly([(s, a) for s, a, t in segs if not t])                   # warm: buffers, pinned rings, code objects
    full = [(s, a) for s, a, t in segs if not t]
    tipsl = [(s, a) for s, a, t in segs if t]
    best_batched = None
    for _ in range(3):
        t0 = time.perf_counter()
        tel.scanSegmentsBlocksOnly(full)
        tel.scanSegmentsBlocksOnly(tipsl, tipsOnly=True)
        d = time.perf_counter() - t0
        best_batched = d if best_batched is None else min(best_batched, d)
    results, errors = [None] * 64, []
    gate = threading.Barrier(64)

    def work(i):
        try:
            s, a, t = segs[i]
            gate.wait()
            results[i] = tel.scanSegmentsBlocksOnly([(s, a)], tipsOnly=t)[0]
        except Exception as e:                                  # noqa: BLE001
            errors.append(e)

    best_threads = None
    for _ in range(3):
        threads = [threading.Thread(target=work, args=(i,)) for i in range(64)]
        for th in threads:
            th.start()
        t0 = time.perf_counter()
        for th in threads:
            th.join()
        d = time.perf_counter() - t0
        best_threads = d if best_threads is None else min(best_threads, d)
        assert not errors, errors
    from tests.backends import BLOCK_FIELDS, WINDOW_FIELDS
    for i, (g, e) in enumerate(zip(results, exp)):
        for f in WINDOW_FIELDS:
            assert np.array_equal(g.windows[f], e["windows"][f]), (i, f)
        for f in BLOCK_FIELDS:
            assert np.array_equal(g.terminalBlocks[f], e["terminal_blocks"][f]), (i, f)
            assert np.array_equal(g.interstitialBlocks[f], e["interstitial_blocks"][f]), (i, f)
    print("64 one-segment calls: %.1f ms; the same segments in two batched calls: %.1f ms" % (best_threads * 1e3, best_batched * 1e3))
    assert best_threads <= 3.0 * best_batched + 0.010, (best_threads, best_batched)
    # the read filter's callers are coalesced the same way
    ropts = H.parse_cli("--fastq-subset -l 42")
    rf = ta.ReadTelomereFilter(user_input(ropts, device=0))
    reads = [seqgen.chromosome(rng, int(rng.integers(2000, 20000)), n_its=1) if i % 3 else bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=9000)) for i in range(64)]
    want = OracleReadFilter(ropts).filter(reads)
    got = [None] * 64

    def rwork(i):
        gate.wait()
        got[i] = rf.matchesBatch([reads[i]])[0]

    threads = [threading.Thread(target=rwork, args=(i,)) for i in range(64)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert got == want


TILINGS = ["16,1", "16,3", "12,5", "10,6", "10,3", "8,2", "4,7", "2,4", "1,1", "1,8"]      # 10 waves = two workgroups per CU


def test_ragged_fasta_text_equals_joined_bases():
    """TS_INPUT_TEXT_PIECES through the staging threads' 32-byte walks (pipeline.cpp: text_locate / strip_take): lines of every
    width from 1 to 130, LF and CRLF mixed, blank lines, a stray carriage return inside a line (a byte that is not a base), pieces
    cut anywhere — between a carriage return and its line feed too — and a segment long enough that every staging thread enters
    its pieces in the middle.  The same bases joined give the same segment, which is the oracle's."""
    import ctypes as C
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.cli import user_input
    cli = "-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 500 -r -g -e -m -i"
    opts = H.parse_cli("x.fa " + cli)
    tel = ta.Teloscope(user_input(opts))
    L = K.lib()
    rng = np.random.default_rng(20261005)
    seq = bytearray(seqgen.chromosome(rng, 3_300_017, opts.canonical_fwd, opts.canonical_rev, n_its=6, iupac=4, lower=0.02))
    for at in rng.integers(1000, len(seq) - 1000, 40):                 # stray carriage returns: bytes like any other non-base
        seq[int(at)] = 13
    seq = bytes(seq)
    out, at, lines = [], 0, 0
    while at < len(seq):
        w = int(rng.integers(1, 131)) if lines % 7 else 80
        line = seq[at:at + w]
        if line.endswith(b"\r") and at + w < len(seq):                 # (a line's own last byte is never a bare carriage return)
            line = line[:-1] if len(line) > 1 else seq[at:at + 2]
        at += len(line)
        out.append(line + (b"\r\n" if rng.random() < 0.3 else b"\n"))
        if rng.random() < 0.01:
            out.append(b"\n")
        lines += 1
    text = b"".join(out)
    cuts = sorted(set([0, len(text)] + [int(x) for x in rng.integers(1, len(text) - 1, 9)]))
    blobs = [text[a:b] for a, b in zip(cuts, cuts[1:])]

    def bases_of(blob, last):
        b = blob.replace(b"\r\n", b"\n").replace(b"\n", b"")
        return len(b) - (1 if b.endswith(b"\r") and not last else 0)     # a '\r' at the very end of a piece belongs to a line end
    # (a stray '\r' must not end a piece: move such cuts one byte on)
    fixed = [0]
    for c in cuts[1:-1]:
        while text[c - 1:c] == b"\r" and text[c:c + 1] != b"\n":
            c += 1
        fixed.append(c)
    cuts = sorted(set(fixed + [len(text)]))
    blobs = [text[a:b] for a, b in zip(cuts, cuts[1:])]
    pieces = (K.TextPiece * len(blobs))()
    keep = []
    for i, bl in enumerate(blobs):
        buf = C.create_string_buffer(bl, len(bl))
        keep.append(buf)
        pieces[i].text = C.cast(buf, C.c_char_p)
        pieces[i].text_len = len(bl)
        pieces[i].n_bases = bases_of(bl, i == len(blobs) - 1)
    assert sum(p.n_bases for p in pieces) == len(seq)
    seg = (K.SegmentIn * 1)()
    seg[0].seq = C.cast(pieces, C.c_char_p)
    seg[0].len = len(seq)
    seg[0].input_format = K.TS_INPUT_TEXT_PIECES
    seg[0].n_pieces = len(blobs)
    res = (K.SegmentOut * 1)()
    assert L.ts_scan_segments(tel._ctx.ptr, seg, 1, res) == 0, tel._ctx.error()
    got = segment_as_dict(ta.SegmentData(res[0], False))
    L.ts_free_segments(res, 1)
    want = OracleBackend(opts).scan_segment(seq.upper(), 0, False)
    assert_segment_equal(got, want, False, ctx="ragged FASTA text")


@pytest.mark.parametrize("cli", ["-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 500 -r -g -e -m -i",
                                 "-r -g -e -m -i", "-c CCCTAAA -w 2000 -s 1000 -r -g -e -m -i",
                                 "-w 300 -s 100 -g -e -m -i", "-t 3000"])
def test_output_independent_of_tiling(cli, monkeypatch):
    """SURVEY 8b determinism contract: the output does not depend on the tile size or the waves per
    workgroup.  TS_GEOMETRY pins the planner to one (waves, chunks per tile) point; every tiling the
    parameter set admits must reproduce the oracle bit for bit (same windows, matches, blocks)."""
    opts = H.parse_cli("x.fa " + cli)
    orac = OracleBackend(opts)
    rng = np.random.default_rng(len(cli) * 31 + 5)
    unit_f, unit_r = opts.canonical_fwd, opts.canonical_rev
    segs = []
    for i, n in enumerate([1, 999, 2016, 4033, 30011, 64513, 129023, 400003]):
        s = seqgen.chromosome(rng, n, unit_f, unit_r, telo_repeats=min(300, max(1, n // 40)),
                              tvr_rate=0.03, n_its=4, iupac=(n // 20000), lower=0.01, n_runs=i % 2)
        segs.append((s, int(rng.integers(0, 10 ** 6)), opts.ultra_fast))
    want = [orac.scan_segment(bytes(s).upper(), ap, tips) for s, ap, tips in segs]   # the product folds case
    ran = 0
    for tiling in TILINGS:
        monkeypatch.setenv("TS_GEOMETRY", tiling)
        try:
            prod = ProductBackend(opts)
            got = prod.scan_segments(segs)
        except Exception as exc:                   # this tiling does not fit the LDS for this window / step
            assert "unsupported" in str(exc) or "does not fit" in str(exc), exc
            continue
        ran += 1
        for (s, ap, tips), g, e in zip(segs, got, want):
            assert_segment_equal(g, e, tips, ctx="tiling=%s cli=%r len=%d" % (tiling, cli, len(s)))
    assert ran >= 4, "too few tilings ran for %r" % cli
    # round 5's forms of the per-match pass against round 4's, which the planner keeps for geometries that need them: 32-bit
    # stage entries (TS_STAGE_U32=1: tiles above 2^14 positions take them anyway) and one accumulator row per window
    # (TS_ACC_PER_WINDOW=1: what a window that is not a multiple of the step takes) — the same bits either way
    monkeypatch.delenv("TS_GEOMETRY", raising=False)
    for env in ("TS_STAGE_U32", "TS_ACC_PER_WINDOW", "TS_REC32"):             # (TS_REC32: 32-bit records in the regions; the default is 16)
        monkeypatch.setenv(env, "1")
        got = ProductBackend(opts).scan_segments(segs)
        monkeypatch.delenv(env)
        for (s, ap, tips), g, e in zip(segs, got, want):
            assert_segment_equal(g, e, tips, ctx="%s=1 cli=%r len=%d" % (env, cli, len(s)))


@pytest.mark.parametrize("cli", ["-l 42", "-l 300 -k 30 -d 200 -y 0.7", "-x 0 -l 100 -k 12 -d 2000 -y 0.3",
                                 "-c CCCTAAA -l 500 -k 60 -d 50 -y 0.5"])
def test_read_filter_long_match_lists(cli):
    """Reads whose match lists are far longer than the 384 records above which the predicate kernel walks a
    read with the whole wave, 64 records per step in parallel: several tracts of both orientations, clean and
    noisy, separated by gaps around -k (sub-block chaining) and -d (block merging), at either end or inside."""
    opts = H.parse_cli("--fastq-subset " + cli)
    rng = np.random.default_rng(len(cli) * 101 + 3)
    k_gap = opts.max_match_dist
    d_gap = opts.max_block_dist
    reads = []
    for i in range(240):
        n = int(rng.integers(30000, 260000))
        s = bytearray(seqgen.random_dna(rng, n).tobytes())
        at = 0 if i % 3 == 0 else int(rng.integers(0, n // 2))
        for _ in range(int(rng.integers(1, 7))):
            unit = opts.canonical_fwd if rng.random() < 0.5 else opts.canonical_rev
            ln = int(rng.integers(2, 1500)) * len(unit)
            t = seqgen.mutate(rng, seqgen.repeat_array(unit, ln // len(unit)), float(rng.choice([0.0, 0.01, 0.08, 0.2]))).tobytes()
            if at + len(t) >= n:
                break
            s[at:at + len(t)] = t
            gap = int(rng.choice([0, 1, k_gap - 1, k_gap, k_gap + 1, k_gap + 7, d_gap - 1, d_gap, d_gap + 1, d_gap + 9, 3000]))
            at += len(t) + max(0, gap)
        if i % 4 == 1:                                          # the same at the far end
            s = bytearray(bytes(s)[::-1])
        reads.append(bytes(s))
    got = ProductReadFilter(opts).filter(reads)
    exp = OracleReadFilter(opts).filter(reads)
    assert got == exp
    assert 0 < sum(got) < len(got)


def test_read_records_of_16_and_of_32_bits_give_the_same_pass_bytes(monkeypatch):
    """ts_batch_set_record_bits: ts_filter_reads keeps its batches' records at 16 bits (the predicate is their only reader);
    TS_REC32=1 (read when the context is made) keeps 32.  Same pass bytes, the oracle's; a batch that cannot have 16-bit records
    says so and stays at 32, and a batch that has them gives no raw view of them (every reader inside the library knows both
    widths: test_output_independent_of_tiling runs the window scan's readers with TS_REC32=1 as well)."""
    import ctypes as C
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.cli import user_input
    opts = H.parse_cli("--fastq-subset -l 42")
    rng = np.random.default_rng(77)
    reads = []
    for i in range(700):
        n = int(rng.integers(200, 40000))
        s = bytearray(seqgen.random_dna(rng, n).tobytes())
        if i % 5 == 0:
            unit = opts.canonical_fwd if i % 2 else opts.canonical_rev
            t = seqgen.mutate(rng, seqgen.repeat_array(unit, int(rng.integers(3, 400))), 0.02).tobytes()
            at = 0 if i % 3 else max(0, n - len(t))
            s[at:at + len(t)] = t[:n - at]
        reads.append(bytes(s))
    exp = OracleReadFilter(opts).filter(reads)
    got16 = ProductReadFilter(opts).filter(reads)
    monkeypatch.setenv("TS_REC32", "1")
    got32 = ProductReadFilter(opts).filter(reads)
    monkeypatch.delenv("TS_REC32")
    assert got16 == exp and got32 == exp and 0 < sum(exp) < len(exp)
    # the batch interface
    L = K.lib()
    rf = ta.ReadTelomereFilter(user_input(opts))
    lens = (C.c_uint64 * 3)(5000, 17000, 900)
    b = L.ts_batch_create(rf._ctx.ptr, lens, None, 3, 1, 0)
    assert b and L.ts_batch_set_record_bits(b, 16) == 0
    assert not L.ts_batch_matches_ptr(b)                                         # no raw view of 16-bit records
    assert L.ts_batch_set_record_bits(b, 32) == 0 and L.ts_batch_set_record_bits(b, 24) == K.TS_ERR_INVALID_ARG
    L.ts_batch_destroy(b)
    tel = ta.Teloscope(user_input(H.parse_cli("x.fa -r")))
    full = L.ts_batch_create(tel._ctx.ptr, lens, None, 3, 0, 0)
    assert full and not L.ts_batch_matches_ptr(full)                              # a window scan's records are 16 bits by default as well
    assert L.ts_batch_set_record_bits(full, 32) == 0                              # ... unless the caller wants the raw 32-bit view
    L.ts_batch_destroy(full)
