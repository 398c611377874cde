"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/teloscan.h
declares, its host-only entry points (pattern expansion, labelling, float metrics) agree with
the oracle, and — without a GPU — context creation fails loudly instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tests import harness as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ta():
    import teloscope_amd
    return teloscope_amd


def test_header_symbols_exported(ta):
    from teloscope_amd import _capi
    hdr = open(os.path.join(ROOT, "include", "teloscan.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)          # declarations only, not comments
    declared = set(re.findall(r"\b(ts_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_capi.SYMBOLS)
    lib = C.CDLL(_capi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert lib.ts_abi_version() == 4


def test_struct_sizes_match_header(ta, tmp_path):
    """ctypes mirrors vs the C compiler's view of include/teloscan.h."""
    import subprocess
    from teloscope_amd import _capi as K
    names = ["ts_params", "ts_match", "ts_window", "ts_block", "ts_pattern", "ts_segment_in",
             "ts_segment_out", "ts_batch_info", "ts_tile_info", "ts_range_info"]
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "teloscan.h"\nint main(void){' +
                   "".join('printf("%%zu\\n", sizeof(%s));' % n for n in names) + "return 0;}")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    mirrors = [K.Params, K.Match, K.Window, K.Block, K.Pattern, K.SegmentIn, K.SegmentOut, K.BatchInfo, K.TileInfo, K.RangeInfo]
    assert sizes == [C.sizeof(m) for m in mirrors]


PATTERN_CASES = [
    (["CCCTAA", "TTAGGG"], 1, "CCCTAA", 38),
    (["CCCTAA", "TTAGGG"], 0, "CCCTAA", 2),
    (["CCCTAA", "TTAGGG"], 2, "CCCTAA", 308),
    (["TTAGGG", "TCAGGG", "TGAGGG", "TTGGGG"], 1, "CCCTAA", 124),
    (["CCCTAAA", "TTTAGGG"], 1, "CCCTAAA", 44),
    (["TTAGGN"], 0, "CCCTAA", None),
    (["TTRGGG", "CCCTAA"], 1, "CCCTAA", None),
    (["TTAGG", "TTAGGG"], 1, "CCCTAA", None),
]


@pytest.mark.parametrize("raw,ed,can,count", PATTERN_CASES)
def test_pattern_expansion_matches_oracle(ta, oracle_lib, raw, ed, can, count):
    """expandPatternsWithOrientation (src/tools.cpp:201-283): product C++ vs oracle C, and the
    pattern counts the reference prints ("Scanning N telomeric variants")."""
    got = ta.expandPatternsWithOrientation(raw, ed, can)
    exp = oracle_lib.expand_patterns(raw, ed, can)
    assert [(p, f) for p, f, _, _ in exp] == got
    assert not any(a for _, _, _, a in exp)
    if count is not None:
        assert len(got) == count


def test_canonical_orientation(ta):
    assert ta.canonicalOrientation("TTAGGG") == ("CCCTAA", "TTAGGG")
    assert ta.canonicalOrientation("ccctaaa") == ("CCCTAAA", "TTTAGGG")


def test_float_metrics_match_oracle(ta, oracle_lib):
    rng = np.random.default_rng(7)
    for _ in range(2000):
        size = int(rng.integers(1, 5000))
        c = rng.multinomial(int(rng.integers(0, size + 1)), [0.3, 0.2, 0.2, 0.3])
        a = ta.getGCContent(c, size), ta.getShannonEntropy(c, size)
        b = oracle_lib.gc_content(c, size), oracle_lib.shannon_entropy(c, size)
        assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()


def test_label_terminal_blocks_matches_oracle(ta, oracle_lib):
    from teloscope_amd import _capi as K
    rng = np.random.default_rng(11)
    for _ in range(500):
        n = int(rng.integers(0, 6))
        path_size = int(rng.integers(100, 200000))
        blocks = np.zeros(n, dtype=K.BLOCK_DT)
        ob = np.zeros(n, dtype=oracle_lib.BLOCK_DT)
        for i in range(n):
            start = int(rng.integers(0, path_size))
            vals = dict(start=start, block_len=int(rng.integers(1, 5000)),
                        can_covered=int(rng.integers(1, 4000)), has_valid_or=int(rng.integers(0, 2)),
                        block_label=(b"p", b"q")[int(rng.integers(0, 2))])
            for k, v in vals.items():
                blocks[i][k] = v
                ob[i][k] = v
        gaps = int(rng.integers(0, 2))
        tl = int(rng.choice([100, 50000]))
        gb, glabel, gtype = ta.Teloscope.labelTerminalBlocks(blocks, gaps, path_size, tl)
        eb, elabel, etype = oracle_lib.label_terminal_blocks(ob, gaps, path_size, tl)
        assert (glabel, gtype) == (elabel, etype)
        assert np.array_equal(gb["is_longest"], eb["is_longest"])
        assert np.array_equal(gb["start"], eb["start"])


def test_no_cpu_fallback(ta):
    """Without a HIP device the product refuses to run (no silent CPU path)."""
    from teloscope_amd import _capi as K
    if K.lib().ts_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ta.TeloscanError) as ei:
        ta.Teloscope(ta.UserInputTeloscope())
    assert ei.value.code == K.TS_ERR_NO_DEVICE


def test_product_does_not_reference_oracle():
    """teloscope_amd/ must never import, link or load anything from oracle/."""
    for dp, _, files in os.walk(os.path.join(ROOT, "teloscope_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.lower().replace("no cpu", ""), os.path.join(dp, f)


def test_bind_thread_needs_a_device(ta):
    """ts_bind_thread_to_device: a planning-only context has no device, hence no NUMA node to go to — 0, and the calling
    thread's CPU mask is left alone (null context likewise)."""
    import os
    from teloscope_amd import _capi as K
    from teloscope_amd.cli import parse_cli, user_input
    tel = ta.Teloscope(user_input(parse_cli("x.fa -r"), device=K.DEVICE_NONE))
    before = os.sched_getaffinity(0)
    assert K.lib().ts_bind_thread_to_device(tel._ctx.ptr) == 0
    assert K.lib().ts_bind_thread_to_device(None) == 0
    assert os.sched_getaffinity(0) == before


def test_rank_exchange_needs_a_device_and_sane_arguments(ta):
    """ts_exchange_*: no communicator on a planning-only context (TS_ERR_NO_DEVICE in ts_last_error, NULL returned), null
    arguments are refused, destroying nothing is fine — none of which needs librccl or a GPU."""
    import ctypes as C
    from teloscope_amd import _capi as K
    from teloscope_amd.cli import parse_cli, user_input
    L = K.lib()
    tel = ta.Teloscope(user_input(parse_cli("x.fa -r"), device=K.DEVICE_NONE))
    ident = (C.c_char * 128)()
    assert not L.ts_exchange_create(tel._ctx.ptr, ident, 0, 1)
    assert "planning-only" in tel._ctx.error()
    assert not L.ts_exchange_create(tel._ctx.ptr, None, 0, 1)
    assert not L.ts_exchange_create(tel._ctx.ptr, ident, 2, 2)
    assert not L.ts_exchange_create(None, ident, 0, 1)
    assert L.ts_exchange_unique_id(None) == K.TS_ERR_INVALID_ARG
    assert L.ts_exchange_gather(None, 0, None, 0, None, None, None) == K.TS_ERR_INVALID_ARG
    L.ts_exchange_destroy(None)


def test_rank_exchange_without_librccl_is_unsupported_not_a_crash():
    """A host without librccl: ts_exchange_unique_id answers TS_ERR_UNSUPPORTED and ts_exchange_last_error says which
    library was missing (include/teloscan.h promises exactly that).  TS_RCCL_LIB points the run-time loader at a file that
    does not exist; the library is opened once per process, hence the child process."""
    import subprocess
    import sys
    code = ("import ctypes as C, sys\n"
            "sys.path.insert(0, %r)\n"
            "from teloscope_amd import _capi as K\n"
            "L = K.lib()\n"
            "ident = (C.c_char * 128)()\n"
            "rc = L.ts_exchange_unique_id(ident)\n"
            "msg = L.ts_exchange_last_error().decode()\n"
            "assert rc == K.TS_ERR_UNSUPPORTED, rc\n"
            "assert 'librccl not found' in msg and 'no_such_rccl' in msg, msg\n"
            "assert L.ts_exchange_unique_id(ident) == K.TS_ERR_UNSUPPORTED\n"
            "print('ok')\n") % ROOT
    env = dict(os.environ, TS_RCCL_LIB="/nonexistent/libno_such_rccl.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_planner_tiling_choices(ta, monkeypatch):
    """plan_geometry's choice between one workgroup of 16 waves per CU and two of 10 (the 80-VGPR build of the scan kernel),
    read off the plan of a planning-only context (no GPU): windows per tile = what the chosen chunks per tile hold.  Two
    workgroups (6 chunks: 23 windows at -w 1000 -s 500) only where the pattern set is dense (patterns / 4^k >= 1.5 %), has
    byte tables (k <= 6) and tiles of six chunks fit; one workgroup of 16 waves (8 chunks: 31) otherwise.  The measurements
    behind the rule: profiles/r02/occupancy_sweep.txt, planner_check.txt."""
    from teloscope_amd import _capi as K
    from teloscope_amd.cli import parse_cli, user_input
    from teloscope_amd.distributed import ShardPlan
    monkeypatch.delenv("TS_GEOMETRY", raising=False)

    def windows_per_tile(cli, n=50_000_000):
        tel = ta.Teloscope(user_input(parse_cli("x.fa " + cli), device=K.DEVICE_NONE))
        plan = ShardPlan(tel, [n], world=1)
        wpt = -(-int(plan.info.n_windows) // int(plan.info.n_tiles))
        plan.close()
        return wpt

    headline = "-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -x 1 -w 1000 -s 500 -r -g -e -m -i"
    assert windows_per_tile(headline) == 23                               # 124 patterns of 4096: two workgroups of ten waves
    assert windows_per_tile("-c TTAGGG -x 0 -w 1000 -s 500 -g -e -i") == 31     # 2 patterns: sparse, one workgroup of 16
    assert windows_per_tile("-c TTAGGG -r -g -e -m -i") == 16                   # default flags (38 patterns, w = s = 1000): sparse
    assert windows_per_tile("-c CCCTAAA -w 2000 -s 1000 -r -g -e -m -i") == 13  # k = 7: the 64 KB byte table, 7 chunks, one workgroup
    monkeypatch.setenv("TS_GEOMETRY", "16,0")                             # the search pinned to one workgroup of 16 waves
    assert windows_per_tile(headline) == 31
    monkeypatch.setenv("TS_GEOMETRY", "10,6")
    assert windows_per_tile("-c TTAGGG -r -g -e -m -i") == 12


def test_pack_bases_against_a_numpy_statement_of_it():
    """ts_pack_bases (the packed upload's host half; AVX2 with a scalar tail): 2-bit codes and invalid runs of random
    buffers of every length around the vector widths, with and without case folding, against numpy."""
    from teloscope_amd import _capi as K
    L = K.lib()
    rng = np.random.default_rng(5)
    alpha = np.frombuffer(b"ACGTacgtNRYKMnX-*\x00\xff", dtype=np.uint8)
    code_of = {ord("A"): 0, ord("C"): 1, ord("T"): 2, ord("G"): 3}
    for n in list(range(0, 140)) + [255, 256, 257, 4095, 4096, 4097, 100_003]:
        for fold in (0, 1):
            for clean in (False, True):
                s = (alpha[:4] if clean else alpha)[rng.integers(0, 4 if clean else len(alpha), size=n)].astype(np.uint8)
                dst = np.full((n + 3) // 4 + 8, 0xEE, dtype=np.uint8)
                runs = np.zeros((n + 1, 2), dtype=np.uint32)
                nr = C.c_uint64(0)
                rc = L.ts_pack_bases(s.tobytes(), n, fold, dst.ctypes.data, runs.ctypes.data, n + 1, C.byref(nr))
                assert rc == 0
                f = np.where(fold, s & 0xDF, s).astype(np.uint8) if n else s
                valid = np.isin(f, np.frombuffer(b"ACGT", dtype=np.uint8))
                codes = np.zeros(n, dtype=np.uint8)
                for ch, cd in code_of.items():
                    codes[f == ch] = cd
                codes[~valid] = 0
                pad = np.concatenate([codes, np.zeros((-n) % 4, dtype=np.uint8)]).reshape(-1, 4)
                want = (pad[:, 0] | (pad[:, 1] << 2) | (pad[:, 2] << 4) | (pad[:, 3] << 6)).astype(np.uint8)
                assert np.array_equal(dst[:len(want)], want), (n, fold, clean)
                assert (dst[len(want):] == 0xEE).all(), "wrote past the packed bytes"
                inv = np.zeros(n, dtype=bool)
                last_end = -1
                for a, ln in runs[:nr.value]:
                    assert ln > 0 and int(a) > last_end, "runs must ascend, be maximal and not touch"
                    inv[a:a + ln] = True
                    last_end = int(a) + int(ln)
                assert np.array_equal(inv, ~valid), (n, fold, clean)
    # a run list that is too short is an error, and says how many runs there are
    nr = C.c_uint64(0)
    dst = np.zeros(8, dtype=np.uint8)
    assert L.ts_pack_bases(b"ANANANAN", 8, 1, dst.ctypes.data, None, 0, C.byref(nr)) == K.TS_ERR_INVALID_ARG and nr.value == 4
