"""The extended seeded fuzzes (profiles/fuzz_long.py, fuzz_reads.py, fuzz_shards.py) under the driver's eyes.

Until round 5 they were builder-run logs under profiles/rNN/ (1 483 + 550 parameter sets, 300 read-filter sets, 300 shard
layouts against the oracle); tests/test_gpu_parity.py ran 16 iterations of the first.  Here each runs with a fixed seed as a
child process (they are scripts: a parameter set per iteration, every segment of it HIP == oracle record for record):

  fuzz_long.py    scan parameters x degenerate segments — uniform-k sets (tiled kernel), mixed-length sets (general kernels'
                  list / strided forms) and, TS_FUZZ_WIDE, a share of wide sets (beyond 8 lengths / 32 bases) — windows, all
                  five match vectors, blocks, the blocks-only entry point, packed uploads
  fuzz_reads.py   read-filter parameters x reads of every kind the predicate kernels tell apart
  fuzz_shards.py  random layouts x 1-9 parts: merged shard messages == oracle, or TS_SHARD_NEED_FULL — never a wrong answer
"""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, args, env=None, timeout=400):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", script)] + [str(a) for a in args],
                       capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)
    assert r.returncode == 0, (script, r.stdout[-2000:], r.stderr[-4000:])
    return r.stdout


def test_fuzz_scan_parameters_300_sets_a_tenth_of_them_wide():
    out = _run("fuzz_long.py", [340, 20261005], env={"TS_FUZZ_WIDE": "0.16"})
    m = re.search(r"fuzz: all (\d+) parameter sets equal the oracle", out)
    assert m and int(m.group(1)) >= 300, out[-1500:]
    w = re.search(r"(\d+) wide", out)
    assert w and int(w.group(1)) * 10 >= int(m.group(1)), out[-1500:]


def test_fuzz_read_filter_100_sets():
    out = _run("fuzz_reads.py", [110, 77])
    m = re.search(r"fuzz_reads: all (\d+) parameter sets", out)
    assert m and int(m.group(1)) >= 100, out[-1500:]


def test_fuzz_shard_layouts_100():
    out = _run("fuzz_shards.py", [110, 99])
    m = re.search(r"shard fuzz: all (\d+) layouts equal the oracle", out)
    assert m and int(m.group(1)) >= 100, out[-1500:]
