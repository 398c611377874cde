"""Pins the CPU oracle against the reference's own expectations: every legacy `.tst`
manifest of validateFiles/ (exact CLI stdout: per-path telomere labels, scaffold type,
ITS / canonical-match / window counts, telomere length statistics) and every FASTQ-subset
manifest (exact passing records, in order).  These are the only golden outputs the
reference ships for the scan path."""
import glob
import os

import pytest

from tests import harness as H
from tests.backends import OracleBackend, OracleReadFilter

MANIFESTS = sorted(glob.glob(os.path.join(H.GOLDEN, "validateFiles", "*.tst")))
LEGACY = [m for m in MANIFESTS if H.load_manifest(m)["mode"] == "embedded"]
FASTQ = [m for m in MANIFESTS if os.path.basename(m).startswith("fastq_subset")]


def test_manifest_inventory():
    assert len(MANIFESTS) == 173
    assert len(LEGACY) >= 140
    assert len(FASTQ) == 12


@pytest.mark.parametrize("path", LEGACY, ids=[os.path.basename(p) for p in LEGACY])
def test_legacy_manifest_stdout(path):
    m = H.load_manifest(path)
    opts = H.parse_cli(m["command"])
    assert not opts.fastq_subset
    be = OracleBackend(opts)
    stdout, _ = H.run_assembly(be, opts, H.golden_path(opts.input))
    assert stdout.split("\n") == m["expected"].split("\n")


@pytest.mark.parametrize("path", FASTQ, ids=[os.path.basename(p) for p in FASTQ])
def test_fastq_manifest(path):
    m = H.load_manifest(path)
    d = dict()
    for k, v in m["directives"]:
        d.setdefault(k, []).append(v)
    opts = H.parse_cli(m["command"])
    assert opts.fastq_subset
    src = opts.input or opts.stdin_redirect
    src = H.golden_path(src)
    if src.endswith(".gz"):
        import gzip
        data = gzip.open(src, "rb").read()
    else:
        data = open(src, "rb").read()
    expect_exit = int(d["expect_exit"][0])
    rf = OracleReadFilter(opts)
    try:
        out, kept, total = H.run_fastq_subset(rf, data)
        code = 0
    except ValueError as e:
        code = 1
        for sub in d.get("expect_stderr_substr", []):
            assert sub in str(e)
        out, kept, total = b"", 0, 0
    assert code == expect_exit
    if code == 0:
        so = d.get("expect_stdout", ["ignore"])[0]
        if so != "ignore" and " -o " in m["command"]:
            # records go to a file under -o; stdout stays empty
            assert open(H.golden_path(so), "rb").read() == b""
            assert out == open(H.golden_path("testFiles/expected/fastq_subset.fq"), "rb").read()
        elif so != "ignore":
            assert out == open(H.golden_path(so), "rb").read()
        for sub in d.get("expect_stderr_substr", []):
            if sub.startswith("FASTQ subset: kept"):
                assert sub == "FASTQ subset: kept %d of %d reads." % (kept, total)
