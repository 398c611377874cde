"""The reference's GFA manifests pin the tips-only scan + terminal block length per segment end: every
`validateFiles/gfa_*.tst` names a `testFiles/expected/gfa/*.tsv` whose rows carry `tl_bp`, the length of the telomere
node attached to that end (scanSegment(seq, 0, true) -> terminalBlocks -> longest block at the end, src/input.cpp:835-939).
Here: the ORACLE against those rows (the GPU counterpart is in tests/test_gpu_parity.py)."""
import glob
import os

import pytest

from tests import harness as H
from tests.backends import OracleBackend

GFA = []
for m in sorted(glob.glob(os.path.join(H.GOLDEN, "validateFiles", "gfa*.tst"))):
    man = H.load_manifest(m)
    exp = [v for k, v in man["directives"] if k == "gfa_expect"]
    if exp:
        GFA.append((os.path.basename(m), man["command"], exp[0]))


def gfa_case(command):
    toks = command.split()
    return " ".join(t for t in toks if t != "%OUTDIR%" and t != "-o")


@pytest.mark.parametrize("name,command,expect", GFA, ids=[g[0] for g in GFA])
def test_oracle_reproduces_gfa_telomere_lengths(name, command, expect):
    opts = H.parse_cli(gfa_case(command))
    got = H.gfa_annotations(OracleBackend(opts), opts, H.golden_path(opts.input))
    assert got == H.read_gfa_expectation(H.golden_path(expect))


def test_all_thirteen_expectations_are_used():
    assert len({g[2] for g in GFA}) >= 13 and len(GFA) >= 15
