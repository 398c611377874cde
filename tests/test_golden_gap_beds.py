"""The reference's expected gap BEDs (testFiles/expected/*_gaps.bed, written by
Teloscope::generateBEDFile, src/teloscope.cpp:919 ff.) pin how a FASTA record is cut into segments
and gaps before scanSegment is called: every run of N is one gap, coordinates are 0-based half-open
on the path.  The harness's path walk (tests/harness.py: path_components / walk_path — the code that
feeds segments to both the oracle and the GPU path) must reproduce them."""
import glob
import os

import pytest

from tests import harness as H
from tests.backends import OracleBackend

BEDS = sorted(glob.glob(os.path.join(H.golden_path("testFiles/expected"), "*_gaps.bed")))


@pytest.mark.parametrize("bed", BEDS, ids=[os.path.basename(b) for b in BEDS])
def test_gap_bed_matches_reference(bed):
    fasta = H.golden_path("testFiles/" + os.path.basename(bed)[:-len("_gaps.bed")])
    opts = H.parse_cli("%s -c TTAGGG" % fasta)
    backend = OracleBackend(opts)
    lines = []
    for i, (header, seq) in enumerate(H.read_fasta(fasta)):
        pd = H.walk_path(backend, opts, i, header, seq)
        name = pd["header"].split()[0]
        for start, length in pd["gaps"]:
            lines.append("%s\t%d\t%d" % (name, start, start + length))
    with open(bed) as fh:
        expected = [l.rstrip("\n") for l in fh if l.strip()]
    assert lines == expected
