"""Row f2: the output files of handleBEDFile / writeBEDFile / printSummary
(src/teloscope.cpp:661-1055), written by include/teloscope_mi355x_io.hpp from GPU results, against an
independent restatement of the formats (tests/harness.py: format_bed_files) fed by the CPU oracle.
The reference ships expected files only for the gap BEDs (testFiles/expected/*_gaps.bed) and, through
the manifests' stdout, for the report rows; those are checked directly."""
import glob
import os
import shlex
import subprocess

import pytest

from tests import harness as H
from tests.backends import OracleBackend
from tests.test_cpp_mirror import cli  # noqa: F401  (fixture: builds tests/cpp/manifest_cli.cpp)

CASES = [
    ("t2t.fa", "-r -g -e -m -i"),
    ("t2t.fa", ""),                                            # ultra-fast: terminal, gaps, report only
    ("gapped_t2t.fa", "-w 500 -s 250 -r -g -e -m -i"),
    ("multi_gap_t2t.fa", "-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 500 -r -g -e -m -i"),
    ("multi.fa", "-r -i"),
    ("discordant.fa", "-i -m -n"),
    ("boundary_multiple_p.fa", "-r -g -i -n -t 700"),
    ("bTaeGut7_chr33_mat.fa.gz", "-w 2000 -s 1000 -r -g -e -i"),
]


def expected_files(fasta, flags):
    opts = H.parse_cli("%s %s" % (fasta, flags))
    backend = OracleBackend(opts)
    records = H.read_fasta(fasta)
    paths = [H.walk_path(backend, opts, i, h, s) for i, (h, s) in enumerate(records)]
    return H.format_bed_files(paths, records, opts), H.format_report(paths, opts)


@pytest.mark.gpu
@pytest.mark.parametrize("name,flags", CASES)
def test_output_files_match_the_reference_formats(cli, tmp_path, name, flags):  # noqa: F811
    fasta = H.golden_path("testFiles/" + name)              # .fa.gz is read through zlib
    base = str(tmp_path / "out")
    r = subprocess.run([cli, "-f", fasta, "--out-base", base] + shlex.split(flags), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    exp, stdout = expected_files(fasta, flags)
    assert r.stdout == stdout
    for sfx in H.BED_SUFFIXES:
        path = base + sfx
        if sfx in exp:
            with open(path) as fh:
                got = fh.read()
            assert got == exp[sfx], "%s differs" % sfx
        else:
            assert not os.path.exists(path), "%s must not be written with flags %r" % (sfx, flags)
    golden = H.golden_path("testFiles/expected/%s_gaps.bed" % name)
    if os.path.exists(golden):
        assert open(base + "_gaps.bed").read() == open(golden).read()


@pytest.mark.gpu
@pytest.mark.parametrize("name,flags", [("multi.fa", "-r -g -e -m -i"), ("multi_gap_t2t.fa", "-w 500 -s 250 -r -g -e -m -i"),
                                        ("bTaeGut7_chr33_mat.fa.gz", "-r -i"), ("multi.fa", ""), ("gapped_t2t.fa", "-t 300"),
                                        ("multi_gap_t2t.fa", "-p TTAGGG,TTAGG -r -g -i")])
def test_streaming_pipeline_equals_three_phase_path(cli, tmp_path, name, flags):  # noqa: F811
    """scanFastaToFiles (records flowing in groups through read / scan / write stages that overlap) against
    readFasta + walkPaths + writeBEDFiles: the same files and the same console text, byte for byte, for groups of
    one record, of a few and of the whole file."""
    fasta = H.golden_path("testFiles/" + name)
    ref_base = str(tmp_path / "ref")
    ref = subprocess.run([cli, "-f", fasta, "--out-base", ref_base, "--no-stream"] + shlex.split(flags), capture_output=True, timeout=300)
    assert ref.returncode == 0, ref.stderr
    # groups of one record, of a few, of the whole file; records as text pieces (the library strips the line ends while
    # staging; tiny pieces, so that every record and every N-cut segment is stitched from many) and joined on the host
    for group, extra in (("1", ["--piece-bytes", "61"]), ("5000", ["--piece-bytes", "1000"]), (str(1 << 30), []),
                         ("5000", ["--join-lines"]), (str(1 << 30), ["--join-lines", "--piece-bytes", "97"])):
        base = str(tmp_path / ("g" + group + "_".join(extra)))
        got = subprocess.run([cli, "-f", fasta, "--out-base", base, "--group-bytes", group] + extra + shlex.split(flags), capture_output=True, timeout=300)
        assert got.returncode == 0, got.stderr
        assert got.stdout == ref.stdout
        for sfx in H.BED_SUFFIXES:
            assert os.path.exists(base + sfx) == os.path.exists(ref_base + sfx), sfx
            if os.path.exists(base + sfx):
                assert open(base + sfx, "rb").read() == open(ref_base + sfx, "rb").read(), (sfx, group, extra)


GAP_BEDS = sorted(glob.glob(os.path.join(H.golden_path("testFiles/expected"), "*_gaps.bed")))


@pytest.mark.parametrize("bed", GAP_BEDS, ids=[os.path.basename(b) for b in GAP_BEDS])
def test_format_restatement_reproduces_the_golden_gap_beds(bed):
    """The formatter the GPU writers are checked against is itself pinned where the reference has files."""
    fasta = H.golden_path("testFiles/" + os.path.basename(bed)[:-len("_gaps.bed")])
    exp, _ = expected_files(fasta, "")
    assert exp["_gaps.bed"] == open(bed).read()
