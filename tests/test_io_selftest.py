"""Host-only checks of the C++ mirror's ingest (include/teloscope_mi355x_io.hpp): the mapped multi-threaded FASTA
reader against the zlib stream reader, the streaming group reader against both, splitPath's word-at-a-time scan against a per-character walk.  No GPU."""
import os
import subprocess

import __graft_entry__ as entry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fasta_reader_and_path_splitter(tmp_path):
    entry.build()                                            # libteloscan.so (the header's inline calls link to it)
    libdir = os.path.join(ROOT, "teloscope_amd")
    exe = str(tmp_path / "io_selftest")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "io_selftest.cpp"), "-L", libdir, "-lteloscan",
                           "-Wl,-rpath," + libdir, "-pthread", "-lz", "-o", exe])
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    r = subprocess.run([exe, str(scratch)], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert b"io_selftest ok" in r.stdout
