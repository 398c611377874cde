"""Known-answer cases for ReadTelomereFilter::matches taken (as data) from the reference's
scripts/test_bam_subset.py:339-404 (default threshold, exact length / density boundaries,
custom canonical).  They pin the oracle's read-filter path; the same vectors are run against
the HIP path in tests/test_gpu_parity.py."""
import random

import pytest

from tests import harness as H
from tests.backends import OracleReadFilter

KATS = [
    # (cli options, {name: sequence}, names expected to pass)
    ("", {"short": "TTAGGG" * 6, "default_pass": "TTAGGG" * 7, "long": "CCCTAA" * 15,
          "fail": "ACGT" * 20}, ["default_pass", "long"]),
    ("-x 0 -l 12 -y 1 -k 10 -d 10",
     {"one_repeat": "TTAGGG", "exact_12": "TTAGGG" * 2,
      "flanked_exact": "ACGT" + "CCCTAA" * 2 + "TGCA", "exact_18": "TTAGGG" * 3},
     ["exact_12", "flanked_exact", "exact_18"]),
    ("-x 0 -l 18 -y 1 -k 10 -d 10",
     {"one_repeat": "TTAGGG", "exact_12": "TTAGGG" * 2,
      "flanked_exact": "ACGT" + "CCCTAA" * 2 + "TGCA", "exact_18": "TTAGGG" * 3},
     ["exact_18"]),
    ("-x 0 -l 18 -y 0.666 -k 20 -d 10", {"two_thirds": "TTAGGGAAAAAATTAGGG"}, ["two_thirds"]),
    ("-x 0 -l 18 -y 0.667 -k 20 -d 10", {"two_thirds": "TTAGGGAAAAAATTAGGG"}, []),
    ("-c CCCTAAA -x 0 -l 21 -y 1",
     {"plant_pass": "TTTAGGG" * 3, "vertebrate_fail": "TTAGGG" * 4}, ["plant_pass"]),
    ("-x 0 -l 18 -y 0.8 -k 10 -d 10", {"crlf": "TTAGGG" * 10 + "\r"}, ["crlf"]),
]


def random_read_set():
    """the 240-read generator of scripts/test_bam_subset.py:384-396 (seed 23)"""
    g = random.Random(23)
    seqs = {}
    for index in range(240):
        length = g.randrange(18, 250)
        s = "".join(g.choice("ACGTN") for _ in range(length))
        if index % 3 == 0:
            ins = g.randrange(len(s) + 1)
            rep = g.choice(("TTAGGG", "CCCTAA")) * g.randrange(2, 18)
            s = s[:ins] + rep + s[ins:]
        if index % 11 == 0:
            s += "TCAGGG" * 8 + "TTAGGG"
        seqs["random_%03d" % index] = s
    return seqs


RANDOM_OPTION_SETS = ["-x 0 -l 18 -y 0.8 -k 10 -d 10", "-x 1 -l 42 -y 0.5 -k 50 -d 50",
                      "-x 0 -l 60 -y 1 -k 10 -d 10"]


@pytest.mark.parametrize("case", range(len(KATS)))
def test_read_filter_kat(case):
    cli, seqs, expected = KATS[case]
    opts = H.parse_cli("--fastq-subset " + cli)
    rf = OracleReadFilter(opts)
    got = [n for n, ok in zip(seqs, rf.filter([s.encode() for s in seqs.values()])) if ok]
    assert got == expected
