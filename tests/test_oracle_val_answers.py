"""The oracle against the known answers the reference's own CI script checks (val.sh:107-196): twelve scaffold
classifications, five flag behaviours (-l, -t, -y twice, -k), the summary's spelling and the multi-contig counts —
reference-held pins on block calling and labelling beyond the .tst manifests."""
import pytest

from tests import val_answers as V
from tests.backends import OracleBackend

CASES = V.cases()


def test_table_is_complete():
    assert len(CASES) == 28


@pytest.mark.parametrize("args,must,sub", CASES, ids=["%s | %s" % (a, s.replace("\t", " ")) for a, _, s in CASES])
def test_known_answer(args, must, sub):
    V.check(OracleBackend, args, must, sub)
