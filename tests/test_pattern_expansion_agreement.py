"""Product (ts_expand_patterns, host code of libteloscan: no device needed) against the oracle's
expandPatternsWithOrientation (src/tools.cpp:201-283) over the flag sets the parity tests use: the same patterns in the
same order, the same canonical flags, and the same orientation for every entry the oracle does not mark ambiguous —
ambiguous entries (the same k-mer reachable from both strands, where the reference's result depends on std::sort's order
among equal keys) are the only ones a GPU parity test may take from the product."""
import pytest

from tests import harness as H

CLIS = ["", "-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG", "-c CCCTAAA", "-x 0", "-x 2", "-p TTAGGN -x 0", "-c TTAGG", "-c TTAG -x 1",
        "-c TTA -x 0", "-c TTTTAGGG -x 1", "-c TTTTTAGGG -x 1", "-p TTAGGG,TTAGG", "-c TTAGGGTTA -x 0", "-c AACCCTAACC -x 1",
        "-c TTAA -x 1", "-c ACGT -x 2", "-c GAATTC -x 1", "-c TTAGG -x 2", "-c AAAAAA -x 0", "-p TTAGGG,CCCTAA,TTAGGR -x 1"]


@pytest.mark.parametrize("cli", CLIS)
def test_product_and_oracle_expansions_agree_on_every_unambiguous_entry(cli):
    from oracle import pyoracle as po
    from teloscope_amd.cli import user_input
    opts = H.parse_cli("x.fa " + cli)
    ui = user_input(opts)
    ui._patterns()                                              # fills ui.patternInfo through ts_expand_patterns
    prod = [(p, f, p in (opts.canonical_fwd, opts.canonical_rev)) for p, f in ui.patternInfo]
    orac = po.expand_patterns(opts.raw_patterns, opts.edit_distance, opts.canonical_fwd)
    assert [p for p, _, _ in prod] == [p for p, _, _, _ in orac]
    n_amb = 0
    for (p, f, c), (_, of, oc, amb) in zip(prod, orac):
        assert c == oc, p
        if amb:
            n_amb += 1
        else:
            assert f == of, "orientation of %s: product %s, oracle %s" % (p, f, of)
    if cli in ("-c TTAG -x 1", "-c TTAA -x 1", "-c ACGT -x 2", "-c GAATTC -x 1"):
        assert n_amb > 0, "expected a self-complementary / doubly reachable pattern in this set"
