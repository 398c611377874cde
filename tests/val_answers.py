"""The reference CI's known answers (tests/golden/val_known_answers.tsv, from .github/workflows/val.sh:107-196) as
(args, must_contain, substring) triples, and the check itself, shared by the oracle test and the GPU test."""
import os

from tests import harness as H

TABLE = os.path.join(H.GOLDEN, "val_known_answers.tsv")


def cases():
    out = []
    for line in open(TABLE):
        line = line.rstrip("\n")
        if not line or line.startswith("#"):
            continue
        args, sign, sub = line.split("\t")
        out.append((args, sign == "+", sub.replace("\\t", "\t")))
    return out


def check(backend_factory, args, must_contain, sub):
    """Runs `teloscope ARGS` through tests/harness.py's restatement of the CLI on the given backend and checks its stdout."""
    opts = H.parse_cli("teloscope " + args)
    stdout, _ = H.run_assembly(backend_factory(opts), opts, H.golden_path(opts.input))
    assert (sub in stdout) == must_contain, (args, sub, stdout)
