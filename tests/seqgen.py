"""Seeded synthetic sequence generators shared by the parity tests, bench.py and smoke()."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_dna(rng, n, gc=0.5):
    p = [(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2]
    return ACGT[rng.choice(4, size=n, p=p)]


def mutate(rng, arr, rate):
    if rate <= 0 or len(arr) == 0:
        return arr
    arr = arr.copy()
    idx = np.flatnonzero(rng.random(len(arr)) < rate)
    arr[idx] = ACGT[rng.integers(0, 4, size=len(idx))]
    return arr


def repeat_array(unit, n):
    return np.tile(np.frombuffer(unit.encode(), dtype=np.uint8), n)


def chromosome(rng, n, unit_fwd="CCCTAA", unit_rev="TTAGGG", telo_repeats=200, tvr_rate=0.02,
               n_its=2, iupac=0, lower=0.0, n_runs=0):
    """Telomere + TVR at both ends (model of src/get-mock-chr.cpp:96-136), random core,
    optional interstitial telomeric blocks, IUPAC codes, soft-masking and N runs."""
    core = random_dna(rng, n)
    p = mutate(rng, repeat_array(unit_fwd, telo_repeats), tvr_rate)
    q = mutate(rng, repeat_array(unit_rev, telo_repeats), tvr_rate)
    if len(p) + len(q) < n:
        core[:len(p)] = p
        core[n - len(q):] = q
    for _ in range(n_its):
        ln = int(rng.integers(4, 60)) * len(unit_fwd)
        if n > 4 * ln + len(p) + len(q):
            at = int(rng.integers(len(p) + ln, n - len(q) - 2 * ln))
            unit = unit_fwd if rng.random() < 0.5 else unit_rev
            core[at:at + ln] = mutate(rng, repeat_array(unit, ln // len(unit)), 0.03)
    if iupac:
        idx = rng.integers(0, n, size=iupac)
        core[idx] = np.frombuffer(b"RYKMSWBDHV", dtype=np.uint8)[rng.integers(0, 10, size=iupac)]
    if lower > 0:
        m = rng.random(n) < lower
        core[m] |= 0x20
    for _ in range(n_runs):
        ln = int(rng.integers(1, 200))
        at = int(rng.integers(0, max(1, n - ln)))
        core[at:at + ln] = ord("N")
    return core.tobytes()
