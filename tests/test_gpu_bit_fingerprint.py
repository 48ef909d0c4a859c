"""Bit fingerprints of the optimised bands (tests/fingerprint_cases.py): every scene must reproduce the hash committed in
tests/golden/bit_fingerprints.json - pose counts, states, cost and chi^2 of all bands, bit for bit. The launch is deterministic
(fixed summation orders, no atomics), so a changed hash means a changed result: a kernel change meant as a pure speed-up has to keep
them; one that legitimately changes rounding has to regenerate the file (tools/bit_fingerprint.py) and say so in its commit."""
import json
import os

import pytest

import fingerprint_cases as F

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bit_fingerprints.json")


@pytest.mark.parametrize("name", list(F.CASES))
def test_fingerprint_is_reproduced(name):
    want = json.load(open(GOLDEN))
    got, ms = F.fingerprint(name)
    again, _ = F.fingerprint(name)
    assert got == again, "two launches of the same scene differ: the kernel is not deterministic"
    assert got == want[name], (name, got, want[name])
