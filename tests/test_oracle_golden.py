"""Oracle (oracle/ngp_oracle.c) against the committed fixture tests/golden/golden_v1.npz, which was minted from the
reference's own kernel headers (tests/golden/make_golden.py).  Needs neither /root/reference nor oracle/_ref."""
import numpy as np
import pytest
from oracle import oracle as O
import golden_cases as GC


@pytest.mark.parametrize("case", GC.CASES)
def test_oracle_matches_golden(case):
    getattr(GC, case)(O, GC.load(), exact=True)
