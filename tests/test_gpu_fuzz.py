"""GPU: a short seeded run of tools/fuzz_parity.py (random shapes, graph parameters, metric / kernel, query kinds and
tau against the oracle).  The long runs that found the cosine clamp and the energy noise floor
are `python tools/fuzz_parity.py 200 <seed>`."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_seeded_fuzz_against_the_oracle():
    import fuzz_parity as fz
    rng = np.random.default_rng(20260101)
    for case in range(40):
        fz.one_case(np.random.default_rng(rng.integers(1 << 62)), case)


def test_seeded_fuzz_of_the_staged_path():
    import fuzz_parity as fz
    rng = np.random.default_rng(20260102)
    for case in range(40):
        fz.one_case(np.random.default_rng(rng.integers(1 << 62)), case, sharded=True)
