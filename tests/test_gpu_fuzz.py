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


def _with_env(name, fn):
    old = os.environ.get(name)
    os.environ[name] = "1"
    try:
        fn()
    finally:
        os.environ.pop(name, None)
        if old is not None:
            os.environ[name] = old


def test_seeded_fuzz_feature_mode_and_entry_points():
    """FUZZ_FEATURE: lambda on the feature-space Laplacian in most cases; FUZZ_EXTRAS: device matrices with a row stride,
    strided host items, save / load, threads."""
    import fuzz_parity as fz

    def run():
        rng = np.random.default_rng(20260103)
        for case in range(60):
            fz.one_case(np.random.default_rng(rng.integers(1 << 62)), case)
    _with_env("FUZZ_FEATURE", lambda: _with_env("FUZZ_EXTRAS", run))


def test_seeded_fuzz_boundary_sizes():
    import fuzz_parity as fz

    def run():
        rng = np.random.default_rng(20260104)
        for case in range(60):
            fz.one_case(np.random.default_rng(rng.integers(1 << 62)), case)
    _with_env("FUZZ_BOUNDARIES", run)


def test_seeded_fuzz_of_the_ring_build():
    """tools/fuzz_ring.py: plain and symmetric ring (forced column chunks) over random blocks against one space."""
    import fuzz_ring as fr
    rng = np.random.default_rng(20260105)
    for case in range(60):
        fr.one_case(np.random.default_rng(rng.integers(1 << 62)), case)


def test_seeded_fuzz_at_larger_sizes():
    """tools/fuzz_big.py: whole builds at 15k-60k items against the oracle, searches at 100k-400k against the CPU scorer."""
    import fuzz_big as fb
    rng = np.random.default_rng(20260106)
    for case in range(6):
        fb.one_case(np.random.default_rng(rng.integers(1 << 62)), case, "build")
    for case in range(8):
        fb.one_case(np.random.default_rng(rng.integers(1 << 62)), 100 + case, "search")


@pytest.mark.parametrize("world", [2, 3])
def test_seeded_fuzz_of_several_ranks_on_one_gpu(world):
    """tools/fuzz_2rank.py: ShardedIndex with 2 and 3 processes sharing the GPU (collectives through gloo), random cuts
    (some of a few rows), item and feature mode, save / load -- every rank against the oracle."""
    import subprocess
    env = dict(os.environ, FUZZ_FEATURE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_2rank.py"), "60" if world == 2 else "25", str(20260107 + world), str(world)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "failures in []" in r.stdout
