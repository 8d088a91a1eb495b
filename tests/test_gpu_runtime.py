"""GPU: process-level behaviour -- the library and PyTorch-ROCm share ONE HIP runtime whichever is imported first
(torch wheels bundle their own libamdhip64; a second runtime in the process finds no GPU)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, {root!r})
import numpy as np
order = sys.argv[1]
if order == "torch_first":
    import torch
    torch.zeros(1, device="cuda")
import pyarrowspace_amd as asp
X = np.random.default_rng(0).standard_normal((300, 32))
X /= np.linalg.norm(X, axis=1, keepdims=True)
aspace, gl = asp.ArrowSpaceBuilder.build({{"eps": 1.2, "k": 5, "topk": 3, "p": 2.0, "sigma": None}}, X)
assert aspace.search(X[0], gl, 1.0)[0][0] == 0
import torch
assert float(torch.ones(4, device="cuda").sum()) == 4.0
n = sum(1 for l in set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l))
assert n == 1, n
print("ok")
"""


@pytest.mark.parametrize("order", ["library_first", "torch_first"])
def test_one_hip_runtime_per_process(order):
    out = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT), order], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
