"""A process that loads the product BEFORE PyTorch must still find its GPU in torch afterwards: torch bundles its own
HIP / HSA runtime copies and the library links the system ones (pbsim3_amd._torch_runtime_first)."""
import os
import subprocess
import sys

import pytest

import harness

pytestmark = pytest.mark.gpu

SCRIPT = """
import pbsim3_amd as P
with P.Context(P.default_params(), 0) as c:
    z = c.deflate_buffer(b"ACGT" * 1000)
import torch
assert torch.cuda.is_available()
print("ok", len(z), int(torch.ones(5, device="cuda").sum().item()))
"""


def test_product_first_then_torch():
    p = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, cwd=harness.ROOT,
                       env=dict(os.environ, PYTHONPATH=harness.ROOT))
    assert p.returncode == 0, p.stderr[-1500:]
    assert p.stdout.split()[0] == "ok" and p.stdout.split()[2] == "5"
