import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """npz -> dict; 0-d string arrays are JSON payloads (shape tables)."""
    out = {}
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        for k in z.files:
            v = z[k]
            out[k] = json.loads(str(v)) if v.dtype.kind == "U" and v.ndim == 0 else v
    return out


def rel_l2(a, b):
    import torch
    a = torch.as_tensor(a, dtype=torch.float64).detach().flatten()
    b = torch.as_tensor(b, dtype=torch.float64).detach().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(autouse=True, scope="module")
def _poison_fences():
    """Under FLOCODER_AMD_POISON=1 (csrc/devmem.hip: every library buffer NaN-filled and fenced) the whole GPU suite doubles as a memory
    check: a value read before it was written shows up as NaN in the test that reads it, and after every module the fences of all live
    buffers are inspected for writes past a buffer's ends."""
    yield
    if os.environ.get("FLOCODER_AMD_POISON", "0") in ("", "0"):
        return
    import torch
    if torch.cuda.device_count() == 0:
        return
    import ctypes as C
    from flocoder_amd import _binding as B
    bad, live = C.c_int(0), C.c_int(0)
    B.check(B.lib().fc_debug_poison_check(C.byref(bad), C.byref(live)))
    assert bad.value == 0, B.lib().fc_last_error().decode(errors="replace")
