"""CPU-only checks of the boundary: the shared library loads, exports every symbol include/flocoder_amd.h declares,
the ctypes signatures cover exactly that set, the parameter table equals the reference's state_dict, and the host
mirror fails loudly instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden

HEADER = os.path.join(ROOT, "include", "flocoder_amd.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from flocoder_amd import _binding as B
    lib = B.lib()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/flocoder_amd.h but not exported"
    assert sorted(B.SIGNATURES) == syms, "ctypes SIGNATURES and the header disagree"
    assert lib.fc_abi_version() == 1


def test_no_gpu_means_loud_failure():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from flocoder_amd import _binding as B
    from flocoder_amd.unet import Unet
    assert B.lib().fc_check_device(0) == B.FC_E_HIP
    with pytest.raises(RuntimeError):
        B.check(B.lib().fc_check_device(0))
    m = Unet(dim=8, channels=4, n_classes=0).eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 4, 8, 8), torch.zeros(1))
    with pytest.raises(RuntimeError):
        from flocoder_amd.sampling import euler_sampler
        euler_sampler(m, (1, 4, 8, 8), 2, source=torch.zeros(1, 4, 8, 8))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "flocoder_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{f} imports the oracle"
                assert "/root/reference" not in text, f"{f} reads the reference at run time"


@pytest.mark.parametrize("tag,kw", [("d16c10", dict(dim=16, channels=4, n_classes=10)),
                                    ("d8mask", dict(dim=8, channels=4, n_classes=0, mask_cond=True)),
                                    ("d32c102", dict(dim=32, channels=4, n_classes=102))])
def test_state_dict_layout_and_default_init_match_reference(tag, kw):
    """Same keys, same order, same shapes, and -- under the same seed -- the same values as the reference's Unet."""
    from flocoder_amd.unet import Unet
    g = load_golden("g0_state_dict")["layout"][tag]
    torch.manual_seed(0)
    sd = Unet(dim_mults=(1, 2, 4, 8), **kw).state_dict()
    assert list(sd.keys()) == g["keys"]
    assert [list(v.shape) for v in sd.values()] == g["shapes"]
    s = np.array([float(v.double().sum()) for v in sd.values()])
    a = np.array([float(v.double().abs().sum()) for v in sd.values()])
    assert np.allclose(s, g["sum"], rtol=0, atol=1e-9) and np.allclose(a, g["abssum"], rtol=1e-12)


def test_param_table_needs_no_gpu_and_is_16_byte_aligned():
    from flocoder_amd.unet import _make_config, param_table
    tab = param_table(_make_config(32, (1, 2, 4, 8), 4, 4, 102, False))
    assert len(tab) == 298 and sum(int(np.prod(s)) for _, s, _ in tab) == 9_919_044     # SURVEY 6 probe
    assert all(off % 4 == 0 for _, _, off in tab)
    names = [n for n, _, _ in tab]
    assert "downs.2.3.1.weight" in names and "ups.3.3.weight" in names and "mid_attn.fn.fn.to_out.bias" in names


def test_time_grids_match_reference_bitwise():
    from flocoder_amd import sampling as S
    g = load_golden("g4_timegrids")
    for n in (3, 5, 16, 64, 100):
        assert np.array_equal(S.rk4_time_grid(n).numpy(), g[f"rk4_{n}"])
    assert np.array_equal(S.warp_time(torch.from_numpy(g["rand_in"])).numpy(), g["rand_out"])
    with pytest.raises(ValueError):
        S.warp_time(torch.zeros(1), s=-0.1)
    e = S.euler_time_grid(64)
    assert e.dtype == torch.float32 and abs(float(e[0]) - 1e-3) < 1e-9 and float(e[-1]) < 1.0


def test_bench_starts_its_own_ranks_and_refuses_without_gpus():
    """`python bench.py --gpus N` called plainly must not die with "launch with torch.distributed.run": the parent spawns the ranks
    itself (bench.launch_ranks) and never touches the GPU.  On a box with fewer GPUs than N it says so and exits 2."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("multi-GPU node: the real launch is the driver's scaling run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FLOCODER_AMD_SINGLE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 2 but this node shows" in r.stderr, (r.returncode, r.stderr[-500:])
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "torch.distributed.run" in src and "launch_ranks" in src
