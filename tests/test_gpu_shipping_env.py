"""GPU parity UNDER THE RUNTIME ENVIRONMENT THE SAMPLER SHIPS WITH (AMD_DIRECT_DISPATCH=0: `flocoder_amd.runtime_defaults("sampling")`,
`bench.py`) -- the mode in which launches of different streams really overlap.  The HIP runtime reads the switch once,
when it is loaded, so each case is a fresh child process with the variable in its environment before anything imports torch; every case
runs ONCE per suite (nothing here loops or retries).

  (1) `sampling.sample_many(in_flight=2)` at the bench size -- B=64, 64 Euler steps, six calls, two replicas on two streams: every call
      bit-equal to the one-at-a-time result of the same plan, that result within the trajectory gate of the CPU oracle on four rows, every
      replica's device error word clean (tools/inflight_diag.py does the comparison and reports per call what differs);
  (2) the same under FLOCODER_AMD_POISON=1: every library buffer NaN-filled and fenced (csrc/devmem.hip), so a value read before it was
      written, or a write past a buffer's end, fails deterministically instead of depending on what the memory held;
  (3) the bench-size parity tests and the shared-device tests re-run inside such a process.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _child_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("FLOCODER_AMD_KEEP_ENV",)}
    env["AMD_DIRECT_DISPATCH"] = "0"
    env["FLOCODER_AMD_IN_CHILD_SUITE"] = "1"
    env.update(extra)
    return env


def _run_diag(env, rounds=1, timeout=420):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "inflight_diag.py"), "--rounds", str(rounds), "--calls", "6", "--in-flight", "2",
                        "--batch", "64", "--steps", "64"], env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    recs = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{")]
    return r, recs


def _check_diag(r, recs):
    assert recs, f"no report (rc {r.returncode}): {r.stderr[-1500:]}"
    head = recs[0]
    assert head["env"]["AMD_DIRECT_DISPATCH"] == "0"
    assert head["exclusive_repeat"]["equal"] and all(d["equal"] for d in head["shared_repeats"]), head
    assert head["shared_vs_exclusive_rel_l2"] < 1e-5, head
    assert head["oracle_rel_l2_rows0_3"]["exclusive"] < 2e-4 and head["oracle_rel_l2_rows0_3"]["shared"] < 2e-4, head   # 64 evaluations: the trajectory gate
    rounds = [x for x in recs if x.get("what", "").startswith("sample_many")]
    assert rounds, recs
    for rec in rounds:
        bad = [c for c in rec["calls"] if not (c["finite"] and c["equal"])]
        assert not bad, f"calls that differ from the one-at-a-time result: {bad}; error words {rec['replica_error_words']}"
        assert all(e == "ok" for e in rec["replica_error_words"]), rec["replica_error_words"]
        assert rec["poison"]["written_out_of_bounds"] == 0, rec["poison"]
    assert recs[-1] == {"what": "verdict", "failed": False} and r.returncode == 0, (recs[-1], r.returncode, r.stderr[-800:])


@pytest.mark.skipif(os.environ.get("FLOCODER_AMD_IN_CHILD_SUITE") == "1", reason="already inside the child suite")
def test_two_in_flight_at_bench_size_under_indirect_dispatch():
    r, recs = _run_diag(_child_env())
    _check_diag(r, recs)


@pytest.mark.skipif(os.environ.get("FLOCODER_AMD_IN_CHILD_SUITE") == "1", reason="already inside the child suite")
def test_two_in_flight_with_poisoned_and_fenced_buffers():
    r, recs = _run_diag(_child_env(FLOCODER_AMD_POISON="1"))
    _check_diag(r, recs)
    assert recs[0]["poison"]["buffers"] > 100       # the switch was really on: arenas, statistics, integrator state, parameter stores


@pytest.mark.skipif(os.environ.get("FLOCODER_AMD_IN_CHILD_SUITE") == "1", reason="already inside the child suite")
@pytest.mark.parametrize("in_flight,method,steps", [(1, "euler", 3), (2, "euler", 3), (2, "rk4", 30)])
def test_calls_with_different_inputs_do_not_see_each_others_conditioning(in_flight, method, steps):
    """Five calls with DIFFERENT noise and class ids, one at a time and two in flight: every output equals the one-at-a-time result of its own
    batch, and that result the CPU oracle's.  Found in round 4: under AMD_DIRECT_DISPATCH=0 a graph replay overtook the launches issued in
    front of it, and a call ran on the previous call's conditioning table (rel-L2 1.6e-2) -- invisible to every check that integrates the
    same samples in every call (tools/inflight_distinct.py reports which batch an output is closest to)."""
    # (the RK4 case: 29 intervals x 4 evaluations with guidance = two graph replays per call, so replay-behind-replay ordering is covered too;
    # three calls keep its CPU oracle leg short)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "inflight_distinct.py"), "--in-flight", str(in_flight), "--oracle", "--method", method,
                        "--steps", str(steps)] + (["--calls", "3", "--batch", "4"] if method == "rk4" else []), env=_child_env(OMP_NUM_THREADS="16"),
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    recs = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{")]
    assert recs and recs[-1] == {"what": "verdict", "failed": False} and r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-600:])
    head = recs[0]
    assert head["env"]["AMD_DIRECT_DISPATCH"] == "0"
    assert all(e < 2e-5 for e in head["exclusive_vs_oracle"]) and all(e < 1e-6 for e in head["shared_vs_exclusive"]), head
    for rec in recs[1:-1]:
        assert all(c["rel_to_own"] < 1e-6 and c["closest_batch"] == c["call"] and c["finite"] for c in rec["calls"]), rec
        assert all(e < 1e-6 for e in rec["exclusive_after_vs_before"]), rec


@pytest.mark.skipif(os.environ.get("FLOCODER_AMD_IN_CHILD_SUITE") == "1", reason="already inside the child suite")
def test_bench_size_and_shared_device_parity_under_indirect_dispatch():
    """tests/test_gpu_bench_sizes.py (the north-star gate, the default plan at B=64, RK4 + CFG) and tests/test_gpu_shared_device.py once more,
    in a process whose runtime was loaded with AMD_DIRECT_DISPATCH=0."""
    # (without test_b / test_d: their CPU oracle legs -- the SD-VAE at 256x256, 792 U-Net evaluations -- take minutes beside the runtime's
    # submission thread; test_d alone measured 220 s in such a process against a few seconds in the parent suite)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_bench_sizes.py", "tests/test_gpu_shared_device.py", "tests/test_gpu_unet.py",
                        "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", "not test_d_rk4 and not test_b_sdvae"],
                       env=_child_env(OMP_NUM_THREADS="16"), capture_output=True, text=True, timeout=900, cwd=ROOT)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0 and " passed" in tail, (r.returncode, r.stdout[-2500:], r.stderr[-800:])
