"""State hazards between the host mirror and the library's private copies (weights packed inside the handle, ONE activation arena).

  * weights written behind the (data_ptr, _version) key -- EMA.eval()/train() (train_flow.py:56-71), a weight broadcast after a first
    forward -- must reach the kernels;
  * a backward whose forward has been overwritten in the arena (second micro-batch, a sampler call in between) must still
    differentiate ITS forward;
  * class ids outside [0, n_classes) raise like nn.Embedding (unet.py:205), forward and backward treat them alike on the device.
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from conftest import rel_l2
from oracle import flow_oracle as fo

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(seed, **kw):
    from flocoder_amd.unet import Unet
    torch.manual_seed(seed)
    return Unet(dim=16, dim_mults=(1, 2, 4, 8), channels=4, **kw).to(DEV)


def _inputs(B=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, 4, 16, 16, generator=g).to(DEV), (torch.rand(B, generator=g) * 999).to(DEV),
            torch.randint(0, 10, (B,), generator=g).to(DEV))


def test_forward_after_ema_eval_runs_the_averaged_weights():
    from flocoder_amd.train import EMA
    m = _model(1, n_classes=10).train()
    x, t, ids = _inputs()
    ema = EMA(m, decay=0.5, device=DEV)
    with torch.no_grad():
        live0 = m(x, t, {"class_cond": ids}).clone()               # the library now holds the live weights
        for p in m.parameters():                                   # "a few optimiser steps"
            p.add_(0.05 * torch.randn_like(p))
        ema.update()
        live1 = m(x, t, {"class_cond": ids}).clone()
        ema.eval()
        got = m(x, t, {"class_cond": ids}).clone()
    fresh = _model(99, n_classes=10)
    fresh.load_state_dict({k: v.clone() for k, v in ema.shadow.items()}, strict=True)
    with torch.no_grad():
        want = fresh(x, t, {"class_cond": ids})
    assert rel_l2(got, want) < 1e-6, "forward after ema.eval() must use the EMA weights"
    assert rel_l2(got, live1) > 1e-3 and rel_l2(live1, live0) > 1e-3
    ema.train()
    with torch.no_grad():
        back = m(x, t, {"class_cond": ids})
    assert torch.equal(back, live1), "ema.train() must restore the live weights inside the library too"


def test_data_copy_needs_mark_dirty_and_gets_it_from_the_codecs_too():
    from flocoder_amd.codecs import SD_VAE_Wrapper
    m = _model(2, n_classes=0).eval()
    x, t, _ = _inputs()
    with torch.no_grad():
        a = m(x, t).clone()
        for p in m.parameters():
            p.data.copy_(p.data * 1.1)                             # invisible to (data_ptr, _version)
        m.mark_dirty()
        b = m(x, t)
    assert rel_l2(b, a) > 1e-3
    w = SD_VAE_Wrapper(weights="random", seed=3).eval().to(DEV)
    z = torch.randn(1, 4, 8, 8, device=DEV)
    y0 = w.decode(z).clone()
    for p in w.parameters():
        p.data.mul_(1.05)
    w.mark_dirty()
    assert rel_l2(w.decode(z), y0) > 1e-3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bcast_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from flocoder_amd import dist as fdist
    from flocoder_amd.unet import Unet
    fdist.init(backend="gloo")                                     # both ranks share cuda:0 here; RCCL differs only in transport
    torch.manual_seed(100 + rank)                                  # different weights per rank ...
    m = Unet(dim=8, channels=4, n_classes=3).eval().to("cuda:0")
    g = torch.Generator().manual_seed(5)
    x, t = torch.randn(2, 4, 8, 8, generator=g).to("cuda:0"), (torch.rand(2, generator=g) * 999).to("cuda:0")
    ids = torch.tensor([0, 2], device="cuda:0")
    with torch.no_grad():
        before = m(x, t, {"class_cond": ids}).cpu()                # ... already packed inside each rank's library handle
        fdist.broadcast_weights(m, src=0)
        after = m(x, t, {"class_cond": ids}).cpu()
    q.put((rank, before, after))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_broadcast_weights_after_a_first_forward_reaches_the_kernels():
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=500) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, b0, a0), (_, b1, a1) = res
    assert not torch.equal(b0, b1)                                 # they started apart
    assert torch.equal(a0, b0)                                     # rank 0 unchanged
    assert torch.equal(a1, a0), "rank 1 must compute with rank 0's weights after the broadcast"


def test_backward_of_an_overwritten_forward_still_gets_its_own_gradients():
    m = _model(3, n_classes=10).train()
    xa, ta, ia = _inputs(3, seed=1)
    xb, tb, ib = _inputs(3, seed=2)
    ga = torch.randn(3, 4, 16, 16, device=DEV)

    def grads_of(fn):
        m.zero_grad(set_to_none=True)
        fn()
        return torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None]).clone()

    def clean():
        (m(xa, ta, {"class_cond": ia}) * ga).sum().backward()

    def interleaved():
        va = m(xa, ta, {"class_cond": ia})                         # forward A
        with torch.no_grad():
            m(xb, tb, {"class_cond": ib})                          # forward B overwrites the arena
        m.eval()
        from flocoder_amd.sampling import euler_sampler
        euler_sampler(m, (3, 4, 16, 16), 2, cond=ib, source=xb)    # and so does a sampler call
        m.train()
        (va * ga).sum().backward()                                 # backward A

    want, got = grads_of(clean), grads_of(interleaved)
    assert rel_l2(got, want) < 1e-6
    # two micro-batches summed into one loss (gradient accumulation shape)
    def two():
        la = (m(xa, ta, {"class_cond": ia}) * ga).sum()
        lb = (m(xb, tb, {"class_cond": ib}) * ga).sum()
        (la + lb).backward()
    def b_only():
        (m(xb, tb, {"class_cond": ib}) * ga).sum().backward()
    both = grads_of(two)
    assert rel_l2(both, want + grads_of(b_only)) < 1e-5


def test_c_abi_refuses_a_backward_without_its_forward():
    from flocoder_amd import _binding as B
    m = _model(4, n_classes=10).train()
    x, t, ids = _inputs(2, seed=3)
    with torch.no_grad():
        m._forward_native(x, t, ids, None, train=True)
        m.integrate("euler", x.clone(), torch.tensor([0.1, 0.5]), dt_euler=0.5, class_ids=ids)
    with pytest.raises(RuntimeError, match="activation arena"):
        m.backward_native(x, t, ids, torch.ones_like(x))


def test_class_ids_out_of_range_raise_index_error():
    m = _model(5, n_classes=10).eval()
    x, t, _ = _inputs(2)
    for bad in ([0, 10], [-1, 3]):
        with pytest.raises(IndexError):
            m(x, t, {"class_cond": torch.tensor(bad, device=DEV)})
        with pytest.raises(IndexError):
            m.integrate("euler", x.clone(), torch.tensor([0.1]), dt_euler=1.0, class_ids=torch.tensor(bad, device=DEV))
    with pytest.raises(ValueError):
        m(x, t, {"class_cond": torch.tensor([1, 2, 3], device=DEV)})


def test_ot_pairing_stays_a_permutation_with_non_finite_rows():
    from flocoder_amd._ops import ot_pairing
    g = torch.Generator().manual_seed(7)
    s, t = torch.randn(70, 32, generator=g), torch.randn(70, 32, generator=g)
    s[5] = float("nan")
    s[40] = float("inf")
    perm, _ = ot_pairing(s.to(DEV), t.to(DEV))
    assert sorted(perm.cpu().tolist()) == list(range(70))
    # rows without NaN/inf are still the greedy first-minimum choice of the reference among what is left (ot.py:68-77)
    d = torch.cdist(s, t)
    used = torch.zeros(70, dtype=torch.bool)
    for i in range(70):
        row = d[i].clone()
        row[used] = float("inf")
        j = int(perm[i])
        if torch.isfinite(d[i]).all():
            assert j == int(torch.argmin(row)), i
        used[j] = True


def test_step_prologue_matches_torch_and_flags_bad_class_ids():
    """fc_flow_prepare = the torch prologue of train_flow.py:346-357 (warp_time with torch's own rounding, interpolation, pairing
    gather); out-of-range class ids surface as IndexError at the trainer's next check."""
    from flocoder_amd.sampling import warp_time
    from flocoder_amd.train import FlowTrainer
    m = _model(5, n_classes=10).train()
    tr = FlowTrainer(m)
    g = torch.Generator().manual_seed(3)
    src, tgt = torch.randn(6, 4, 16, 16, generator=g).to(DEV), torch.randn(6, 4, 16, 16, generator=g).to(DEV)
    u = torch.rand(6, generator=g).to(DEV)
    perm = torch.randperm(6, generator=g).to(DEV)
    t, time, x, v = tr.prepare(src, tgt, u, None, perm)
    t_ref = warp_time(u * (1 - tr.t_eps) + tr.t_eps)
    assert torch.equal(t, t_ref) and torch.equal(time, t_ref * tr.t_scale)
    x_ref, v_ref = tr.interpolate(src, tgt[perm].contiguous(), t_ref)
    assert torch.equal(x, x_ref) and torch.equal(v, v_ref)
    t2, _, x2, _ = tr.prepare(src, tgt, u)                       # no pairing
    assert torch.equal(t2, t_ref) and torch.equal(x2, tr.interpolate(src, tgt, t_ref)[0])
    tr.step(src, tgt, {"class_cond": torch.tensor([0, 1, 2, 3, 4, 9], device=DEV)})
    tr.check_class_ids()                                          # in range: nothing raised
    with pytest.raises(IndexError):
        tr.step(src, tgt, {"class_cond": torch.tensor([0, 1, 2, 3, 4, 10], device=DEV)})
        tr.check_class_ids()
    tr.check_class_ids()                                          # the flag was cleared by the raise


def test_step_with_pairing_equals_step_on_gathered_targets():
    """FlowTrainer.step(pairing=perm) trains against target[perm] without materialising it: same loss, same parameters."""
    from flocoder_amd.train import FlowTrainer
    g = torch.Generator().manual_seed(11)
    src, tgt = torch.randn(6, 4, 16, 16, generator=g).to(DEV), torch.randn(6, 4, 16, 16, generator=g).to(DEV)
    u = torch.rand(6, generator=g).to(DEV)
    perm = torch.randperm(6, generator=g).to(DEV)
    ids = torch.tensor([0, 1, 2, 3, 4, 9], device=DEV)
    out = []
    for use_pairing in (True, False):
        m = _model(5, n_classes=10).train()
        tr = FlowTrainer(m)
        if use_pairing:
            loss = tr.step(src, tgt, {"class_cond": ids}, u=u, pairing=perm)
        else:
            loss = tr.step(src, tgt[perm].contiguous(), {"class_cond": ids}, u=u)
        out.append((float(loss), tr.params.clone()))
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1])


def test_backward_of_a_batch_smaller_than_the_plan_matches_a_plan_of_that_size():
    """A plan reserved for B rows serves smaller batches (the last batch of an epoch): the backward then launches every weight
    gradient on its own instead of through the table launches of the full-batch path.  Same gradients either way."""
    from flocoder_amd.train import FlowTrainer
    g = torch.Generator().manual_seed(21)
    src, tgt = torch.randn(6, 4, 16, 16, generator=g).to(DEV), torch.randn(6, 4, 16, 16, generator=g).to(DEV)
    t = (torch.rand(6, generator=g) * 0.9 + 0.05).to(DEV)
    ids = torch.randint(0, 10, (6,), generator=g).to(DEV)
    big = FlowTrainer(_model(7, n_classes=10).train())
    x6, v6 = big.interpolate(src, tgt, t)
    big.loss_and_grads(x6, t, ids, v6)                                   # plan (and backward plan) for 6 rows
    x4, v4 = big.interpolate(src[:4].contiguous(), tgt[:4].contiguous(), t[:4].contiguous())
    loss_a, _ = big.loss_and_grads(x4, t[:4].contiguous(), ids[:4].contiguous(), v4)
    ga = big.grads.clone()
    small = FlowTrainer(_model(7, n_classes=10).train())                 # same weights, plan for 4 rows: the table path
    loss_b, _ = small.loss_and_grads(x4, t[:4].contiguous(), ids[:4].contiguous(), v4)
    gb = small.grads
    assert abs(float(loss_a) - float(loss_b)) <= 1e-6 * abs(float(loss_b))
    assert float((ga - gb).norm() / gb.norm()) < 2e-6
