"""The exclusive plan's residency assumption in the PRODUCT path (round-2 verdict, item 1).

The default plan closes most Blocks inside their second convolution: the workgroups of a sample exchange GroupNorm partials inside the
launch and wait for each other (conv_dev.h).  That is only correct while the whole grid is resident, so

  * two replicas on two streams must give oracle-equal results -- through the plan without cross-workgroup waits (what a non-default
    stream or ``set_shared_device(True)`` selects) or, when both insist on the exclusive plan, because the library orders the two
    plans one behind the other (meeting guard, unet.hip);
  * a wait that times out anyway must be LOUD: NaN samples, an exception from the call that waited (``Unet.integrate``), from
    ``check_errors`` and from every later call on the model -- never finite garbage.  ``fc_debug_unet_break_meeting`` provokes it.

The U-Net rows a Block must still equal: /root/reference/flocoder/unet.py:57-96.
"""
import pytest
import torch

from conftest import rel_l2
from oracle import flow_oracle as fo

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, STEPS = 8, 3


def _model(seed=0):
    from flocoder_amd.unet import Unet
    torch.manual_seed(seed)
    return Unet(dim=32, dim_mults=(1, 2, 4, 8), channels=4, n_classes=102).eval().to(DEV)


def _inputs():
    g = torch.Generator().manual_seed(11)
    return torch.randn(B, 4, 32, 32, generator=g), torch.randint(102, (B,), generator=g)


def _two_streams(models, x, ids):
    from flocoder_amd.sampling import euler_sampler
    streams = [torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)]
    cur = torch.cuda.current_stream(DEV)
    outs = []
    for st in streams:
        st.wait_stream(cur)
    for rep in range(2):                       # two trajectories per replica, interleaved: both streams have work in flight
        for m, st in zip(models, streams):
            with torch.cuda.stream(st):
                outs.append(euler_sampler(m, (B, 4, 32, 32), STEPS, cond=ids.to(DEV), source=x.to(DEV))[0])
    for st in streams:
        cur.wait_stream(st)
    torch.cuda.synchronize()
    return outs


@pytest.fixture(scope="module")
def reference():
    m = _model()
    x, ids = _inputs()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    want = fo.euler_sampler(sd, x, STEPS, ids)[0]
    return sd, x, ids, want


def test_default_stream_runs_the_exclusive_plan_and_matches_the_oracle(reference):
    from flocoder_amd.sampling import euler_sampler
    sd, x, ids, want = reference
    m = _model()
    got = euler_sampler(m, (B, 4, 32, 32), STEPS, cond=ids.to(DEV), source=x.to(DEV))[0]
    assert m.meeting_launches > 0, "the default plan on an exclusive device closes Blocks across workgroups"
    assert rel_l2(got.cpu(), want) < 1e-4
    m.check_errors()
    assert m.fused_tail_errors() == 0


def test_two_replicas_on_two_streams_take_the_shared_plan(reference):
    sd, x, ids, want = reference
    a, b = _model(), _model()
    outs = _two_streams([a, b], x, ids)
    assert a.meeting_launches == 0 and b.meeting_launches == 0, "a caller on a non-default stream gets the plan without cross-workgroup waits"
    for o in outs:
        assert torch.isfinite(o).all() and rel_l2(o.cpu(), want) < 1e-4
    assert all(torch.equal(o, outs[0]) for o in outs)
    a.check_errors(); b.check_errors()
    # ... and the two plans agree to fp32 rounding (the fused tail combines the same GroupNorm partials, in a different order)
    from flocoder_amd.sampling import euler_sampler
    c = _model()
    excl = euler_sampler(c, (B, 4, 32, 32), STEPS, cond=ids.to(DEV), source=x.to(DEV))[0]
    assert c.meeting_launches > 0 and rel_l2(excl, outs[0]) < 1e-6


def test_two_exclusive_replicas_on_two_streams_are_ordered_by_the_library(reference):
    sd, x, ids, want = reference
    a, b = _model(), _model()
    a.set_shared_device(False); b.set_shared_device(False)
    outs = _two_streams([a, b], x, ids)
    assert a.meeting_launches > 0 and b.meeting_launches > 0
    for o in outs:
        assert torch.isfinite(o).all() and rel_l2(o.cpu(), want) < 1e-4
    a.check_errors(); b.check_errors()
    assert a.fused_tail_errors() == 0 and b.fused_tail_errors() == 0
    # plain forwards of both replicas on the two streams, back to back, many times: still ordered, still right
    t = torch.full((B,), 300.0, device=DEV)
    streams = [torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)]
    xd, idd = x.to(DEV), ids.to(DEV)
    torch.cuda.synchronize()
    vs = []
    with torch.no_grad():
        for i in range(6):
            with torch.cuda.stream(streams[i & 1]):
                vs.append((a, b)[i & 1](xd, t, {"class_cond": idd}))
    torch.cuda.synchronize()
    ref_v = fo.unet_forward(sd, x, t.cpu(), {"class_cond": ids})
    for v in vs:
        assert rel_l2(v.cpu(), ref_v) < 2e-5
    a.check_errors(); b.check_errors()


def test_a_timed_out_meeting_is_loud_not_garbage(reference):
    from flocoder_amd import _binding as Bn
    from flocoder_amd.sampling import euler_sampler
    sd, x, ids, want = reference
    m = _model()
    t = torch.full((B,), 300.0, device=DEV)
    with torch.no_grad():
        good = m(x.to(DEV), t, {"class_cond": ids.to(DEV)}).clone()
    assert m.meeting_launches > 0
    Bn.check(Bn.lib().fc_debug_unet_break_meeting(m._handle))       # one launch's workgroups now disagree about the epoch
    with torch.no_grad():
        bad = m(x.to(DEV), t, {"class_cond": ids.to(DEV)})
    torch.cuda.synchronize()
    assert not torch.isfinite(bad).all(), "a timed-out wait must poison its samples"
    assert torch.isnan(bad).any()
    with pytest.raises(RuntimeError, match="timed out"):
        m.check_errors()
    with pytest.raises(RuntimeError, match="timed out"):            # sticky: every later call refuses
        with torch.no_grad():
            m(x.to(DEV), t, {"class_cond": ids.to(DEV)})
    with pytest.raises(RuntimeError, match="timed out"):
        euler_sampler(m, (B, 4, 32, 32), STEPS, cond=ids.to(DEV), source=x.to(DEV))
    assert m.fused_tail_errors() == 1
    # the remedy the message names: rebuild the plan without cross-workgroup waits
    m.set_shared_device(True)
    with torch.no_grad():
        again = m(x.to(DEV), t, {"class_cond": ids.to(DEV)})
    assert m.meeting_launches == 0 and rel_l2(again, good) < 1e-6
    m.check_errors()


def test_linear_attention_closes_itself_on_the_exclusive_plan_and_its_wait_is_bounded(reference):
    """linattn_fused.hip: at n >= 256 the apply launch of a LinearAttention module normalises (to_out.1) and adds x itself on the exclusive
    plan -- the tiles of a sample exchange their statistics and wait for each other -- while the shared plan keeps the separate finalize
    launch; same bits either way (the same arithmetic in the same order).  A wait that cannot end is bounded and loud, like the Block tails'."""
    from flocoder_amd import _binding as Bn
    sd, x, ids, want = reference
    m = _model()
    t = torch.full((B,), 300.0, device=DEV)
    with torch.no_grad():
        excl = m(x.to(DEV), t, {"class_cond": ids.to(DEV)}).clone()
    kernels = [r["kernel"] for r in m.profile_ops(B, repeats=1)]
    assert kernels.count("linattn_fused+fin") == 4 and "linattn_fused" not in kernels, kernels
    m.set_shared_device(True)
    with torch.no_grad():
        shared = m(x.to(DEV), t, {"class_cond": ids.to(DEV)}).clone()
    kernels = [r["kernel"] for r in m.profile_ops(B, repeats=1)]
    assert kernels.count("linattn_fused") == 4 and "linattn_fused+fin" not in kernels, kernels
    assert rel_l2(shared, excl) < 1e-6
    m.set_shared_device(None)
    with torch.no_grad():
        assert torch.equal(m(x.to(DEV), t, {"class_cond": ids.to(DEV)}), excl)
    Bn.check(Bn.lib().fc_debug_unet_break_meeting_kind(m._handle, 1))       # the tiles of sample 0 now disagree about the epoch
    with torch.no_grad():
        bad = m(x.to(DEV), t, {"class_cond": ids.to(DEV)})
    torch.cuda.synchronize()
    assert torch.isnan(bad[0]).any(), "a timed-out wait must poison its sample"
    with pytest.raises(RuntimeError, match="timed out"):
        m.check_errors()


def test_the_sampler_raises_inside_the_call_that_waited(reference):
    """Unet.integrate(check=True) (the default behind every sampler entry point) waits for the trajectory and raises itself."""
    from flocoder_amd import _binding as Bn
    from flocoder_amd.sampling import euler_sampler
    sd, x, ids, want = reference
    m = _model()
    euler_sampler(m, (B, 4, 32, 32), 1, cond=ids.to(DEV), source=x.to(DEV))          # builds the plan
    Bn.check(Bn.lib().fc_debug_unet_break_meeting(m._handle))
    with pytest.raises(RuntimeError, match="timed out"):
        euler_sampler(m, (B, 4, 32, 32), 1, cond=ids.to(DEV), source=x.to(DEV))


def test_sample_many_equals_one_at_a_time(reference):
    """sampling.sample_many: batches in flight on replicas / streams give what the one-at-a-time calls give (trajectories are independent)."""
    from flocoder_amd.sampling import euler_sampler, sample_many
    sd, x, ids, want = reference
    m = _model()
    g = torch.Generator().manual_seed(5)
    srcs = [torch.randn(B, 4, 32, 32, generator=g).to(DEV) for _ in range(5)]
    cls = [torch.randint(102, (B,), generator=g).to(DEV) for _ in range(5)]
    outs = sample_many(m, (B, 4, 32, 32), [({"class_cond": c}, s) for c, s in zip(cls, srcs)], method="euler", n_steps=STEPS, in_flight=2)
    torch.cuda.synchronize()
    assert len(outs) == 5 and len(m._replicas) == 1
    for c, s, o in zip(cls, srcs, outs):
        one = euler_sampler(m, (B, 4, 32, 32), STEPS, cond=c, source=s)[0]       # back on the default stream: the exclusive plan
        assert torch.isfinite(o).all() and rel_l2(o, one) < 1e-6
    assert m.meeting_launches > 0
    assert rel_l2(sample_many(m, (B, 4, 32, 32), [({"class_cond": ids.to(DEV)}, x.to(DEV))], n_steps=STEPS, in_flight=1)[0].cpu(), want) < 1e-4
