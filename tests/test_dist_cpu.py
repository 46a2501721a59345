"""World-size-2 gloo rehearsal of the multi-GPU path (flocoder_amd/dist.py): sample sharding, the one weight
broadcast, the optional result gather.  CPU only; the GPU ranks differ only in backend ("nccl" = RCCL)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from flocoder_amd import dist as fdist
    from flocoder_amd.unet import Unet
    r, lr, w = fdist.init(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                       # ranks start with DIFFERENT weights
    m = Unet(dim=8, channels=4, n_classes=3)
    before = torch.cat([p.reshape(-1) for p in m.parameters()]).clone()
    moved = fdist.broadcast_weights(m, src=0)
    after = torch.cat([p.reshape(-1) for p in m.parameters()])
    torch.manual_seed(100)
    ref = torch.cat([p.reshape(-1) for p in Unet(dim=8, channels=4, n_classes=3).parameters()])
    lo, hi = fdist.shard_range(7, rank, world)          # ragged: 7 samples over 2 ranks -> 4 + 3
    local = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1, 1).repeat(1, 4, 2, 2)
    full = fdist.gather_samples(local, 7)
    q.put((rank, moved, bool(torch.equal(after, ref)), bool(torch.equal(before, ref)), (lo, hi), full[:, 0, 0, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    from flocoder_amd.dist import shard_range
    for n in (0, 1, 7, 64, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(256, 3, 8) == (96, 128)          # BASELINE config 3: 256 samples -> 32 per GPU


@pytest.mark.timeout(300)
def test_broadcast_and_gather_world2_gloo():
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, moved0, same0, was0, span0, full0), (r1, moved1, same1, was1, span1, full1) = res
    assert moved0 == moved1 > 0
    assert same0 and same1                               # both ranks hold rank 0's weights afterwards
    assert was0 and not was1                             # ... and rank 1 did not before
    assert span0 == (0, 4) and span1 == (4, 7)
    assert full0 == full1 == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0]   # gather restores global sample order


def _grad_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from flocoder_amd import dist as fdist
    fdist.init(backend="gloo")
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)          # rank r holds (r+1) * base
    n = fdist.average_gradients(g, bucket_bytes=1024)                 # 256-float buckets -> 4 collectives
    q.put((rank, n, g.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_averaging_world2_gloo():
    """The training step's only collective (flocoder_amd.train.FlowTrainer.step): bucketed all-reduce + divide == DDP averaging."""
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = (torch.arange(1000, dtype=torch.float32) * 1.5).tolist()
    assert res[0][1] == res[1][1] == 4
    assert res[0][2] == want and res[1][2] == want


def _agree_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import types
    import torch.distributed as dist
    from flocoder_amd import dist as fdist
    from flocoder_amd.train import FlowTrainer
    fdist.init(backend="gloo")
    # the trainer's agreement step on a stand-in that carries exactly the state it touches (the real object needs a GPU)
    t = types.SimpleNamespace(distributed=True, pg=None, device=torch.device("cpu"),
                              _groups={"class": (10, 20), "fusion": (30, 40), "inject": (0, 0)},
                              grads=torch.full((50,), float(rank + 1)),
                              _id_flag=torch.zeros(1, dtype=torch.int32), _id_flag_host=torch.zeros(1, dtype=torch.int32),
                              model=types.SimpleNamespace(_cfg=types.SimpleNamespace(n_classes=10)))
    t.check_class_ids = lambda: FlowTrainer.check_class_ids(t)
    # rank 0 trained WITH class conditioning, rank 1 dropped it (cond=None); nobody had a mask
    present = FlowTrainer._agree(t, {"class": rank == 0, "fusion": False, "inject": False})
    fdist.average_gradients(t.grads)
    # second round: rank 1's batch carried a class id outside the table (its step prologue set the device flag): BOTH ranks must raise,
    # inside the agreement, before either enters the gradient all-reduce (one alone would leave the other hanging there)
    if rank == 1:
        t._id_flag.fill_(1)
    raised = False
    try:
        FlowTrainer._agree(t, {"class": True, "fusion": False, "inject": False})
    except IndexError:
        raised = True
    q.put((rank, present, t.grads.tolist(), raised))
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_agree_on_adam_groups_world2_gloo():
    """One rank drops its conditioning, the other does not: both must step class_cond_mlp.* (any-rank-has-a-gradient), and the rank
    without one must contribute zeros -- not last step's leftovers -- to the average (flocoder_amd.train.FlowTrainer._agree)."""
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0, g0, e0), (_, p1, g1, e1) = res
    assert e0 and e1, "a bad class id on one rank must raise IndexError on every rank, in the same step"
    assert p0 == p1 == {"class": True, "fusion": False, "inject": False}
    assert g0 == g1                                              # identical averaged gradients -> identical Adam updates
    assert g0[0] == g0[45] == 1.5 and g0[15] == 0.5              # class range: (1 + 0) / 2 -- rank 1's stale 2.0 was zeroed first
    assert g0[35] == 0.0                                         # a group nobody had: zeros everywhere, and Adam skips it on every rank
