"""The hand-scheduled stepper's data-parallel schedule (flat-buffer all-reduces issued asynchronously and overlapped
with the next phase's independent forward, one HIP graph per segment) on the single GPU of the test box: two ranks
share cuda:0 and exchange over gloo (RCCL refuses two ranks on one device; the collective calls are the same
``ali_hip.dp`` functions).

Every rank trains on its OWN shard, z and Dropout2d stream.  The reference result is a single-process
emulation: two replicas of the same stepper advanced segment by segment, their flat gradient buffers summed where the
schedule all-reduces (1/world folded into Adam), per-replica BatchNorm batch statistics, running statistics averaged
at the end of the iteration.  The 2-rank run must reproduce it bit for bit, and its replicas must stay identical."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BS = 64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _paths():
    for p in (os.path.join(ROOT, "imagecfgen-pytorch_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _replica(rank, process_group, capture, pipeline=False):
    """Identical weights on every rank (seed 7); shard, z and dropout seed by rank."""
    _paths()
    import ali_hip
    import ali_oracle as orc
    import image_scms.mnist as pm
    from ali_hip.step import AliStepper
    torch.manual_seed(7)
    E, G, D = pm.Encoder(), pm.Generator(), pm.Discriminator()
    for i, m in enumerate((E, G, D)):
        m.apply(pm.init_weights)
        orc.rescale_for_test_(m, 0.01, bias_seed=7 + i)
        m.cuda().train()
    ali_hip.manual_seed(11 + rank)
    st = AliStepper(E, G, D, process_group=process_group, capture=capture, pipeline_reduce=pipeline)
    x, a = orc.synth_morphomnist(2 * BS, seed=1)
    stats = {k: (v.min(dim=0).values, v.max(dim=0).values) for k, v in a.items() if k != "digit"}
    lo = rank * BS
    images, c = orc.mnist_scale_batch(x[lo:lo + BS], {k: v[lo:lo + BS] for k, v in a.items()}, stats)
    z = torch.randn(BS, 512, 1, 1, generator=torch.Generator().manual_seed(3 + rank))
    return st, D, (images.cuda(), {k: v.cuda() for k, v in c.items()}, z.cuda())


def _digest(st, D):
    import ali_oracle as orc
    torch.cuda.synchronize()
    state = [st.opt_eg.flat, st.opt_d.flat, st.opt_eg.m, st.opt_d.v] + [b.float() for _, b in D.named_buffers()]
    return orc.tensor_digest(torch.cat([t.reshape(-1).cpu() for t in state]))


def _worker(rank, world, port, capture, iters, out_dir, pipeline=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.set_num_threads(2)
    st, D, (images, c, z) = _replica(rank, dist.group.WORLD, capture, pipeline)
    out = None
    for _ in range(iters):       # (pipelined: every step announces the next batch -- here the same tensors again)
        out = st.step(images, c, z, ahead=(images, c, z) if pipeline else None)
    torch.save({"digest": _digest(st, D), "out": {k: float(v) for k, v in out.items()}},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


class _DropoutState:
    """The Dropout2d stream is process-global state (ali_hip.dropout._state); a replica of the emulation owns a copy
    that is swapped in while its kernels are being launched."""

    def __init__(self):
        from ali_hip import dropout
        self.mod = dropout
        self.saved = dict(dropout._state)

    def __enter__(self):
        self.outer = dict(self.mod._state)
        self.mod._state.clear()
        self.mod._state.update(self.saved)

    def __exit__(self, *exc):
        self.saved = dict(self.mod._state)
        self.mod._state.clear()
        self.mod._state.update(self.outer)


def _emulate(world, iters):
    """Single-process gradient-accumulation emulation of the ``world``-rank run (eager launches)."""
    reps = []
    for r in range(world):
        st, D, data = _replica(r, None, False)
        st.world = world                     # 1/world is folded into the Adam kernel; no process group: no collectives
        st.segmented = True                  # ... but the data-parallel schedule (segments cut where the all-reduces go)
        reps.append((st, D, data, _DropoutState()))
    outs = None
    with torch.no_grad():
        for _ in range(iters):
            cxs = []
            for st, D, (images, c, z), ds in reps:
                with ds:
                    cxs.append(st._begin(images, c, z, True))
            n_seg = len(reps[0][0]._segments(True))
            for i in range(n_seg):
                group_attr = None
                for (st, D, data, ds), cx in zip(reps, cxs):
                    work, group, _wait = st._segments(True)[i]
                    with ds:
                        work(cx)
                    if group is not None:
                        group_attr = "opt_eg" if group is st.opt_eg else "opt_d"
                if group_attr is not None:   # the all-reduce: every replica ends up with the sum over ranks
                    total = getattr(reps[0][0], group_attr).grad.clone()
                    for st, *_ in reps[1:]:
                        total += getattr(st, group_attr).grad
                    for st, *_ in reps:
                        getattr(st, group_attr).grad.copy_(total)
            # dp.average_buffers_: sum over ranks, times 1/world
            for bufs in zip(*[st.bn_buffers for st, *_ in reps]):
                total = bufs[0].float().clone()
                for b in bufs[1:]:
                    total += b.float()
                total.mul_(1.0 / world)
                for b in bufs:
                    b.copy_(total)
            outs = [{k: float(v) for k, v in cx["out"].items()} for cx in cxs]
    from ali_hip import chain
    chain.abort_batch_counts()
    return [_digest(st, D) for st, D, *_ in reps], outs


@pytest.mark.parametrize("capture,pipeline", [(False, False), (True, False), (False, True), (True, True)])
def test_stepper_two_ranks_own_shards_vs_emulation(tmp_path, capture, pipeline):
    """``pipeline``: AliStepper(pipeline_reduce=True) -- the last all-reduce of an iteration overlaps with the next
    iteration's E(x) / G(z) forward passes; same result as the plain data-parallel schedule, bit for bit."""
    iters = 3
    port = _free_port()
    mp.spawn(_worker, args=(2, port, capture, iters, str(tmp_path), pipeline), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"r{i}.pt") for i in range(2))
    assert r0["digest"] == r1["digest"], "replicas diverged"
    assert r0["out"] != r1["out"], "the ranks were supposed to see different shards"
    digests, outs = _emulate(2, iters)
    assert digests[0] == digests[1]
    assert outs[0] == r0["out"] and outs[1] == r1["out"], (outs, r0["out"], r1["out"])
    assert digests[0] == r0["digest"], "2-rank data-parallel run != single-process gradient-accumulation emulation"
