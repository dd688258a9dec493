"""The hand-scheduled stepper's data-parallel schedule (flat-buffer all-reduces issued asynchronously and overlapped
with the next phase's independent forward, one HIP graph per segment) on the single GPU of the test box: two ranks
share cuda:0 and exchange over gloo (RCCL refuses two ranks on one device; the collective calls are the same
``ali_hip.dp`` functions).  Both ranks see the same batch, z and dropout seed, so (g + g) * 1/2 == g exactly and the
replicas must reproduce the single-process run bit for bit."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(process_group, capture, iters=3):
    for p in (os.path.join(ROOT, "imagecfgen-pytorch_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
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
    ali_hip.manual_seed(11)
    st = AliStepper(E, G, D, process_group=process_group, capture=capture)
    x, a = orc.synth_morphomnist(64, seed=1)
    stats = {k: (v.min(dim=0).values, v.max(dim=0).values) for k, v in a.items() if k != "digit"}
    images, c = orc.mnist_scale_batch(x, a, stats)
    z = torch.randn(64, 512, 1, 1, generator=torch.Generator().manual_seed(3))
    out = None
    for _ in range(iters):
        out = st.step(images.cuda(), {k: v.cuda() for k, v in c.items()}, z.cuda())
    torch.cuda.synchronize()
    state = [st.opt_eg.flat, st.opt_d.flat, st.opt_eg.m, st.opt_d.v] + [b.float() for _, b in D.named_buffers()]
    return {"digest": orc.tensor_digest(torch.cat([t.reshape(-1).cpu() for t in state])),
            "out": {k: float(v) for k, v in out.items()}}


def _worker(rank, world, port, capture, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    res = _run(dist.group.WORLD, capture)
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("capture", [False, True])
def test_stepper_two_ranks_overlapped_allreduce(tmp_path, capture):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, capture, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"r{i}.pt") for i in range(2))
    assert r0["digest"] == r1["digest"], "replicas diverged"
    single = _run(None, capture)
    assert single["out"] == r0["out"], (single["out"], r0["out"])
    assert single["digest"] == r0["digest"], "2-rank data-parallel run != single-process run on the same shard"
