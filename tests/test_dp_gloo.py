"""Data-parallel path on CPU: world_size=2, gloo (the GPU path uses the same ali_hip.dp collectives over RCCL)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, same_data, out_dir):
    for p in (os.path.join(ROOT, "imagecfgen-pytorch_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import ali_oracle as orc
    import image_scms.mnist as pm
    from ali_hip import dp
    from image_scms.training_utils import ali_step
    torch.manual_seed(7)                                   # identical replicas
    E, G, D = pm.Encoder(), pm.Generator(), pm.Discriminator()
    for i, m in enumerate((E, G, D)):
        m.apply(pm.init_weights)
        orc.rescale_for_test_(m, 0.01, bias_seed=7 + i)
        m.train()
    oe = torch.optim.Adam(list(E.parameters()) + list(G.parameters()), lr=1e-4, betas=(0.5, 0.999))
    od = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.999))
    x, a = orc.synth_morphomnist(32, seed=1)
    stats = {k: (v.min(dim=0).values, v.max(dim=0).values) for k, v in a.items() if k != "digit"}
    lo = 0 if same_data else 16 * rank                     # shard the global batch of 32 across the 2 replicas
    images, c = orc.mnist_scale_batch(x[lo:lo + 16], {k: v[lo:lo + 16] for k, v in a.items()}, stats)
    torch.manual_seed(100 if same_data else 100 + rank)    # per-rank z / dropout streams
    z = torch.randn(16, 512, 1, 1)
    seen = []

    class CheckedSync(dp.GradSync):
        def __call__(self, params):
            local = torch.cat([p.grad.reshape(-1) for p in params if p.grad is not None]).clone()
            gathered = [torch.empty_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            super().__call__(params)
            synced = torch.cat([p.grad.reshape(-1) for p in params if p.grad is not None])
            seen.append(torch.allclose(synced, sum(gathered) / world, rtol=1e-6, atol=1e-9))

    r = ali_step(E, G, D, oe, od, images, c, z, grad_sync=CheckedSync())
    dp.average_buffers_([b for n, b in D.named_buffers() if "running" in n])
    digest = orc.weights_digest(E, G, D)
    torch.save({"digest": digest, "seen": seen, "loss": float(r["loss_eg"])}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("same_data", [True, False])
def test_two_replicas_gloo(tmp_path, same_data):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, same_data, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"r{i}.pt") for i in range(2))
    assert r0["digest"] == r1["digest"], "replicas diverged"           # weights + BN buffers identical on all ranks
    assert len(r0["seen"]) == 3 and all(r0["seen"]) and all(r1["seen"])  # 3 collectives / iteration, each = mean of locals
    if same_data:
        # (g + g) / 2 == g exactly: must equal the single-process run
        for p in (os.path.join(ROOT, "imagecfgen-pytorch_amd"), os.path.join(ROOT, "oracle")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import ali_oracle as orc
        import image_scms.mnist as pm
        from image_scms.training_utils import ali_step
        torch.manual_seed(7)
        E, G, D = pm.Encoder(), pm.Generator(), pm.Discriminator()
        for i, m in enumerate((E, G, D)):
            m.apply(pm.init_weights)
            orc.rescale_for_test_(m, 0.01, bias_seed=7 + i)
            m.train()
        oe = torch.optim.Adam(list(E.parameters()) + list(G.parameters()), lr=1e-4, betas=(0.5, 0.999))
        od = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.999))
        x, a = orc.synth_morphomnist(32, seed=1)
        stats = {k: (v.min(dim=0).values, v.max(dim=0).values) for k, v in a.items() if k != "digit"}
        images, c = orc.mnist_scale_batch(x[:16], {k: v[:16] for k, v in a.items()}, stats)
        torch.manual_seed(100)
        z = torch.randn(16, 512, 1, 1)
        torch.set_num_threads(2)
        ali_step(E, G, D, oe, od, images, c, z)
        assert orc.weights_digest(E, G, D) == r0["digest"]
