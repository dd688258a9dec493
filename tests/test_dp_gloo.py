"""Data-parallel path on CPU: world_size 2 and 4, gloo (the GPU path uses the same ali_hip.dp collectives over RCCL)."""
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
    per = 16 if same_data else 32 // world                 # shard the global batch of 32 across the replicas
    lo = 0 if same_data else per * rank
    images, c = orc.mnist_scale_batch(x[lo:lo + per], {k: v[lo:lo + per] for k, v in a.items()}, stats)
    torch.manual_seed(100 if same_data else 100 + rank)    # per-rank z / dropout streams
    z = torch.randn(per, 512, 1, 1)
    seen = []

    class CheckedSync(dp.GradSync):
        def __call__(self, params):
            local = torch.cat([p.grad.reshape(-1) for p in params if p.grad is not None]).clone()
            gathered = [torch.empty_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            super().__call__(params)
            synced = torch.cat([p.grad.reshape(-1) for p in params if p.grad is not None])
            # equal up to the order in which the ring adds the ranks' terms: fp32 rounding of the terms' magnitudes
            ref, mag = sum(gathered) / world, sum(g.abs() for g in gathered) / world
            seen.append(bool(((synced - ref).abs() <= 1e-6 * mag + 1e-12).all()))

    r = ali_step(E, G, D, oe, od, images, c, z, grad_sync=CheckedSync())
    dp.average_buffers_([b for n, b in D.named_buffers() if "running" in n])
    digest = orc.weights_digest(E, G, D)
    torch.save({"digest": digest, "seen": seen, "loss": float(r["loss_eg"])}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,same_data", [(2, True), (2, False), (4, False), (4, True)])
def test_replicas_gloo(tmp_path, world, same_data):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, same_data, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"r{i}.pt") for i in range(world)]
    r0 = res[0]
    assert all(r["digest"] == r0["digest"] for r in res), "replicas diverged"   # weights + BN buffers identical on all ranks
    # 3 collectives / iteration, each = mean of the ranks' local gradients
    assert all(len(r["seen"]) == 3 and all(r["seen"]) for r in res)
    if not same_data:
        assert len({r["loss"] for r in res}) == world, "the ranks were supposed to see different shards"
    if same_data and world == 2:
        # (g + g) / 2 == g exactly: must equal the single-process run (with 4 ranks the sum of four equal values is
        # exact too, but gloo's ring adds them pairwise in an order that differs from rank to rank only in theory;
        # the 2-rank case is the pin)
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
