"""Generate tests/golden/*.npz by running the REAL reference (build container only).

    python oracle/gen_golden.py [--ref /root/reference]

The reference (wtaylor17/ImageCFGen-Pytorch, mounted read-only) is imported with
empty stubs for four absent third-party modules that take no part in the E/G/D
arithmetic (pytorch_msssim, torchaudio, librosa, seaborn -- SURVEY.md 8c).
Only *data* is written: seeded inputs and the reference's outputs / gradient
pins.  No reference source is copied.  The fixtures pin ``oracle/ali_oracle.py``
(see tests/test_oracle_golden.py); the GPU path is then checked against the oracle.
"""
import argparse
import os
import sys
import types

sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only reference

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _stub_missing():
    for name in ["pytorch_msssim", "torchaudio", "torchaudio.transforms", "librosa", "librosa.core", "seaborn"]:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
    if not hasattr(sys.modules["pytorch_msssim"], "ssim"):
        def _no_ssim(*a, **k):
            raise RuntimeError("pytorch_msssim is not installed")
        sys.modules["pytorch_msssim"].ssim = _no_ssim
    if "torchaudio.transforms" in sys.modules:
        sys.modules["torchaudio"].transforms = sys.modules["torchaudio.transforms"]
    if "librosa.core" in sys.modules:
        sys.modules["librosa"].core = sys.modules["librosa.core"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    args = ap.parse_args()
    _stub_missing()
    sys.path.insert(0, args.ref)

    import numpy as np
    import torch
    import torch.nn as nn
    import ali_oracle as orc

    import image_scms.mnist as ref_mnist
    import image_scms.audio_mnist as ref_audio
    import image_scms.whalecalls as ref_whale
    import image_scms.esrf_acoustic as ref_esrf

    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(8)
    meta = dict(torch_version=torch.__version__, numpy_version=np.__version__)

    # ---------------------------------------------------------------- A. trajectory
    def run_traj(n, bs, tag):
        x, a = orc.synth_morphomnist(n, seed=1)
        losses = []

        class RecBCE(nn.BCEWithLogitsLoss):
            def forward(self, inp, tgt):
                out = super().forward(inp, tgt)
                losses.append(float(out.detach()))
                return out

        orig = nn.BCEWithLogitsLoss
        nn.BCEWithLogitsLoss = RecBCE
        printed = []
        import builtins
        orig_print = builtins.print
        builtins.print = lambda *p, **k: printed.append(p)
        try:
            torch.manual_seed(1)
            np.random.seed(1)
            E, G, D, oD, oE = ref_mnist.train(x, a, n_epochs=1, save_images_every=None, batch_size=bs,
                                              d_updates_per_g_update=1)
        finally:
            nn.BCEWithLogitsLoss = orig
            builtins.print = orig_print
        scores = [p for p in printed if len(p) == 2 and isinstance(p[0], float)][-1]
        # per iteration the reference calls gan_loss 4x: (valid,fake) for EG, D real, D fake
        L = np.array(losses, dtype=np.float64).reshape(-1, 4)
        out = dict(n=n, bs=bs, scores=np.array(scores, dtype=np.float64),
                   loss_eg=(L[:, 0] + L[:, 1]) / 2, bce_calls=L,
                   weights_digest=orc.weights_digest(E, G, D))
        for nm, mod in (("E", E), ("G", G), ("D", D)):
            for k, v in mod.state_dict().items():
                out[f"stats.{nm}.{k}"] = np.array(orc.tensor_stats(v))
        out["x_digest"] = orc.tensor_digest(x)
        out["a_digest"] = orc.tensor_digest(torch.cat([a[k] for k in sorted(a)], dim=1))
        if n <= 256:
            out["x_u8"] = x.to(torch.uint8).numpy()
            out["a_cat"] = torch.cat([a[k] for k in sorted(a)], dim=1).numpy()
        np.savez_compressed(os.path.join(args.out, f"mnist_traj_{tag}.npz"), **meta, **out)
        orig_print(tag, "scores", scores, "digest", out["weights_digest"], "losses", L[:, 2:].tolist())

    run_traj(192, 64, "n192_bs64")
    run_traj(1024, 512, "n1024_bs512")

    # ---------------------------------------------------------------- B/C. module level
    def grads_pin(out_dict, prefix, module):
        for k, p in module.named_parameters():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            out_dict[f"{prefix}.grad.{k}"] = np.array(orc.tensor_stats(g))
            out_dict[f"{prefix}.grad_digest.{k}"] = orc.tensor_digest(g)

    def module_fixture(family, ref_mod, B, d, nominal_std, tag):
        out = {}
        torch.manual_seed(11)
        np.random.seed(11)
        if family == "mnist":
            E, G, D = ref_mod.Encoder(), ref_mod.Generator(), ref_mod.Discriminator()
            for m in (E, G, D):
                m.apply(ref_mod.init_weights)
        else:
            E, G, D = ref_mod.Encoder(d), ref_mod.Generator(d), ref_mod.Discriminator(d)
            for m in (E, G, D):
                m.apply(ref_mod.init_weights)
        for i, m in enumerate((E, G, D)):
            orc.rescale_for_test_(m, nominal_std, bias_seed=7 + i)
        out["init_digest"] = orc.weights_digest(E, G, D)

        if family == "mnist":
            xs, a = orc.synth_morphomnist(B, seed=3)
            stats = {k: (v.min(dim=0).values, v.max(dim=0).values) for k, v in a.items() if k != "digit"}
            images, c = orc.mnist_scale_batch(xs, a, stats)
            z = torch.randn(B, 512, 1, 1, generator=torch.Generator().manual_seed(5))
        else:
            images, c, z = orc.synth_spect_batch(family, B, seed=3)
        gcot = torch.Generator().manual_seed(9)

        def cot(t):
            return torch.randn(t.shape, generator=gcot)

        # --- E
        E.train()
        ex = E(images, c)
        w = cot(ex)
        (ex * w).sum().backward()
        out["E.out"] = ex.detach().numpy()
        grads_pin(out, "E", E)
        # --- G (grads w.r.t. z and params)
        G.train()
        zz = z.clone().requires_grad_(True)
        gz = G(zz, c)
        w = cot(gz)
        (gz * w).sum().backward()
        out["G.out_stats"] = np.array(orc.tensor_stats(gz))
        out["G.out_digest"] = orc.tensor_digest(gz)
        out["G.out_head"] = gz.detach().reshape(B, -1)[:, :64].numpy()
        out["G.gz"] = zz.grad.reshape(B, -1).numpy()
        grads_pin(out, "G", G)
        # --- D eval mode (no dropout, BN running stats) -- inputs (images, ex) both get grads
        D.eval()
        xi = images.clone().requires_grad_(True)
        zi = ex.detach().clone().requires_grad_(True)
        dl = D(xi, zi, c)
        w = cot(dl)
        (dl * w).sum().backward()
        out["D_eval.out"] = dl.detach().numpy()
        out["D_eval.gz"] = zi.grad.reshape(B, -1).numpy()
        out["D_eval.gx_stats"] = np.array(orc.tensor_stats(xi.grad))
        out["D_eval.gx_digest"] = orc.tensor_digest(xi.grad)
        grads_pin(out, "D_eval", D)
        D.zero_grad()
        # --- D train mode (dropout masks drawn from the global generator after manual_seed(21))
        D.train()
        torch.manual_seed(21)
        xi = images.clone().requires_grad_(True)
        zi = ex.detach().clone().requires_grad_(True)
        dl = D(xi, zi, c)
        w = cot(dl)
        (dl * w).sum().backward()
        out["D_train.out"] = dl.detach().numpy()
        out["D_train.gz"] = zi.grad.reshape(B, -1).numpy()
        out["D_train.gx_stats"] = np.array(orc.tensor_stats(xi.grad))
        out["D_train.gx_digest"] = orc.tensor_digest(xi.grad)
        grads_pin(out, "D_train", D)
        for k, v in D.state_dict().items():
            if "running" in k:
                out[f"D_train.buf.{k}"] = v.numpy()
        out["B"], out["d"], out["nominal_std"] = B, d, nominal_std
        np.savez_compressed(os.path.join(args.out, f"modules_{tag}.npz"), **meta, **out)
        print(tag, "E.out", orc.tensor_stats(ex)[:2], "D_eval", dl.detach().reshape(-1)[:2].tolist())

    module_fixture("mnist", ref_mnist, 4, 64, 0.01, "mnist_b4")
    module_fixture("audio", ref_audio, 2, 64, 0.001, "audio_d64_b2")
    module_fixture("audio", ref_audio, 2, 8, 0.001, "audio_d8_b2")
    module_fixture("whale", ref_whale, 1, 16, 0.001, "whale_d16_b1")
    module_fixture("esrf", ref_esrf, 1, 8, 0.001, "esrf_d8_b1")

    # ---------------------------------------------------------------- D. caller traces
    # finetune_mnist_bigan.py:64-85 (mse metric): 2 optimiser steps from a fixed E/G
    torch.manual_seed(31)
    np.random.seed(31)
    E, G = ref_mnist.Encoder(), ref_mnist.Generator()
    E.apply(ref_mnist.init_weights), G.apply(ref_mnist.init_weights)
    orc.rescale_for_test_(E, 0.01, bias_seed=7), orc.rescale_for_test_(G, 0.01, bias_seed=8)
    xs, a = orc.synth_morphomnist(8, seed=4)
    stats = {k: (v.min(dim=0).values, v.max(dim=0).values) for k, v in a.items() if k != "digit"}
    x, c = orc.mnist_scale_batch(xs, a, stats)
    E.train(), G.eval()
    opt = torch.optim.Adam(E.parameters(), lr=1e-4)
    rec, lat = [], []
    for _ in range(2):
        opt.zero_grad()
        codes = E(x, c)
        xr = G(codes, c)
        rl = torch.square(x - xr).mean()
        ll = torch.square(codes).mean()
        (rl + ll).backward()
        opt.step()
        rec.append(rl.item()), lat.append(ll.item())
    with torch.no_grad():  # mnist_generator_score.py:69-74
        zf = torch.randn(8, 512, 1, 1, generator=torch.Generator().manual_seed(6))
        gen = G(zf, c)
    np.savez_compressed(os.path.join(args.out, "callers_mnist.npz"), **meta, rec=np.array(rec), lat=np.array(lat),
                        E_digest=orc.weights_digest(E), gen_stats=np.array(orc.tensor_stats(gen)),
                        gen_head=gen.reshape(8, -1)[:, 300:364].numpy())
    print("callers", rec, lat)


if __name__ == "__main__":
    main()
