"""MorphoMNIST ALI models and training loop -- drop-in for the reference's
``image_scms/mnist.py`` (Encoder :21-56, Generator :59-86, Discriminator :89-154,
train :157-299, load_model :302-313).

Same class names, constructor order (=> identical RNG consumption and
``state_dict`` keys: ``layers.{0,2,4,6,8}``, ``dx.{1,5,9,13,17}``, ...), same call
signatures.  The ``nn`` sub-modules own the parameters in the reference layouts;
on CUDA tensors ``forward`` never calls them -- it hands the stack to
``ali_hip.chain`` (NHWC activations, fp32-MFMA implicit-GEMM kernels, fused
epilogues).  On CPU tensors the stock torch ops of the same modules run (that is
the torch-CPU meaning of the module, used for plumbing tests and `device='cpu'`
callers; it is never used for CUDA inputs and is not the oracle).
"""
from typing import Dict

import numpy as np
import torch
import torch.nn as nn

from .training_utils import AdversariallyLearnedInference  # noqa: F401  (re-exported like the reference)
from .training_utils import ali_step, batchify, batchify_dict, init_weights

LATENT_DIM = 512
N_CONTINUOUS = 3
AttributeDict = Dict[str, torch.Tensor]
_IMG = 28


def continuous_feature_map(c: torch.Tensor, size: tuple = (28, 28)):
    return c.reshape((c.size(0), 1, 1, 1)).repeat(1, 1, *size)


def _plane_embedding():
    return nn.Sequential(nn.Embedding(10, 256), nn.Unflatten(1, (1, 16, 16)), nn.Upsample(size=(_IMG, _IMG)),
                         nn.Tanh())


def _cont_keys(c):
    return sorted(k for k in c if k != "digit")


def _torch_features(embedding, X, c):
    planes = [continuous_feature_map(c[k], size=(_IMG, _IMG)) for k in _cont_keys(c)]
    return torch.concat([X, embedding(c["digit"].argmax(1))] + planes, dim=1)


def _hip_features(embedding, X, c):
    """[B,28,28,8] NHWC conv input: image, tanh(upsampled digit embedding), attribute planes, zero pad."""
    from ali_hip import planes
    B = X.shape[0]
    keys = _cont_keys(c)
    idx = c["digit"].argmax(1).to(torch.int32).reshape(B, 1).contiguous()
    cont = torch.cat([c[k].reshape(B, 1).float() for k in keys], dim=1) if keys else None
    n_log = 2 + len(keys)
    x0 = planes.assemble(X.reshape(B, _IMG, _IMG).float(), idx, cont, (n_log + 3) // 4 * 4, [embedding[0].weight])
    return x0, n_log


class Encoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.digit_embedding = _plane_embedding()
        widths = [(1 + N_CONTINUOUS + 1, 64, 3, 1), (64, 128, 4, 1), (128, 256, 4, 1), (256, 512, 4, 1)]
        mods = []
        for ci, co, k, p in widths:
            mods += [nn.Conv2d(ci, co, (k, k), (2, 2), p), nn.LeakyReLU(0.2)]
        mods.append(nn.Conv2d(512, LATENT_DIM, (1, 1), (2, 2)))
        self.layers = nn.Sequential(*mods)

    @property
    def device(self):
        return next(self.parameters()).device

    def forward(self, X: torch.Tensor, c: AttributeDict):
        if not X.is_cuda:
            return self.layers(_torch_features(self.digit_embedding, X, c))
        from ali_hip.chain import run_chain
        x0, n_log = _hip_features(self.digit_embedding, X, c)
        return run_chain(self.layers, x0, n_log).reshape(X.shape[0], LATENT_DIM, 1, 1)


class Generator(nn.Module):
    def __init__(self):
        super().__init__()
        self.digit_embedding = nn.Embedding(10, 256)
        spec = [(LATENT_DIM + 256 + N_CONTINUOUS, 512, 3, 1, 0), (512, 256, 3, 2, 0), (256, 128, 3, 2, 1),
                (128, 64, 3, 2, 1)]
        mods = []
        for ci, co, k, s, p in spec:
            mods += [nn.ConvTranspose2d(ci, co, (k, k), (s, s), (p, p)), nn.LeakyReLU(0.2)]
        mods += [nn.ConvTranspose2d(64, 1, (4, 4)), nn.Tanh()]
        self.layers = nn.Sequential(*mods)

    def forward(self, z: torch.Tensor, c: AttributeDict):
        B = z.shape[0]
        digit = c["digit"].matmul(self.digit_embedding.weight)          # soft one-hots stay differentiable
        if not z.is_cuda:
            planes = [continuous_feature_map(c[k], size=(1, 1)) for k in _cont_keys(c)]
            return self.layers(torch.concat([z, digit.reshape((-1, 256, 1, 1))] + planes, dim=1))
        from ali_hip.chain import run_chain
        feats = [z.reshape(B, LATENT_DIM).float(), digit.reshape(B, 256)] + [c[k].reshape(B, 1).float()
                                                                             for k in _cont_keys(c)]
        n_log = sum(f.shape[1] for f in feats)
        pad = (-n_log) % 32          # channel stride % 32 == 0 -> uniform-tap fast path of the GEMM kernel
        if pad:
            feats.append(torch.zeros(B, pad, device=z.device))
        x0 = torch.cat(feats, dim=1).reshape(B, 1, 1, n_log + pad)
        return run_chain(self.layers, x0, n_log).reshape(B, 1, _IMG, _IMG)


class Discriminator(nn.Module):
    def __init__(self):
        super().__init__()
        self.digit_embedding = _plane_embedding()
        self.dz = nn.Sequential(
            nn.Dropout2d(0.2), nn.Conv2d(512, 512, (1, 1), (1, 1)), nn.LeakyReLU(0.1),
            nn.Dropout2d(0.5), nn.Conv2d(512, 512, (1, 1), (1, 1)), nn.LeakyReLU(0.1))
        dx = [nn.Dropout2d(0.2), nn.Conv2d(1 + N_CONTINUOUS + 1, 32, (5, 5), (1, 1)), nn.LeakyReLU(0.1),
              nn.Dropout2d(0.2), nn.BatchNorm2d(32)]
        for ci, co, k, s in [(32, 64, 4, 2), (64, 128, 4, 1), (128, 256, 4, 2)]:
            dx += [nn.Conv2d(ci, co, (k, k), (s, s)), nn.LeakyReLU(0.1), nn.BatchNorm2d(co), nn.Dropout2d(0.5)]
        dx += [nn.Conv2d(256, 512, (3, 3), (1, 1)), nn.LeakyReLU(0.1)]
        self.dx = nn.Sequential(*dx)
        self.dxz = nn.Sequential(
            nn.Dropout2d(0.2), nn.Conv2d(1024, 1024, (1, 1), (1, 1)), nn.LeakyReLU(0.1),
            nn.Dropout2d(0.2), nn.Conv2d(1024, 1024, (1, 1), (1, 1)), nn.LeakyReLU(0.1),
            nn.Dropout2d(0.2), nn.Conv2d(1024, 1, (1, 1), (1, 1)))

    @property
    def device(self):
        return next(self.parameters()).device

    def forward(self, X: torch.Tensor, z: torch.Tensor, c: AttributeDict):
        if not X.is_cuda:
            dx = self.dx(_torch_features(self.digit_embedding, X, c))
            dz = self.dz(z)
            return self.dxz(torch.concat([dx, dz], dim=1)).reshape((-1, 1))
        from ali_hip.chain import run_chain
        B = X.shape[0]
        x0, n_log = _hip_features(self.digit_embedding, X, c)
        dx = run_chain(self.dx, x0, n_log)                                   # masks: dx first ...
        dz = run_chain(self.dz, z.reshape(B, 1, 1, LATENT_DIM).float())      # ... then dz ...
        joint = torch.cat([dx.reshape(B, 512), dz.reshape(B, LATENT_DIM)], dim=1).reshape(B, 1, 1, 1024)
        return run_chain(self.dxz, joint).reshape(-1, 1)                     # ... then dxz (reference :152-154)


def _scale_batch(images, attrs, attr_stats, device):
    images = 2 * images.reshape((-1, 1, _IMG, _IMG)).float().to(device) / 255 - 1
    c = {k: (2 * (attrs[k] - lo) / (hi - lo) - 1).to(device) for k, (lo, hi) in attr_stats.items()}
    c["digit"] = attrs["digit"].to(device)
    return images, c


def train(x_train: torch.Tensor,
          a_train: AttributeDict,
          x_test=None,
          a_test=None,
          n_epochs=200,
          l_rate=1e-4,
          device='cpu',
          save_images_every=2,
          image_output_path='',
          batch_size=64,
          d_updates_per_g_update=1,
          use_stepper=None,
          checkpoint_every=None,
          checkpoint_path=None):
    """Same signature, RNG order and return value as the reference's train (mnist.py:157-299).
    ``checkpoint_every`` (epochs) + ``checkpoint_path``: periodic resumable checkpoints in the state-dict format
    ``load_model`` reads, plus both Adam states (the reference only saves at the end, from its caller).

    On a CUDA device the iteration runs on the hand-scheduled ``AliStepper`` (HIP-graph replay; ``use_stepper=False``
    keeps the autograd schedule on the same kernels); the two returned optimisers are then its flat Adam groups
    (``state_dict`` / ``zero_grad`` / ``step`` like ``torch.optim.Adam``).  CPU devices run the stock torch ops."""
    E, G, D = Encoder().to(device), Generator().to(device), Discriminator().to(device)
    for m in (E, G, D):
        m.apply(init_weights)
    if use_stepper is None:
        use_stepper = torch.device(device).type == "cuda"
    stepper = None
    if use_stepper:
        from ali_hip.step import AliStepper
        stepper = AliStepper(E, G, D, lr=l_rate, betas=(0.5, 0.999), capture=True)
        optimizer_E, optimizer_D = stepper.opt_eg, stepper.opt_d
    else:
        optimizer_E = torch.optim.Adam(list(E.parameters()) + list(G.parameters()), lr=l_rate, betas=(0.5, 0.999))
        optimizer_D = torch.optim.Adam(D.parameters(), lr=l_rate, betas=(0.5, 0.999))
    gan_loss = nn.BCEWithLogitsLoss()

    for epoch in range(n_epochs):
        for m in (D, E, G):
            m.train()
        perm = np.random.permutation(len(x_train))
        img_batches = batchify(x_train[perm], batch_size=batch_size)
        attr_batches = batchify_dict({k: v[perm] for k, v in a_train.items()}, batch_size=batch_size)
        attr_stats = {k: (v.min(dim=0).values, v.max(dim=0).values) for k, v in a_train.items() if k != "digit"}
        d_score = torch.zeros((), device=device)
        eg_score = torch.zeros((), device=device)
        num_batches = 0
        for i, ((images,), attrs) in enumerate(zip(img_batches, attr_batches)):
            num_batches += 1
            images, c = _scale_batch(images, attrs, attr_stats, device)
            z_mean = torch.zeros((len(images), LATENT_DIM, 1, 1)).float()
            z = torch.normal(z_mean, z_mean + 1).to(device)               # sampled on the host like the reference
            if stepper is not None:
                r = stepper.step(images, c, z, do_eg=(i % d_updates_per_g_update == 0))
            else:
                r = ali_step(E, G, D, optimizer_E, optimizer_D, images, c, z,
                             do_eg=(i % d_updates_per_g_update == 0), gan_loss=gan_loss)
            d_score += r["dg"]                                            # accumulated on device: one sync per epoch
            eg_score += r["de"]
        print(d_score.item() / num_batches, eg_score.item() / num_batches)

        if save_images_every and (epoch + 1) % save_images_every == 0 and x_test is not None:
            _save_demo(E, G, D, x_test, a_test, attr_stats, device, epoch, image_output_path)
        if checkpoint_every and checkpoint_path and (epoch + 1) % checkpoint_every == 0:
            from ._spect import save_checkpoint
            save_checkpoint(checkpoint_path, E, G, D, stepper, optimizer_E, optimizer_D)
    return E, G, D, optimizer_D, optimizer_E


def _save_demo(E, G, D, x_test, a_test, attr_stats, device, epoch, path, n_show=10):
    """generated / real / reconstructed rows for the first test digits (reference :251-297)."""
    for m in (D, E, G):
        m.eval()
    with torch.no_grad():
        x, c = _scale_batch(x_test[:n_show], {k: v[:n_show] for k, v in a_test.items()}, attr_stats, device)
        z = torch.randn(len(x), LATENT_DIM, 1, 1).to(device)
        rows = [G(z, c).reshape(n_show, _IMG, _IMG).cpu().numpy(),
                2 * x_test[:n_show].cpu().numpy() / 255 - 1,
                G(E(x, c), c).reshape(n_show, _IMG, _IMG).cpu().numpy()]
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots(3, n_show, figsize=(15, 5))
    fig.suptitle(f'Epoch {epoch + 1}')
    for r, (row, label) in enumerate(zip(rows, ('G(z, c)', 'x', 'G(E(x, c), c)'))):
        fig.text(0.04, 0.75 - 0.25 * r, label, ha='left')
        for j in range(n_show):
            ax[r, j].imshow(row[j], cmap='gray', vmin=-1, vmax=1)
            ax[r, j].axis('off')
    plt.savefig(f'{path}/epoch-{epoch + 1}.png')
    plt.close()


def load_model(tar_path, device='cpu', return_raw=False):
    """State-dict checkpoint loader (reference :302-313)."""
    obj = torch.load(tar_path, map_location=device)
    E, G, D = Encoder(), Generator(), Discriminator()
    E.load_state_dict(obj['E_state_dict'])
    G.load_state_dict(obj['G_state_dict'])
    D.load_state_dict(obj['D_state_dict'])
    return (E, G, D, obj) if return_raw else (E, G, D)
