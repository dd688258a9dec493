"""Shared implementation of the three spectrogram ALI/BiGAN families of the reference
(``image_scms/audio_mnist.py``, ``whalecalls.py``, ``esrf_acoustic.py``): 5x5 stride-2 Conv2d encoder /
discriminator stacks, Linear + 5x5 stride-2 ConvTranspose2d generator stack, categorical attributes as tanh'd
16x16 embedding planes, Adam betas (0.5, 0.9), init std 0.001.

The public modules (``audio_mnist.Encoder`` ...) subclass these with the reference's attribute names and constructor
order, so RNG consumption, ``state_dict`` keys and pickled-module class paths match the reference.  CUDA tensors run
on the HIP kernels through ``ali_hip.chain``; CPU tensors run the stock torch ops of the same sub-modules.
"""
from functools import partial

import torch
import torch.nn as nn

LATENT_DIM = 512


def init_weights(layer, std=0.001):
    """N(0, std) on modules whose class name starts with 'Conv' (NOT the Generator's nn.Linear), zero bias
    (reference audio_mnist.py:33-38, whalecalls.py:23-28, esrf_acoustic.py:24-29)."""
    if layer.__class__.__name__.startswith('Conv'):
        torch.nn.init.normal_(layer.weight, mean=0, std=std)
        if layer.bias is not None:
            torch.nn.init.constant_(layer.bias, 0)


def plane_embedding(n_classes, scale):
    return nn.Sequential(nn.Embedding(n_classes, 256), nn.Unflatten(1, (1, 16, 16)), nn.Upsample(scale_factor=scale),
                         nn.Tanh())


def conv_stack(c_in, widths, d):
    """c2d(...) 5x5 stride-2 pad-1 stack; widths in multiples of d, None = LATENT_DIM; LeakyReLU(0.2) between."""
    c2d = partial(nn.Conv2d, stride=(2, 2), padding=1)
    mods, prev = [], c_in
    for i, w in enumerate(widths):
        co = LATENT_DIM if w is None else w * d
        mods.append(c2d(prev, co, (5, 5)))
        if i + 1 < len(widths):
            mods.append(nn.LeakyReLU(0.2))
        prev = co
    return nn.Sequential(*mods)


def deconv_stack(in_features, widths, d):
    """Linear -> [16d,4,4] -> ct2d(...) 5x5 stride-2 pad-2 outpad-1 stack, LeakyReLU(0.2) ... Tanh."""
    ct2d = partial(nn.ConvTranspose2d, stride=2, padding=2, output_padding=1)
    mods = [nn.Linear(in_features, 256 * d), nn.Unflatten(1, (16 * d, 4, 4)), nn.LeakyReLU(0.2)]
    prev = 16 * d
    for w in widths:
        co = 1 if w is None else w * d
        mods += [ct2d(prev, co, (5, 5)), nn.Tanh() if w is None else nn.LeakyReLU(0.2)]
        prev = co
    return nn.Sequential(*mods)


def dz_stack():
    return nn.Sequential(nn.Conv2d(LATENT_DIM, LATENT_DIM, (1, 1), (1, 1)), nn.LeakyReLU(0.2),
                         nn.Conv2d(LATENT_DIM, LATENT_DIM, (1, 1), (1, 1)), nn.LeakyReLU(0.2))


def dxz_stack():
    return nn.Sequential(nn.Conv2d(2 * LATENT_DIM, 1024, (1, 1), (1, 1)), nn.LeakyReLU(0.2),
                         nn.Conv2d(1024, 1024, (1, 1), (1, 1)), nn.LeakyReLU(0.2),
                         nn.Conv2d(1024, 1, (1, 1), (1, 1)))


class SpectBase(nn.Module):
    """Family description used by the forward passes below.

    image_hw          input size;
    cat_keys          categorical attribute keys in the order their planes / embeddings are concatenated;
    cont_key          optional continuous attribute appended after the categorical ones (ESRF closest_boat);
    plane_module(k)   the Sequential(Embedding, Unflatten, Upsample, Tanh) of key k (Encoder / Discriminator);
    table(k)          the nn.Embedding of key k (Generator).
    """
    image_hw = (128, 128)
    cat_keys = ()
    cont_key = None

    @property
    def device(self):
        return next(self.parameters()).device

    # ---- hooks the subclasses provide
    def plane_module(self, k):
        raise NotImplementedError

    def table(self, k):
        raise NotImplementedError

    # ---- conv input of Encoder / Discriminator
    def _features_torch(self, X, a):
        H, W = self.image_hw
        planes = [self.plane_module(k)(a[k].argmax(dim=1)) for k in self.cat_keys]
        if self.cont_key is not None:
            c = a[self.cont_key].reshape((-1, 1))
            planes.append(c.reshape((c.size(0), 1, 1, 1)).repeat(1, 1, H, W))
        return torch.concat([X.reshape((-1, 1, H, W)), *planes], dim=1)

    def _features_hip(self, X, a):
        from ali_hip import planes
        H, W = self.image_hw
        X = X.reshape(-1, H, W).float()
        B = X.shape[0]
        idx = torch.stack([a[k].argmax(dim=1) for k in self.cat_keys], dim=1).to(torch.int32).contiguous()
        cont = a[self.cont_key].reshape(B, 1).float() if self.cont_key is not None else None
        n_log = 1 + len(self.cat_keys) + (1 if cont is not None else 0)
        x0 = planes.assemble(X, idx, cont, (n_log + 3) // 4 * 4, [self.plane_module(k)[0].weight for k in self.cat_keys])
        return x0, n_log


class SpectEncoder(SpectBase):
    def forward(self, X, a):
        if not X.is_cuda:
            return self.layers(self._features_torch(X, a))
        from ali_hip.chain import run_chain
        x0, n_log = self._features_hip(X, a)
        return run_chain(self.layers, x0, n_log).reshape(x0.shape[0], LATENT_DIM, 1, 1)


class SpectGenerator(SpectBase):
    def forward(self, z, a):
        z = z.reshape((-1, LATENT_DIM))
        feats = [z] + [a[k].float().matmul(self.table(k).weight) for k in self.cat_keys]
        if self.cont_key is not None:
            feats.append(a[self.cont_key].reshape((-1, 1)))
        if not z.is_cuda:
            return self.layers(torch.concat(feats, dim=1))
        from ali_hip.chain import run_chain
        B = z.shape[0]
        feats = [f.float() for f in feats]
        n_log = sum(f.shape[1] for f in feats)
        pad = (-n_log) % 32          # channel stride % 32 == 0 -> uniform-tap fast path of the GEMM kernel
        if pad:
            feats.append(torch.zeros(B, pad, device=z.device))
        x0 = torch.cat(feats, dim=1).reshape(B, 1, 1, n_log + pad)
        H, W = self.image_hw
        return run_chain(self.layers, x0, n_log).reshape(B, 1, H, W)


class SpectDiscriminator(SpectBase):
    def forward(self, X, z, a):
        if not X.is_cuda:
            dx = self.dx(self._features_torch(X, a))
            dz = self.dz(z.reshape((-1, LATENT_DIM, 1, 1)))
            return self.dxz(torch.concat([dx, dz], dim=1)).reshape((-1, 1))
        from ali_hip.chain import run_chain
        x0, n_log = self._features_hip(X, a)
        B = x0.shape[0]
        dx = run_chain(self.dx, x0, n_log)
        dz = run_chain(self.dz, z.reshape(B, 1, 1, LATENT_DIM).float())
        joint = torch.cat([dx.reshape(B, -1), dz.reshape(B, -1)], dim=1).reshape(B, 1, 1, -1)
        return run_chain(self.dxz, joint).reshape(-1, 1)


def data_adapter_unavailable(name, needs):
    """The dataset adapters (zip / wav readers + torchaudio spectrogram front-ends) are the step *before* the hot
    path and are not part of this package; the class names stay importable for the callers."""

    class _Unavailable:
        def __init__(self, *args, **kwargs):
            raise ImportError(f"{name} needs {needs} and the original dataset; it is outside the MI355X hot path. "
                              f"Use the reference's {name} to produce batches and feed them to train_on_stream().")
    _Unavailable.__name__ = name
    return _Unavailable


def train_on_stream(E, G, D, stream_fn, n_batches_hint=None, n_epochs=1, l_rate=1e-4, device='cpu',
                    preprocess=None, attr_keys=(), use_stepper=None, family=None, capture=True):
    """The training loop of audio_mnist.train / whalecalls.train / esrf_acoustic.train (audio_mnist.py:372-420 etc.)
    over any generator of batch dicts ``{"audio": [B,H,W], <attr>: one-hot ...}``.

    Adam(lr, betas=(0.5, 0.9)) for E+G and for D; z ~ N(0,1) sampled on the host like the reference.  On a CUDA device
    the hand-scheduled ``AliStepper`` is used (``use_stepper``; ``capture``: replay the iteration from a HIP graph per
    batch shape), otherwise the autograd ``ali_step``.
    Returns (E, G, D, optimizer_D, optimizer_E, epoch_scores)."""
    from .training_utils import ali_step
    dev = torch.device(device)
    if use_stepper is None:
        use_stepper = dev.type == "cuda"
    scores = []
    if use_stepper:
        from ali_hip.step import AliStepper
        stepper = AliStepper(E, G, D, lr=l_rate, betas=(0.5, 0.9), family=family, capture=capture)
        opt_e, opt_d = stepper.opt_eg, stepper.opt_d
    else:
        opt_e = torch.optim.Adam(list(E.parameters()) + list(G.parameters()), lr=l_rate, betas=(0.5, 0.9))
        opt_d = torch.optim.Adam(D.parameters(), lr=l_rate, betas=(0.5, 0.9))
    gan_loss = nn.BCEWithLogitsLoss()
    H, W = E.image_hw
    for _ in range(n_epochs):
        for m in (D, E, G):
            m.train()
        d_score = torch.zeros((), device=dev)
        eg_score = torch.zeros((), device=dev)
        n = 0
        for batch in stream_fn():
            images = batch["audio"].reshape((-1, 1, H, W)).float().to(dev)
            c = {k: torch.clone(batch[k]).float().to(dev) for k in attr_keys}
            if preprocess is not None:
                images = preprocess(images)
            z_mean = torch.zeros((len(images), LATENT_DIM, 1, 1)).float()
            z = torch.normal(z_mean, z_mean + 1).to(dev)
            if use_stepper:
                r = stepper.step(images, c, z)
            else:
                r = ali_step(E, G, D, opt_e, opt_d, images, c, z, gan_loss=gan_loss)
            d_score += r["dg"]
            eg_score += r["de"]
            n += 1
        scores.append((d_score.item() / max(n, 1), eg_score.item() / max(n, 1)))
        print(*scores[-1])
    return E, G, D, opt_d, opt_e, scores
