"""CPU oracle for the ALI/BiGAN training path -- TEST INFRASTRUCTURE ONLY.

This file is a plain torch-CPU fp32 restatement of the reference's
``image_scms`` Encoder / Generator / Discriminator stacks and of the inline
ALI training iteration.  It exists so that the hand-written HIP path can be
checked against something that is itself pinned to the reference:

* pinned by ``tests/golden/*.npz`` which ``oracle/gen_golden.py`` produced by
  importing the real reference in the build container (the reference never
  travels to the GPU box);
* only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
  ``cpu_baseline`` leg may import it.  Nothing under ``imagecfgen-pytorch_amd/``
  imports it -- the product path is the HIP library and fails loudly without it.

Every class cites the reference file:line it restates (paths relative to the
reference checkout).  The layer stacks are table driven so that constructor
order (and therefore RNG consumption) matches the reference exactly.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

LATENT = 512  # mnist.py:12, audio_mnist.py:20, whalecalls.py:20, esrf_acoustic.py:21


# --------------------------------------------------------------------------
# dropout with an explicit mask tape (recipe: SURVEY.md K7)
# --------------------------------------------------------------------------
class MaskTape:
    """Records (or replays) the Dropout2d masks of Discriminator forwards.

    ``Dropout2d(p)(x)`` on CPU is ``x * empty(N, C, 1, 1).bernoulli_(1-p) / (1-p)``
    with the global CPU generator; drawing masks that way keeps the oracle in
    RNG lock-step with the reference (verified bit-for-bit by the trajectory
    fixture) while exposing the masks so they can be uploaded to the GPU path.
    """

    def __init__(self, replay: Optional[Sequence[torch.Tensor]] = None):
        self.masks: List[torch.Tensor] = []
        self._replay = list(replay) if replay is not None else None
        self._pos = 0

    def draw(self, n: int, c: int, p: float) -> torch.Tensor:
        if self._replay is not None:
            m = self._replay[self._pos]
            self._pos += 1
            assert m.shape == (n, c), (m.shape, (n, c))
        else:
            m = torch.empty(n, c, 1, 1).bernoulli_(1 - p).div_(1 - p).reshape(n, c)
        self.masks.append(m)
        return m


_ACTIVE_TAPE: List[Optional[MaskTape]] = [None]


class use_tape:
    def __init__(self, tape: Optional[MaskTape]):
        self.tape = tape

    def __enter__(self):
        self.prev = _ACTIVE_TAPE[0]
        _ACTIVE_TAPE[0] = self.tape
        return self.tape

    def __exit__(self, *exc):
        _ACTIVE_TAPE[0] = self.prev


class TapedDropout2d(nn.Module):
    """nn.Dropout2d stand-in (mnist.py:99-134) drawing its mask through the tape."""

    def __init__(self, p: float):
        super().__init__()
        self.p = p

    def forward(self, x):
        if not self.training:
            return x
        tape = _ACTIVE_TAPE[0]
        if tape is None:
            tape = MaskTape()
        m = tape.draw(x.shape[0], x.shape[1], self.p)
        return x * m.reshape(x.shape[0], x.shape[1], 1, 1).to(x.device)


# --------------------------------------------------------------------------
# layer-table helpers
# --------------------------------------------------------------------------
def _build(rows) -> nn.Sequential:
    """rows: tuples ('conv',ci,co,k,s,p) ('convT',ci,co,k,s,p,op) ('lrelu',a)
    ('tanh',) ('bn',c) ('drop',p) ('linear',i,o) ('unflat',c,h,w)."""
    mods = []
    for r in rows:
        kind = r[0]
        if kind == "conv":
            _, ci, co, k, s, p = r
            mods.append(nn.Conv2d(ci, co, (k, k), (s, s), p))
        elif kind == "convT":
            _, ci, co, k, s, p, op = r
            mods.append(nn.ConvTranspose2d(ci, co, (k, k), (s, s), p, op))
        elif kind == "lrelu":
            mods.append(nn.LeakyReLU(r[1]))
        elif kind == "tanh":
            mods.append(nn.Tanh())
        elif kind == "bn":
            mods.append(nn.BatchNorm2d(r[1]))
        elif kind == "drop":
            mods.append(TapedDropout2d(r[1]))
        elif kind == "linear":
            mods.append(nn.Linear(r[1], r[2]))
        elif kind == "unflat":
            mods.append(nn.Unflatten(1, (r[1], r[2], r[3])))
        else:
            raise ValueError(kind)
    return nn.Sequential(*mods)


def _plane_embed(n_classes: int, size=None, scale=None) -> nn.Sequential:
    """Embedding -> [1,16,16] -> nearest upsample -> tanh (mnist.py:24-29,
    audio_mnist.py:178-183, whalecalls.py:235-240, esrf_acoustic.py:138-143)."""
    up = nn.Upsample(size=size) if size is not None else nn.Upsample(scale_factor=scale)
    return nn.Sequential(nn.Embedding(n_classes, 256), nn.Unflatten(1, (1, 16, 16)), up, nn.Tanh())


def _plane(c: torch.Tensor, hw) -> torch.Tensor:
    """continuous_feature_map, mnist.py:17-18 / esrf_acoustic.py:13-14."""
    return c.reshape(c.size(0), 1, 1, 1).repeat(1, 1, *hw)


# --------------------------------------------------------------------------
# MorphoMNIST (image_scms/mnist.py)
# --------------------------------------------------------------------------
class MnistEncoder(nn.Module):
    """mnist.py:21-56."""

    def __init__(self):
        super().__init__()
        self.digit_embedding = _plane_embed(10, size=(28, 28))
        self.layers = _build([
            ("conv", 5, 64, 3, 2, 1), ("lrelu", 0.2),
            ("conv", 64, 128, 4, 2, 1), ("lrelu", 0.2),
            ("conv", 128, 256, 4, 2, 1), ("lrelu", 0.2),
            ("conv", 256, 512, 4, 2, 1), ("lrelu", 0.2),
            ("conv", 512, LATENT, 1, 2, 0),
        ])

    def forward(self, X, c):
        cont = [_plane(c[k], (28, 28)) for k in sorted(c) if k != "digit"]
        dig = self.digit_embedding(c["digit"].argmax(1))
        return self.layers(torch.cat([X, dig] + cont, dim=1))


class MnistGenerator(nn.Module):
    """mnist.py:59-86."""

    def __init__(self):
        super().__init__()
        self.digit_embedding = nn.Embedding(10, 256)
        self.layers = _build([
            ("convT", LATENT + 256 + 3, 512, 3, 1, 0, 0), ("lrelu", 0.2),
            ("convT", 512, 256, 3, 2, 0, 0), ("lrelu", 0.2),
            ("convT", 256, 128, 3, 2, 1, 0), ("lrelu", 0.2),
            ("convT", 128, 64, 3, 2, 1, 0), ("lrelu", 0.2),
            ("convT", 64, 1, 4, 1, 0, 0), ("tanh",),
        ])

    def forward(self, z, c):
        dig = c["digit"].matmul(self.digit_embedding.weight).reshape(-1, 256, 1, 1)
        cont = [_plane(c[k], (1, 1)) for k in sorted(c) if k != "digit"]
        return self.layers(torch.cat([z, dig] + cont, dim=1))


class MnistDiscriminator(nn.Module):
    """mnist.py:89-154 (ctor order: digit_embedding, dz, dx, dxz)."""

    def __init__(self):
        super().__init__()
        self.digit_embedding = _plane_embed(10, size=(28, 28))
        self.dz = _build([
            ("drop", 0.2), ("conv", 512, 512, 1, 1, 0), ("lrelu", 0.1),
            ("drop", 0.5), ("conv", 512, 512, 1, 1, 0), ("lrelu", 0.1),
        ])
        self.dx = _build([
            ("drop", 0.2), ("conv", 5, 32, 5, 1, 0), ("lrelu", 0.1), ("drop", 0.2), ("bn", 32),
            ("conv", 32, 64, 4, 2, 0), ("lrelu", 0.1), ("bn", 64), ("drop", 0.5),
            ("conv", 64, 128, 4, 1, 0), ("lrelu", 0.1), ("bn", 128), ("drop", 0.5),
            ("conv", 128, 256, 4, 2, 0), ("lrelu", 0.1), ("bn", 256), ("drop", 0.5),
            ("conv", 256, 512, 3, 1, 0), ("lrelu", 0.1),
        ])
        self.dxz = _build([
            ("drop", 0.2), ("conv", 1024, 1024, 1, 1, 0), ("lrelu", 0.1),
            ("drop", 0.2), ("conv", 1024, 1024, 1, 1, 0), ("lrelu", 0.1),
            ("drop", 0.2), ("conv", 1024, 1, 1, 1, 0),
        ])

    def forward(self, X, z, c):
        cont = [_plane(c[k], (28, 28)) for k in sorted(c) if k != "digit"]
        dig = self.digit_embedding(c["digit"].argmax(1))
        dx = self.dx(torch.cat([X, dig] + cont, dim=1))   # dx masks first ...
        dz = self.dz(z)                                   # ... then dz, then dxz (mnist.py:152-154)
        return self.dxz(torch.cat([dx, dz], dim=1)).reshape(-1, 1)


# --------------------------------------------------------------------------
# spectrogram families (audio_mnist.py / whalecalls.py / esrf_acoustic.py)
# --------------------------------------------------------------------------
def _enc_rows(cin, widths, d):
    rows = []
    c_prev = cin
    for i, w in enumerate(widths):
        co = LATENT if w is None else w * d
        rows.append(("conv", c_prev, co, 5, 2, 1))
        if i + 1 < len(widths):
            rows.append(("lrelu", 0.2))
        c_prev = co
    return rows


def _gen_rows(in_features, widths, d):
    rows = [("linear", in_features, 256 * d), ("unflat", 16 * d, 4, 4), ("lrelu", 0.2)]
    c_prev = 16 * d
    for w in widths:
        co = 1 if w is None else w * d
        rows.append(("convT", c_prev, co, 5, 2, 2, 1))
        rows.append(("tanh",) if w is None else ("lrelu", 0.2))
        c_prev = co
    return rows


_DZ = [("conv", LATENT, LATENT, 1, 1, 0), ("lrelu", 0.2), ("conv", LATENT, LATENT, 1, 1, 0), ("lrelu", 0.2)]
_DXZ = [("conv", 2 * LATENT, 1024, 1, 1, 0), ("lrelu", 0.2), ("conv", 1024, 1024, 1, 1, 0), ("lrelu", 0.2),
        ("conv", 1024, 1, 1, 1, 0)]

FAMILIES = {
    # image side, categorical attrs (sorted use order), per-family stacks
    "audio": dict(hw=(128, 128), scale=8,
                  attrs={"country_of_origin": 13, "native_speaker": 2, "accent": 15, "digit": 10, "age": 5,
                         "gender": 2},                                # audio_mnist.py:23-30
                  enc=[1, 2, 4, 8, 16, None],                         # audio_mnist.py:186-198
                  dis=[1, 2, 4, 8, 16, None],                         # audio_mnist.py:278-296
                  gen=[8, 4, 2, 1, None]),                            # audio_mnist.py:224-243
    "whale": dict(hw=(256, 256), scale=16, attrs={"call_type": 3},    # whalecalls.py:14-19 (time/path skipped)
                  enc=[1, 2, 4, 8, 16, 16, None],                     # whalecalls.py:244-258
                  dis=[1, 2, 2, 4, 8, 16, None],                      # whalecalls.py:345-366
                  gen=[16, 8, 4, 2, 1, None]),                        # whalecalls.py:286-309
    "esrf": dict(hw=(512, 512), scale=32, attrs={"has_boat": 2},      # esrf_acoustic.py:17-20
                 enc=[1, 2, 4, 8, 16, 32, 64, None],                  # esrf_acoustic.py:144-160
                 dis=[1, 2, 4, 8, 16, 32, 64, None],                  # esrf_acoustic.py:218-234
                 gen=[16, 8, 4, 2, 1, 1, None]),                      # esrf_acoustic.py:181-199
}


class SpectEncoder(nn.Module):
    """audio_mnist.py:173-210, whalecalls.py:230-271, esrf_acoustic.py:134-170."""

    def __init__(self, family: str, d: int = 64):
        super().__init__()
        f = FAMILIES[family]
        self.family, self.f = family, f
        n_planes = len(f["attrs"]) + (1 if family == "esrf" else 0)
        if family == "esrf":
            self.has_boat_embedding = _plane_embed(2, scale=f["scale"])
        else:
            self.embedding_dict = nn.ModuleDict({k: _plane_embed(v, scale=f["scale"]) for k, v in f["attrs"].items()})
        self.layers = _build(_enc_rows(1 + n_planes, f["enc"], d))

    def planes(self, a):
        if self.family == "esrf":
            return [self.has_boat_embedding(a["has_boat"].argmax(1)),
                    _plane(a["closest_boat"].reshape(-1, 1), self.f["hw"])]
        return [self.embedding_dict[k](a[k].argmax(dim=1)) for k in sorted(self.f["attrs"])]

    def forward(self, X, a):
        X = X.reshape(-1, 1, *self.f["hw"])
        return self.layers(torch.cat([X, *self.planes(a)], dim=1))


class SpectGenerator(nn.Module):
    """audio_mnist.py:213-256, whalecalls.py:274-321, esrf_acoustic.py:173-205."""

    def __init__(self, family: str, d: int = 64):
        super().__init__()
        f = FAMILIES[family]
        self.family, self.f = family, f
        if family == "esrf":
            self.has_boat_embedding = nn.Embedding(2, 256)
            in_features = LATENT + 257
        else:
            self.embedding_dict = nn.ModuleDict({k: nn.Embedding(v, 256) for k, v in f["attrs"].items()})
            in_features = LATENT + 256 * len(f["attrs"])
        self.layers = _build(_gen_rows(in_features, f["gen"], d))

    def forward(self, z, a):
        z = z.reshape(-1, LATENT)
        if self.family == "esrf":
            feats = [a["has_boat"].matmul(self.has_boat_embedding.weight), a["closest_boat"].reshape(-1, 1)]
        else:
            feats = [a[k].float().matmul(self.embedding_dict[k].weight) for k in sorted(self.f["attrs"])]
        return self.layers(torch.cat([z, *feats], dim=1))


class SpectDiscriminator(nn.Module):
    """audio_mnist.py:259-318 (ctor order embedding, dz, dx, dxz), whalecalls.py:324-387 (same),
    esrf_acoustic.py:208-260 (ctor order embedding, dx, dz, dxz)."""

    def __init__(self, family: str, d: int = 64):
        super().__init__()
        f = FAMILIES[family]
        self.family, self.f = family, f
        n_planes = len(f["attrs"]) + (1 if family == "esrf" else 0)
        if family == "esrf":
            self.has_boat_embedding = _plane_embed(2, scale=f["scale"])
            self.dx = _build(_enc_rows(1 + n_planes, f["dis"], d))
            self.dz = _build(_DZ)
        else:
            self.embedding_dict = nn.ModuleDict({k: _plane_embed(v, scale=f["scale"]) for k, v in f["attrs"].items()})
            self.dz = _build(_DZ)
            self.dx = _build(_enc_rows(1 + n_planes, f["dis"], d))
        self.dxz = _build(_DXZ)

    planes = SpectEncoder.planes

    def forward(self, X, z, a):
        X = X.reshape(-1, 1, *self.f["hw"])
        z = z.reshape(-1, LATENT, 1, 1)
        dx = self.dx(torch.cat([X, *self.planes(a)], dim=1))
        dz = self.dz(z)
        return self.dxz(torch.cat([dx, dz], dim=1)).reshape(-1, 1)


def init_weights(layer, std):
    """training_utils.py:114-119 (std .01) / audio_mnist.py:33-38 etc. (std .001):
    only modules whose class name starts with 'Conv'."""
    if layer.__class__.__name__.startswith("Conv"):
        torch.nn.init.normal_(layer.weight, mean=0, std=std)
        if layer.bias is not None:
            torch.nn.init.constant_(layer.bias, 0)


def build_models(family: str = "mnist", d: int = 64, std: Optional[float] = None):
    """E, G, D in the reference's construction + init order (mnist.py:168-174)."""
    if family == "mnist":
        E, G, D = MnistEncoder(), MnistGenerator(), MnistDiscriminator()
        std = 0.01 if std is None else std
    else:
        E, G, D = SpectEncoder(family, d), SpectGenerator(family, d), SpectDiscriminator(family, d)
        std = 0.001 if std is None else std
    for m in (E, G, D):
        m.apply(lambda l: init_weights(l, std))
    return E, G, D


def build_optimizers(E, G, D, family="mnist", lr=1e-4):
    """mnist.py:176-179 betas (0.5, 0.999); audio_mnist.py:336-339 etc. betas (0.5, 0.9)."""
    betas = (0.5, 0.999) if family == "mnist" else (0.5, 0.9)
    opt_e = torch.optim.Adam(list(E.parameters()) + list(G.parameters()), lr=lr, betas=betas)
    opt_d = torch.optim.Adam(D.parameters(), lr=lr, betas=betas)
    return opt_e, opt_d


# --------------------------------------------------------------------------
# the ALI iteration (mnist.py:224-248 and its three copies)
# --------------------------------------------------------------------------
def ali_step(E, G, D, opt_e, opt_d, images, c, z, do_eg=True, tape: Optional[MaskTape] = None):
    """One reference iteration, exactly as executed (wasted work included).

    Returns dict(loss_eg, loss_d_real, loss_d_fake, dg, de) of python floats.
    """
    bce = nn.BCEWithLogitsLoss()
    B = images.size(0)
    valid = torch.ones(B, 1, device=images.device)
    fake = torch.zeros(B, 1, device=images.device)
    out = {}
    with use_tape(tape):
        if do_eg:                                           # mnist.py:224-230
            opt_e.zero_grad()
            d_valid = D(images, E(images, c), c)
            d_fake = D(G(z, c), z, c)
            loss_eg = (bce(d_valid, fake) + bce(d_fake, valid)) / 2
            loss_eg.backward()
            opt_e.step()
            out["loss_eg"] = loss_eg.item()
        opt_d.zero_grad()                                   # mnist.py:232-236
        loss_dr = bce(D(images, E(images, c), c), valid)
        loss_dr.backward()
        opt_d.step()
        opt_d.zero_grad()                                   # mnist.py:237-241
        loss_df = bce(D(G(z, c), z, c), fake)
        loss_df.backward()
        opt_d.step()
        gz = G(z, c).detach()                               # mnist.py:243-248
        ex = E(images, c).detach()
        out["dg"] = D(gz, z, c).sigmoid().mean().item()
        out["de"] = D(images, ex, c).sigmoid().mean().item()
    out["loss_d_real"] = loss_dr.item()
    out["loss_d_fake"] = loss_df.item()
    return out


def mnist_scale_batch(images_u8, attrs, attr_stats):
    """mnist.py:204-209: pixel and attribute scaling of one batch."""
    images = 2 * images_u8.reshape(-1, 1, 28, 28).float() / 255 - 1
    c = {k: 2 * (attrs[k] - attr_stats[k][0]) / (attr_stats[k][1] - attr_stats[k][0]) - 1 for k in attr_stats}
    c["digit"] = attrs["digit"]
    return images, c


def mnist_train(x_train, a_train, n_epochs=1, l_rate=1e-4, batch_size=64, d_updates_per_g_update=1,
                record=None):
    """Restatement of mnist.train (mnist.py:157-299) without the plotting branch.

    RNG order: ctors E,G,D -> init_weights E,G,D -> np permutation -> per iteration
    randn z -> dropout masks in call order.  ``record`` (a list) receives per-step
    dicts with losses, z and the mask tape.
    """
    E, G, D = build_models("mnist")
    opt_e, opt_d = build_optimizers(E, G, D, "mnist", l_rate)
    scores = []
    for _ in range(n_epochs):
        D.train(), E.train(), G.train()
        perm = np.random.permutation(len(x_train))
        xs = x_train[perm]
        as_ = {k: v[perm] for k, v in a_train.items()}
        stats = {k: (v.min(dim=0).values, v.max(dim=0).values) for k, v in a_train.items() if k != "digit"}
        d_score = eg_score = 0.0
        nb = 0
        for i, lo in enumerate(range(0, len(xs), batch_size)):
            images, c = mnist_scale_batch(xs[lo:lo + batch_size],
                                          {k: v[lo:lo + batch_size] for k, v in as_.items()}, stats)
            zm = torch.zeros(len(images), LATENT, 1, 1)
            z = torch.normal(zm, zm + 1)                     # mnist.py:220-221
            tape = MaskTape()
            r = ali_step(E, G, D, opt_e, opt_d, images, c, z, do_eg=(i % d_updates_per_g_update == 0), tape=tape)
            d_score += r["dg"]
            eg_score += r["de"]
            nb += 1
            if record is not None:
                r.update(z=z, masks=tape.masks, images=images, c=c)
                record.append(r)
        scores.append((d_score / nb, eg_score / nb))
    return E, G, D, opt_d, opt_e, scores


# --------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------
def synth_morphomnist(n: int, seed: int = 1):
    """MorphoMNIST-shaped synthetic data following the generating SCM of
    create_train_dataset.py:23-46: x [n,28,28] float 0..255, attrs dict."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(0, 256, (n, 28, 28), generator=g).float()
    x = x * (torch.rand(n, 28, 28, generator=g) > 0.8)        # sparse strokes: 80 % zeros
    digit = torch.nn.functional.one_hot(torch.randint(0, 10, (n,), generator=g), 10).float()
    gam = torch.distributions.Gamma(10.0, 5.0)
    torch.manual_seed(seed + 12345)
    thickness = gam.sample((n, 1)) + 0.5
    intensity = 191 * torch.sigmoid(0.5 * torch.randn(n, 1, generator=g) + 2 * thickness - 5) + 64
    slant = np.pi * 0.1 * torch.randn(n, 1, generator=g)
    return x, {"digit": digit, "thickness": thickness, "intensity": intensity, "slant": slant}


def synth_spect_batch(family: str, b: int, seed: int = 1):
    """images ~ clip(N(0,1),-3,3)/3 (what spect_to_img yields, audio_mnist.py:361-363), one-hot attrs."""
    f = FAMILIES[family]
    g = torch.Generator().manual_seed(seed)
    images = torch.clip(torch.randn(b, 1, *f["hw"], generator=g), -3, 3) / 3
    a = {k: torch.nn.functional.one_hot(torch.randint(0, v, (b,), generator=g), v).float()
         for k, v in f["attrs"].items()}
    if family == "esrf":
        a["closest_boat"] = torch.rand(b, 1, generator=g) * 2 - 1
    z = torch.randn(b, LATENT, 1, 1, generator=g)
    return images, a, z


def tensor_digest(t: torch.Tensor) -> str:
    import hashlib
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


def weights_digest(*modules) -> str:
    import hashlib
    h = hashlib.sha256()
    for m in modules:
        for _, v in sorted(m.state_dict().items()):
            h.update(v.detach().contiguous().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


# --------------------------------------------------------------------------
# helpers shared by gen_golden.py and the tests
# --------------------------------------------------------------------------
def rescale_for_test_(module: nn.Module, nominal_std: float, gain: float = 1.3, bias_seed: int = 7):
    """Make activations O(1) (reference inits leave every loss at ln 2, SURVEY.md 7 'Hard parts').

    Conv / ConvT weights drawn as N(0, nominal_std) are rescaled in place to
    std = gain / sqrt(effective fan-in); biases get seeded N(0, 0.05) values.
    Applied identically (same parameter order) to reference and oracle modules.
    """
    g = torch.Generator().manual_seed(bias_seed)
    with torch.no_grad():
        for m in module.modules():
            name = m.__class__.__name__
            if name.startswith("ConvTranspose"):
                fan = m.in_channels * m.kernel_size[0] * m.kernel_size[1] / (m.stride[0] * m.stride[1])
            elif name.startswith("Conv"):
                fan = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
            else:
                continue
            m.weight.mul_(gain / (nominal_std * fan ** 0.5))
            if m.bias is not None:
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
    return module


def tensor_stats(t: torch.Tensor):
    """(sum, abs-sum, first 4 values) in float64 -- compact pins for large tensors."""
    t = t.detach().double().reshape(-1)
    head = t[:4].tolist() + [0.0] * (4 - min(4, t.numel()))
    return [t.sum().item(), t.abs().sum().item()] + head[:4]
