"""Drop-in for the reference's ``image_scms/training_utils.py`` (same names, same
argument meaning) plus the ALI iteration as a function: the reference keeps the loop
body inline in every ``image_scms/<dataset>.py::train`` (mnist.py:202-248); here it
is ``ali_step`` so callers and tests can drive one iteration.
"""
import torch
import torch.nn as nn

try:  # only rec_loss(metric='ssim') needs it (reference training_utils.py:3,91)
    from pytorch_msssim import ssim
except Exception:  # pragma: no cover
    def ssim(*args, **kwargs):
        raise ImportError("pytorch_msssim is not installed; use metric='mse'")


def batchify(*tensors, batch_size=128, device='cpu'):
    """Yield tuples of contiguous slices (reference training_utils.py:6-13)."""
    n = min(len(t) for t in tensors)
    for lo in range(0, n, batch_size):
        yield tuple(t[lo:lo + batch_size] for t in tensors)


def batchify_dict(tensors: dict, batch_size=128, device='cpu'):
    """Dict flavour of batchify (reference training_utils.py:16-27)."""
    n = min(len(v) for v in tensors.values())
    for lo in range(0, n, batch_size):
        yield {k: v[lo:lo + batch_size] for k, v in tensors.items()}


def binarized_attribute_channel(image, attributes, device='cpu'):
    """One constant 0/1 plane per class, 1 on the arg-max class (reference :30-35)."""
    n, _, w, h = image.shape
    planes = torch.zeros((n, attributes.shape[1], w, h), dtype=torch.float32, device=device)
    planes[torch.arange(n), attributes.argmax(dim=1)] = 1.0
    return planes


def attributes_image(image, attributes, device='cpu'):
    """Append a plane carrying the attribute vector in its middle columns (reference :38-46)."""
    n, _, w, h = image.shape
    k = attributes.shape[1]
    plane = torch.zeros((n, 1, w, h), dtype=torch.float32, device=device)
    lo = h // 2 - k // 2 - k % 2
    plane[:, :, :, lo:h // 2 + k // 2] = attributes.reshape(n, 1, 1, k)
    return torch.cat([image.to(device), plane], dim=1)


def log_loss(score_0, score_1, eps=1e-6):
    return -(torch.log(score_1 + eps) + torch.log(1 - score_0 + eps)).mean()


class AdversariallyLearnedInference(nn.Module):
    """API-parity wrapper (reference :54-111); the image_scms train loops never instantiate it."""

    def __init__(self, encoder, decoder, discriminator):
        super().__init__()
        self.encoder, self.decoder, self.discriminator = encoder, decoder, discriminator

    def __call__(self, x, z, a=None, add_noise=False, noise_scale=0.1):
        extra = () if a is None else (a,)
        ex = self.encoder(x, *extra)
        gz = self.decoder(z, *extra)
        x_in = x
        if add_noise:
            dev = next(self.encoder.parameters()).device
            x_in = x + torch.normal(0, noise_scale, x.shape).to(dev)
        return self.discriminator(gz, z, *extra), self.discriminator(x_in, ex, *extra)

    def discriminator_loss(self, x, z, a=None, eps=1e-6, **kw):
        dg, de = self(x, z, a=a, **kw)
        return log_loss(dg, de, eps)

    def generator_loss(self, x, z, a=None, eps=1e-6, **kw):
        dg, de = self(x, z, a=a, **kw)
        return log_loss(de, dg, eps)

    def rec_loss(self, x, z=None, a=None, metric='ssim'):
        if metric not in ('mse', 'ssim'):
            raise ValueError(f'Invalid metric {metric}')
        extra = () if a is None else (a,)
        if z is None:
            z = self.encoder(x, *extra)
        rec = self.decoder(z, *extra)
        if metric == 'mse':
            return torch.square(x - rec).mean()
        return 1 - ssim(x, rec, data_range=1.0, size_average=True)


def init_weights(layer, std=0.01):
    """N(0, std) on every module whose class name starts with 'Conv', zero bias (reference :114-119)."""
    if layer.__class__.__name__.startswith('Conv'):
        torch.nn.init.normal_(layer.weight, mean=0, std=std)
        if layer.bias is not None:
            torch.nn.init.constant_(layer.bias, 0)


class LambdaLayer(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, *args, **kwargs):
        return self.fn(*args, **kwargs)


def compute_gradient_penalty(disc, interpolates):
    """WGAN-GP penalty (reference :131-147); needs double backward, i.e. a torch-native ``disc``."""
    interpolates = interpolates.requires_grad_(True)
    d_out = disc(interpolates)
    grads, = torch.autograd.grad(outputs=d_out, inputs=interpolates, grad_outputs=torch.ones_like(d_out),
                                 create_graph=True, retain_graph=True, only_inputs=True)
    grads = grads.view(grads.size(0), -1)
    return ((grads.norm(2, dim=1) - 1) ** 2).mean()


def wgan_loss_it(disc, x_real, x_fake, penalty_weight=10.0):
    assert x_real.shape[0] == x_fake.shape[0], "batch size must be constant"
    base = disc(x_fake) - disc(x_real)
    eps = torch.rand((x_real.shape[0],))
    x_rand = eps * x_real + (1 - eps) * x_fake
    return base + penalty_weight * compute_gradient_penalty(disc, x_rand)


# ----------------------------------------------------------------------------
# new: the ALI iteration as a function (the reference inlines it, mnist.py:224-248)
# ----------------------------------------------------------------------------
def _group_params(opt):
    return [p for g in opt.param_groups for p in g["params"]]


def ali_step(E, G, D, optimizer_E, optimizer_D, images, c, z, do_eg=True, gan_loss=None, grad_sync=None):
    """One ALI/BiGAN iteration exactly as the reference executes it: E+G update,
    D update on (x, E(x)), D update on (G(z), z), then the two diagnostic scores.
    Works for any device; on CUDA the modules dispatch to the HIP kernels.
    ``grad_sync(params)`` (e.g. ``ali_hip.dp.GradSync``) is called after every backward, before the optimiser
    step, to average gradients across data-parallel replicas.
    Returns dict(loss_eg, loss_d_real, loss_d_fake, dg, de) of 0-d tensors (no host sync)."""
    gan_loss = gan_loss or nn.BCEWithLogitsLoss()
    n = images.size(0)
    valid = torch.ones(n, 1, device=images.device)
    fake = torch.zeros(n, 1, device=images.device)
    out = {}
    if do_eg:
        optimizer_E.zero_grad()
        d_valid = D(images, E(images, c), c)
        d_fake = D(G(z, c), z, c)
        loss_eg = (gan_loss(d_valid, fake) + gan_loss(d_fake, valid)) / 2
        loss_eg.backward()
        if grad_sync is not None:
            grad_sync(_group_params(optimizer_E))
        optimizer_E.step()
        out["loss_eg"] = loss_eg.detach()
    optimizer_D.zero_grad()
    loss_dr = gan_loss(D(images, E(images, c), c), valid)
    loss_dr.backward()
    if grad_sync is not None:
        grad_sync(_group_params(optimizer_D))
    optimizer_D.step()
    optimizer_D.zero_grad()
    loss_df = gan_loss(D(G(z, c), z, c), fake)
    loss_df.backward()
    if grad_sync is not None:
        grad_sync(_group_params(optimizer_D))
    optimizer_D.step()
    gz = G(z, c).detach()
    ex = E(images, c).detach()
    out["dg"] = D(gz, z, c).sigmoid().mean().detach()
    out["de"] = D(images, ex, c).sigmoid().mean().detach()
    out["loss_d_real"] = loss_dr.detach()
    out["loss_d_fake"] = loss_df.detach()
    return out
