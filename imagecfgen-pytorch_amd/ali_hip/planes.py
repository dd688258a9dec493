"""Conditioning-plane assembly for Encoder / Discriminator inputs.

Replaces ``Embedding -> Unflatten(1,16,16) -> Upsample(nearest) -> Tanh``, the
``continuous_feature_map`` broadcasts and the channel concat of the reference
(mnist.py:17-18,24-29,46-55; audio_mnist.py:178-183,203-210; esrf_acoustic.py:163-170)
with one kernel that writes the NHWC, channel-padded conv input directly.
Backward ([B,H,W] plane gradients -> [n,256] tables): ali_plane_table_grad (fixed summation order).
"""
import torch

from . import ops

_MAPS = {}


def _src_index(H, W, device):
    key = (H, W, str(device))
    m = _MAPS.get(key)
    if m is None:
        hh = (torch.arange(H, device=device) * 16) // H
        ww = (torch.arange(W, device=device) * 16) // W
        m = (hh[:, None] * 16 + ww[None, :]).reshape(-1)
        _MAPS[key] = m
    return m


def plane_to_table_grad(gp, idx_col, n_rows, H, W):
    """gp [B, H*W] (gradient w.r.t. the pre-tanh plane) -> gradient of the [n_rows, 256] embedding table.
    Deterministic (no atomics): nearest up-sampling is undone by a block sum (integer factors) or by a product
    with the fixed 0/1 pixel->cell matrix, the per-class scatter by a product with the one-hot matrix."""
    B = gp.shape[0]
    if H % 16 == 0 and W % 16 == 0:
        per_sample = gp.reshape(B, 16, H // 16, 16, W // 16).sum(dim=(2, 4)).reshape(B, 256)
    else:
        key = ("S", H, W, str(gp.device))
        S = _MAPS.get(key)
        if S is None:
            S = torch.zeros(H * W, 256, device=gp.device)
            S[torch.arange(H * W, device=gp.device), _src_index(H, W, gp.device)] = 1.0
            _MAPS[key] = S
        per_sample = gp.matmul(S)
    onehot = (idx_col.reshape(B, 1) == torch.arange(n_rows, device=gp.device, dtype=idx_col.dtype).reshape(1, -1))
    return onehot.float().t().matmul(per_sample)


class PlanesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, idx, cont, cpad, *tables):
        B, H, W = X.shape
        tabs = [t.detach().contiguous() for t in tables]
        out = ops.assemble_planes(X.contiguous(), idx, tabs, cont.contiguous() if cont is not None else None, B, H, W,
                                  cpad)
        ctx.save_for_backward(idx, out)
        ctx.n_emb, ctx.n_cont = len(tables), (0 if cont is None else cont.shape[1])
        ctx.table_rows = [t.shape[0] for t in tables]
        return out

    @staticmethod
    def backward(ctx, g):
        idx, out = ctx.saved_tensors
        B, H, W, _ = g.shape
        gX = g[..., 0].contiguous() if ctx.needs_input_grad[0] else None
        gcont = None
        if ctx.n_cont and ctx.needs_input_grad[2]:
            gcont = g[..., 1 + ctx.n_emb:1 + ctx.n_emb + ctx.n_cont].sum(dim=(1, 2))
        gtabs = []
        for j in range(ctx.n_emb):
            if not ctx.needs_input_grad[4 + j]:
                gtabs.append(None)
                continue
            gtabs.append(ops.plane_table_grad(g.contiguous(), 1 + j, out, 1 + j, idx, j, ctx.table_rows[j]))
        return (gX, None, gcont, None) + tuple(gtabs)


def assemble(X, idx, cont, cpad, tables):
    """X [B,H,W]; idx int32 [B,n_emb]; cont [B,n_cont] or None; tables list of [n,256] params."""
    return PlanesFn.apply(X, idx, cont, cpad, *tables)
