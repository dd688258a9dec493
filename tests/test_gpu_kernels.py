"""GPU parity of the individual HIP kernels (through the C ABI) against torch-CPU fp32/fp64
references of the same op.  Shapes: every distinct layer of the MorphoMNIST stacks
plus the 5x5 stride-2 family of the spectrogram models and ragged / degenerate cases."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL = 2e-4  # fp32 MFMA = fp32 fma chain; only the summation order differs from the CPU reference


def _ops():
    from ali_hip import ops
    return ops


# Tile shapes the GEMM kernels are instantiated for and pick by grid depth / precision (csrc/gconv.hip: pick_tile,
# csrc/wgrad.hip: wgrad_tile).  The small oracle-sized cases below always resolve to 64x64 (128x32 for narrow outputs)
# on their own, while the spectrogram benches run almost entirely on the larger ones: every kernel test therefore also
# runs with each tile FORCED (developer knobs ALI_BM / ALI_BN for the forward / data-gradient kernel and ALI_WBM /
# ALI_WBN for the weight-gradient kernel, re-read through ops.tuning) against the same torch-CPU references.
FORCED_TILES = {"auto": None, "g64x128_w128x32": (64, 128, 128, 32), "g128x64_w128x64": (128, 64, 128, 64),
                "g128x128_w128x128": (128, 128, 128, 128), "g64x64_w64x64": (64, 64, 64, 64)}


@pytest.fixture(params=list(FORCED_TILES), ids=list(FORCED_TILES))
def forced_tile(request):
    cfg = FORCED_TILES[request.param]
    if cfg is None:
        yield None
        return
    with _ops().tuning(ALI_BM=cfg[0], ALI_BN=cfg[1], ALI_WBM=cfg[2], ALI_WBN=cfg[3]):
        yield cfg


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def close(got, ref, rtol=RTOL, what=""):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def pack_conv_fwd(ops, w, cpad):
    K, C, R, S = w.shape
    dst = torch.empty(K, R * S, cpad, device="cuda")
    return ops.pack_weights(w.cuda().contiguous(), dst, K, R * S, C, cpad, C * R * S, 1, R * S)


def pack_conv_dgrad(ops, w, cpad):
    K, C, R, S = w.shape
    dst = torch.zeros(cpad, R * S, K, device="cuda")
    ops.pack_weights(w.cuda().contiguous(), dst, C, R * S, K, K, R * S, 1, C * R * S)
    return dst


CONV_CASES = [
    # B, C, H, K, R, stride, pad      (mnist.py:31-39, 100-135)
    (6, 5, 28, 64, 3, 2, 1), (6, 64, 14, 128, 4, 2, 1), (6, 128, 7, 256, 4, 2, 1), (6, 256, 3, 512, 4, 2, 1),
    (6, 512, 1, 512, 1, 2, 0), (6, 5, 28, 32, 5, 1, 0), (6, 32, 24, 64, 4, 2, 0), (6, 64, 11, 128, 4, 1, 0),
    (6, 128, 8, 256, 4, 2, 0), (6, 256, 3, 512, 3, 1, 0), (130, 512, 1, 512, 1, 1, 0), (130, 1024, 1, 1024, 1, 1, 0),
    (130, 1024, 1, 1, 1, 1, 0),
    # 5x5 stride-2 pad-1 family (audio_mnist.py:186-198), odd maps 63 -> 31 -> 15 -> 7 -> 3 -> 1
    (2, 7, 32, 16, 5, 2, 1), (3, 16, 15, 32, 5, 2, 1), (3, 32, 7, 64, 5, 2, 1), (5, 64, 3, 128, 5, 2, 1),
    (1, 3, 37, 8, 5, 2, 1),
    (512, 256, 3, 512, 4, 2, 1),   # split-K path at the bench batch
    (70, 5, 28, 32, 5, 1, 0), (64, 7, 20, 32, 3, 1, 0),   # per-image first-layer kernel (B >= 64, 8 -> 32 channels)
    (64, 5, 9, 32, 5, 1, 0), (130, 8, 12, 32, 5, 1, 0),
    # first layers of the spectrogram stacks (3 / 7 planes -> 64 channels, 5x5 stride 2 pad 1): the row-walking kernel
    (3, 3, 67, 64, 5, 2, 1), (2, 7, 66, 64, 5, 2, 1), (5, 4, 131, 64, 5, 2, 1), (2, 8, 70, 64, 5, 2, 2),
]


@pytest.mark.parametrize("B,C,H,K,R,stride,pad", CONV_CASES)
def test_conv2d_fwd_bwd(B, C, H, K, R, stride, pad, forced_tile):
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + C + K)
    x = torch.randn(B, C, H, H, generator=g)
    w = torch.randn(K, C, R, R, generator=g) / (C * R * R) ** 0.5
    b = torch.randn(K, generator=g)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = F.leaky_relu(F.conv2d(xr, wr, br, stride=stride, padding=pad), 0.2)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    P = yr.shape[2]
    cpad = (C + 3) // 4 * 4
    xh = torch.zeros(B, H, H, cpad, device="cuda")
    xh[..., :C] = nhwc(x).cuda()
    geom = ops.geom(B, H, H, cpad, P, P, K, R, R, stride, pad)
    y = torch.empty(B, P, P, K, device="cuda")
    ops.conv_fwd(geom, xh, pack_conv_fwd(ops, w, cpad), y, ops.epilogue(bias=b.cuda(), act=ops.ACT_LEAKY, slope=0.2))
    close(nchw(y), yr, what="fwd")
    if cpad != C:   # the live-channel hint (AliEpilogue.in_ch_live): padding channels are skipped where a kernel can
        bc = b.cuda()               # (the epilogue holds raw pointers: the tensor must outlive the launch)
        ep = ops.epilogue(bias=bc, act=ops.ACT_LEAKY, slope=0.2)
        ep.in_ch_live = C
        wp = pack_conv_fwd(ops, w, cpad).clone()
        wp[..., C:] = 7.0          # weights of the (zero) padding channels must not matter
        y2 = torch.full_like(y, float("nan"))
        ops.conv_fwd(geom, xh, wp, y2, ep)
        close(nchw(y2), yr, what="fwd, live-channel hint")
    # backward: act', bias grad, data grad, weight grad
    gpre = ops.act_bwd(nhwc(gy).cuda(), y, ops.ACT_LEAKY, 0.2)
    db = ops.colsum(B * P * P, K, K, gpre)
    close(db, br.grad, what="db")
    dx = torch.empty(B, H, H, cpad, device="cuda")
    ops.conv_bwd_data(geom, gpre, pack_conv_dgrad(ops, w, cpad), dx, ops.epilogue())
    close(nchw(dx[..., :C]), xr.grad, what="dgrad")
    if cpad != C:
        assert dx[..., C:].abs().max().item() == 0.0
    dw = torch.empty(K, C, R, R, device="cuda")
    db2 = torch.empty(K, device="cuda")
    ops.conv_bwd_weight(geom, xh, gpre, dw, C, K, C * R * R, R * R, 1, db=db2)
    close(dw, wr.grad, what="wgrad")
    close(db2, br.grad, what="db fused into wgrad")


CONVT_CASES = [
    # B, Cin, Hin, Cout, R, stride, pad, out_pad   (mnist.py:64-72; ct2d audio_mnist.py:228-242)
    (6, 771, 1, 512, 3, 1, 0, 0), (6, 512, 3, 256, 3, 2, 0, 0), (6, 256, 7, 128, 3, 2, 1, 0),
    (6, 128, 13, 64, 3, 2, 1, 0), (6, 64, 25, 1, 4, 1, 0, 0),
    (3, 64, 4, 32, 5, 2, 2, 1), (3, 32, 8, 16, 5, 2, 2, 1), (2, 16, 16, 1, 5, 2, 2, 1), (2, 8, 5, 4, 5, 2, 2, 1),
    (512, 771, 1, 512, 3, 1, 0, 0),
]


@pytest.mark.parametrize("B,Ci,H,Co,R,stride,pad,opad", CONVT_CASES)
def test_conv_transpose2d_fwd_bwd(B, Ci, H, Co, R, stride, pad, opad, forced_tile):
    ops = _ops()
    g = torch.Generator().manual_seed(B * 77 + Ci + Co)
    x = torch.randn(B, Ci, H, H, generator=g)
    w = torch.randn(Ci, Co, R, R, generator=g) / (Ci * R * R / stride ** 2) ** 0.5
    b = torch.randn(Co, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = torch.tanh(F.conv_transpose2d(xr, wr, br, stride=stride, padding=pad, output_padding=opad))
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    Ho = yr.shape[2]
    cpad = (Ci + 3) // 4 * 4
    xh = torch.zeros(B, H, H, cpad, device="cuda")
    xh[..., :Ci] = nhwc(x).cuda()
    T = R * R
    geom = ops.geom(B, Ho, Ho, Co, H, H, cpad, R, R, stride, pad)   # the conv this convT is the dgrad of
    wf = torch.empty(Co, T, cpad, device="cuda")
    ops.pack_weights(w.cuda().contiguous(), wf, Co, T, Ci, cpad, T, 1, Co * T)
    y = torch.empty(B, Ho, Ho, Co, device="cuda")
    ops.conv_bwd_data(geom, xh, wf, y, ops.epilogue(bias=b.cuda(), act=ops.ACT_TANH))
    close(nchw(y), yr, what="convT fwd")
    gpre = ops.act_bwd(nhwc(gy).cuda(), y, ops.ACT_TANH, 0.0)
    close(ops.colsum(B * Ho * Ho, Co, Co, gpre), br.grad, what="db")
    wd = torch.empty(Ci, T, Co, device="cuda")
    ops.pack_weights(w.cuda().contiguous(), wd, Ci, T, Co, Co, Co * T, 1, T)
    dx = torch.empty(B, H, H, Ci, device="cuda")
    g2 = ops.geom(B, Ho, Ho, Co, H, H, Ci, R, R, stride, pad)
    ops.conv_fwd(g2, gpre, wd, dx, ops.epilogue())
    close(nchw(dx), xr.grad, what="convT dgrad")
    dw = torch.empty(Ci, Co, R, R, device="cuda")
    ops.conv_bwd_weight(geom, gpre, xh, dw, Co, Ci, Co * T, T, 1)
    close(dw, wr.grad, what="convT wgrad")


def test_fused_epilogue_mask_and_dact(forced_tile):
    """dgrad epilogue: v *= mask[img, c]; v *= leaky'(y_prev)  (Dropout2d + LeakyReLU backward folded in)."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    B, C, H, K, R = 9, 64, 6, 32, 3
    w = torch.randn(K, C, R, R, generator=g) * 0.1
    yprev = torch.randn(B, H, H, C, generator=g)
    mask = (torch.rand(B, C, generator=g) > 0.5).float() * 2
    gy = torch.randn(B, H - 2, H - 2, K, generator=g)
    geom = ops.geom(B, H, H, C, H - 2, H - 2, K, R, R, 1, 0)
    dx = torch.empty(B, H, H, C, device="cuda")
    ops.conv_bwd_data(geom, gy.cuda(), pack_conv_dgrad(ops, w, C), dx,
                      ops.epilogue(mask=mask.cuda(), dact_y=yprev.cuda(), dact=ops.ACT_LEAKY, dslope=0.1))
    ref = F.conv_transpose2d(nchw(gy), w)                       # plain dgrad
    ref = ref * mask[:, :, None, None] * torch.where(nchw(yprev) > 0, 1.0, 0.1)
    close(nchw(dx), ref, what="fused dgrad epilogue")


@pytest.mark.parametrize("shape", [(7, 64, 5), (33, 32, 9), (5, 6, 4)])     # float4 kernels / scalar kernels (C % 4)
@pytest.mark.parametrize("order", ["bn_drop", "drop_bn", "bn", "eval"])
def test_batchnorm_fwd_bwd(order, shape):
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    B, C, H = shape
    y = torch.randn(B, C, H, H, generator=g) * 2 + 0.5
    mask = (torch.rand(B, C, generator=g) > 0.5).float() * 2
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    go = torch.randn(B, C, H, H, generator=g)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(gam), bn.bias.copy_(bet)
        bn.running_mean.normal_(generator=g), bn.running_var.uniform_(0.5, 2, generator=g)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    bn.train(order != "eval")
    yr = y.clone().requires_grad_(True)
    m4 = mask[:, :, None, None]
    a = F.leaky_relu(yr, 0.1)                       # BN input is a LeakyReLU output in the reference stacks
    if order == "drop_bn":
        out = bn(a * m4)
    elif order == "bn_drop":
        out = bn(a) * m4
    else:
        out = bn(a)
    out.backward(go)
    ah = nhwc(a.detach()).cuda()
    mk = mask.cuda()
    rm, rv = rm0.clone().cuda(), rv0.clone().cuda()
    mask_in = mk if order == "drop_bn" else None
    mask_post = mk if order == "bn_drop" else None
    st = ops.bn_stats(ah, mask_in, B, H * H, C, gam.cuda(), bet.cuda(), rm, rv, 0.1, 1e-5, order != "eval")
    o = ops.bn_apply(ah, st, mask_in, mask_post, B, H * H, C)
    close(nchw(o), out, what="bn fwd")
    close(rm, bn.running_mean, rtol=1e-5, what="running_mean")
    close(rv, bn.running_var, rtol=1e-5, what="running_var")
    dgam, dbet, gx = ops.bn_bwd(ah, nhwc(go).cuda(), mask_in, mask_post, st, gam.cuda(), B, H * H, C,
                                order != "eval", 0.1)
    close(dgam, bn.weight.grad, what="dgamma")
    close(dbet, bn.bias.grad, what="dbeta")
    close(nchw(gx), yr.grad, what="bn bwd fused with leaky'")


def test_bce_adam_colsum_dropout():
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    logit = torch.randn(512, 1, generator=g) * 3
    for tgt in (0.0, 1.0):
        lr_ = logit.clone().requires_grad_(True)
        loss = F.binary_cross_entropy_with_logits(lr_, torch.full_like(lr_, tgt))
        (0.5 * loss).backward()
        out2, gl = ops.bce_logits(logit.cuda(), tgt, 0.5)
        close(out2[0:1], loss.reshape(1), rtol=1e-6, what="bce")
        close(out2[1:2], torch.sigmoid(logit).mean().reshape(1), rtol=1e-6, what="sigmoid mean")
        close(gl, lr_.grad, rtol=1e-5, what="bce grad")
    # Adam vs torch.optim.Adam, 3 steps, betas (0.5, 0.999) (mnist.py:176-179)
    p = torch.randn(10007, generator=g)
    pr = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([pr], lr=1e-4, betas=(0.5, 0.999))
    pd, m, v = p.cuda(), torch.zeros(10007, device="cuda"), torch.zeros(10007, device="cuda")
    for step in range(1, 4):
        gr = torch.randn(10007, generator=g)
        pr.grad = gr.clone()
        opt.step()
        ops.adam(pd, gr.cuda(), m, v, 1e-4, 0.5, 0.999, 1e-8, step)
    assert (pd.cpu() - pr.detach()).abs().max().item() < 1e-7
    x = torch.randn(3001, 40, generator=g)
    close(ops.colsum(3001, 40, 40, x.cuda()), x.sum(0), rtol=1e-5, what="colsum")
    close(ops.colsum(3001, 37, 40, x.cuda()), x[:, :37].sum(0), rtol=1e-5, what="colsum scalar path")
    xw = torch.randn(515, 1024, generator=g)
    close(ops.colsum(515, 1024, 1024, xw.cuda()), xw.sum(0), rtol=1e-5, what="colsum wide")
    mk = ops.dropout_mask(7, 0, 0.2, 4096, 64, torch.device("cuda"))
    vals = torch.unique(mk).cpu()
    assert set(np.round(vals.numpy(), 4).tolist()) == {0.0, 1.25}
    assert abs((mk > 0).float().mean().item() - 0.8) < 0.01
    assert not torch.equal(mk, ops.dropout_mask(7, 4096 * 64, 0.2, 4096, 64, torch.device("cuda")))


@pytest.mark.parametrize("Co,stride,pad,opad,R", [(1, 2, 2, 1, 5), (2, 2, 1, 0, 4), (1, 1, 0, 0, 3), (3, 2, 1, 1, 5), (7, 2, 1, 0, 5)])
def test_col2im_gathers_scatter_form_transposed_conv(Co, stride, pad, opad, R):
    """ali_col2im over per-pixel tap contributions == F.conv_transpose2d (the 1x1 GEMM that produces the contributions is
    the ordinary ali_conv_fwd, covered above; here it is an einsum)."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    B, H, W, Ci = 3, 9, 7, 16
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Ci, Co, R, R, generator=g) * 0.2
    b = torch.randn(Co, generator=g)
    ref = torch.tanh(F.conv_transpose2d(x, w, b, stride=stride, padding=pad, output_padding=opad))
    contrib = torch.einsum("bchw,cot->bhwto", x, w.reshape(Ci, Co, R * R)).reshape(B, H, W, Co * R * R).contiguous()
    Ho, Wo = ref.shape[2], ref.shape[3]
    out = torch.empty(B, Ho, Wo, Co, device="cuda")
    ops.col2im(contrib.cuda(), Co * R * R, b.cuda(), out, B, H, W, Ho, Wo, Co, Co, R, R, stride, pad, ops.ACT_TANH, 0.0)
    close(nchw(out), ref, rtol=1e-5, what="col2im")


@pytest.mark.parametrize("B,H,W,Co,stride,pad,opad,R,act", [
    (3, 20, 28, 1, 2, 2, 1, 5, "tanh"),     # Generator tail of the spectrogram stacks (audio_mnist.py:243), small
    (2, 70, 90, 1, 2, 2, 1, 5, "none"),     # several output tiles per image, ragged edges
    (2, 31, 31, 1, 2, 1, 1, 5, "none"),     # first Conv2d's consumed input plane: 31^2 x 64 -> 64^2 (pad 1)
    (2, 40, 33, 2, 2, 1, 0, 4, "leaky"),    # two output channels, 4x4
    (2, 25, 25, 1, 1, 0, 0, 3, "none"),     # stride 1
    (1, 9, 150, 2, 3, 1, 2, 5, "tanh"),     # stride 3, wide
])
def test_fused_scatter_transposed_conv(B, H, W, Co, stride, pad, opad, R, act):
    """ali_tconv_scatter (tap contributions on the matrix cores, kept in LDS, summed per output pixel: one launch)
    == F.conv_transpose2d with bias and activation, and == the two-launch form (1x1 GEMM + ali_col2im) it replaces."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 7 + H + R)
    Ci = 64
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Ci, Co, R, R, generator=g) / (Ci * R * R / stride ** 2) ** 0.5
    b = torch.randn(Co, generator=g) * 0.1
    a_id, fn = {"tanh": (ops.ACT_TANH, torch.tanh), "none": (ops.ACT_NONE, lambda v: v),
                "leaky": (ops.ACT_LEAKY, lambda v: F.leaky_relu(v, 0.2))}[act]
    ref = fn(F.conv_transpose2d(x, w, b, stride=stride, padding=pad, output_padding=opad))
    Ho, Wo = ref.shape[2], ref.shape[3]
    assert ops.tconv_scatter_ok(Ci, Co, R, R, stride)
    wn = w.reshape(Ci, Co, R * R).permute(2, 1, 0).reshape(R * R * Co, Ci).contiguous().cuda()    # rows tap * Co + co
    xh = nhwc(x).cuda()
    out = torch.full((B, Ho, Wo, Co), float("nan"), device="cuda")
    ops.tconv_scatter(xh, wn, b.cuda(), out, B, H, W, Ci, Ho, Wo, Co, Co, R, R, stride, pad, a_id, 0.2)
    close(nchw(out), ref, what="fused scatter transposed conv")
    contrib = torch.empty(B, H, W, Co * R * R, device="cuda")
    ops.conv_fwd(ops.geom(B, H, W, Ci, H, W, Co * R * R, 1, 1, 1, 0), xh, wn.reshape(Co * R * R, 1, Ci), contrib, ops.epilogue())
    two = torch.empty(B, Ho, Wo, Co, device="cuda")
    ops.col2im(contrib, Co * R * R, b.cuda(), two, B, H, W, Ho, Wo, Co, Co, R, R, stride, pad, a_id, 0.2)
    close(out, two, rtol=2e-5, what="fused vs GEMM + col2im")
    # strided output (one plane of a wider tensor)
    wide = torch.zeros(B, Ho, Wo, Co + 3, device="cuda")
    ops.tconv_scatter(xh, wn, None, wide, B, H, W, Ci, Ho, Wo, Co, Co + 3, R, R, stride, pad)
    assert wide[..., Co:].abs().max().item() == 0.0
    close(nchw(wide[..., :Co].contiguous()), F.conv_transpose2d(x, w, None, stride=stride, padding=pad, output_padding=opad),
          what="fused scatter, strided output")
    with ops.tuning(ALI_NO_T1_MFMA=1):
        assert not ops.tconv_scatter_ok(Ci, Co, R, R, stride)


@pytest.mark.parametrize("B,H,W,stride,pad,opad,R", [(3, 20, 28, 2, 2, 1, 5), (2, 37, 19, 2, 1, 1, 5), (2, 64, 64, 2, 2, 1, 5),
                                                     (2, 12, 40, 1, 1, 0, 3), (1, 9, 31, 3, 0, 2, 4)])
def test_strided_one_channel_transposed_conv_weight_gradient(B, H, W, stride, pad, opad, R):
    """ali_tconv_scatter_wgrad: dW of ConvTranspose2d(64 -> 1, R, stride) as a contraction over pixels on the matrix cores
    (bands of the 64-channel input, the output gradient's rows in LDS) vs torch autograd, contiguous and pack-order dW."""
    ops = _ops()
    g = torch.Generator().manual_seed(H * 5 + W + R)
    Ci = 64
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Ci, 1, R, R, generator=g, requires_grad=True)
    y = F.conv_transpose2d(x, w, None, stride=stride, padding=pad, output_padding=opad)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    Ho, Wo = y.shape[2], y.shape[3]
    xh, gyh = nhwc(x).cuda(), nhwc(gy).cuda()
    dw = torch.full((Ci, 1, R, R), float("nan"), device="cuda")
    assert ops.tconv_scatter_wgrad(xh, gyh, 1, dw, dw.stride(0), dw.stride(3), B, H, W, Ci, Ho, Wo, R, R, stride, pad) is dw
    close(dw, w.grad, what="strided one-channel convT weight gradient")
    buf = torch.full((R * R, Ci), float("nan"), device="cuda")          # [taps][Ci]: the forward pack's order
    dwp = torch.as_strided(buf, (Ci, 1, R, R), (1, R * R * Ci, R * Ci, Ci))
    ops.tconv_scatter_wgrad(xh, gyh, 1, dwp, dwp.stride(0), dwp.stride(3), B, H, W, Ci, Ho, Wo, R, R, stride, pad)
    assert torch.equal(dwp.contiguous(), dw)
    # its slabs live behind the workspace's reserved head (the GEMM kernels' arrival counters, left at zero)
    assert not ops.workspace(torch.device("cuda"))[:4096].any().item()


@pytest.mark.parametrize("n_fft,win,hop,pad,L", [(255, 128, None, 96, 8000), (511, 128, 24, 64, 2900),
                                                   (1023, 256, 79, 200, 9000)])
def test_spectrogram_front_end_vs_torch_stft(n_fft, win, hop, pad, L):
    """the three dataset adapters' torchaudio.transforms.Spectrogram settings.  torchaudio is absent here:
    pinned to torch.stft with torchaudio's parameter mapping (pad both sides with zeros, centre=True / reflect, Hann
    window centred in the n_fft frame, power 2, one-sided)."""
    _ops()
    from ali_hip.spectrogram import SpectrogramFrontEnd
    g = torch.Generator().manual_seed(8)
    x = torch.randn(3, L, generator=g) * torch.linspace(0.1, 1.0, L)
    hop_ = hop or win // 2
    xp = F.pad(x, (pad, pad))
    spec = torch.stft(xp.double(), n_fft=n_fft, hop_length=hop_, win_length=win,
                      window=torch.hann_window(win, dtype=torch.float64), center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True).abs().pow(2)
    ref = (spec + 1e-6).log().float()
    fe = SpectrogramFrontEnd(n_fft, win, hop, pad)
    got = fe(x.cuda())
    assert got.shape == ref.shape, (got.shape, ref.shape)
    # power is compared relative to its scale (the log of a near-zero bin amplifies rounding of the bin itself)
    close(got.exp(), ref.exp(), rtol=1e-5, what="power spectrogram")
    mean, std = ref.mean(dim=(0, 1)), ref.std(dim=(0, 1))
    img = torch.clip((ref - mean) / (std + 1e-6), -3, 3) / 3
    sel = (ref > ref.max() - 12)                    # bins well above the 1e-6 floor: log is well conditioned there
    got_img = fe(x.cuda(), mean.cuda(), std.cuda()).cpu()
    assert (got_img - img)[sel].abs().max().item() < 1e-4


@pytest.mark.parametrize("kind,B,C,H,K,R,stride,pad", [
    ("dgrad", 512, 128, 8, 256, 4, 2, 0), ("dgrad", 512, 256, 7, 512, 3, 2, 0), ("dgrad", 128, 64, 11, 128, 4, 1, 0),
    ("fwd", 512, 128, 13, 256, 3, 2, 1), ("fwd", 192, 64, 14, 128, 4, 2, 1), ("dgrad", 576, 128, 13, 256, 3, 2, 1)])
def test_cost_ordered_dispatch_is_a_pure_reordering(kind, B, C, H, K, R, stride, pad):
    """ali_conv_tile_order: launches whose M-tiles have different k-loop lengths (padding / strided transposed taps
    skipped tile-wide) are dispatched in a cost-aware order (tests/test_host_cpu.py checks the balance it achieves).
    The table is a permutation of the M-tiles, and the launch computes bit-identical results with and without it
    (incl. tail-split grids and the fused BatchNorm partial sums, whose slots are indexed by the logical tile)."""
    import ctypes
    import os
    import ali_hip
    ops = _ops()
    P = (H + 2 * pad - R) // stride + 1
    geom = ops.geom(B, H, H, C, P, P, K, R, R, stride, pad)
    which = 0 if kind == "fwd" else 1
    buf = (ctypes.c_int32 * 65536)()
    n = ali_hip.load().ali_conv_tile_order(ctypes.byref(geom), which, 0, ctypes.cast(buf, ctypes.c_void_p), 65536)
    tiles, rows, pm = ops.conv_mtiles(geom, which)
    assert n == tiles and pm and sorted(buf[:n]) == list(range(n))
    g = torch.Generator(device="cuda").manual_seed(3)
    T = R * R
    if kind == "fwd":
        a = torch.randn(B, H, H, C, device="cuda", generator=g)
        w = torch.randn(K, T, C, device="cuda", generator=g) / (C * T) ** 0.5
        out_shape, run = (B, P, P, K), ops.conv_fwd
    else:
        a = torch.randn(B, P, P, K, device="cuda", generator=g)
        w = torch.randn(C, T, K, device="cuda", generator=g) / (K * T) ** 0.5
        out_shape, run = (B, H, H, C), ops.conv_bwd_data
    outs = []
    try:
        for off in ("0", "1"):
            os.environ["ALI_NO_ORDER"] = off
            ops.reload_tuning()
            y = torch.full(out_shape, float("nan"), device="cuda")
            slots = ops.conv_mtiles(geom, which)[0]
            part = torch.zeros(2 * out_shape[3] * slots, device="cuda")
            if kind == "fwd":
                ep = ops.epilogue(act=ops.ACT_LEAKY, slope=0.1, bn_fwd=(part, 1, None))
            else:
                ep = ops.epilogue()
            run(geom, a, w, y, ep)
            outs.append((y, part))
    finally:
        os.environ.pop("ALI_NO_ORDER", None)
        ops.reload_tuning()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert not torch.isnan(outs[0][0]).any()


@pytest.mark.parametrize("B,C,K,R,H", [(96, 64, 128, 1, 1), (64, 256, 512, 3, 3)])
def test_column_range_operands(B, C, K, R, H):
    """AliEpilogue.in_ld / out_ld and dy_ld: operands that are column ranges of wider row-major buffers (D's joint
    [dx | dz] rows, mnist.py:152-154) give exactly what dense copies of them give -- forward output (+ mask) into a
    column range, data gradient and weight gradient (+ fused bias gradient) from one."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(11)
    T = R * R
    x = torch.randn(B, H, H, C, device="cuda", generator=g)
    w = torch.randn(K, T, C, device="cuda", generator=g) / (C * T) ** 0.5
    wd = torch.randn(C, T, K, device="cuda", generator=g) / (K * T) ** 0.5
    bias = torch.randn(K, device="cuda", generator=g)
    geom = ops.geom(B, H, H, C, 1, 1, K, R, R, 1, 0)
    ld, off = K + 96, 32
    wide = torch.full((B, ld), 3.0, device="cuda")
    mask_w = (torch.rand(B, ld, device="cuda", generator=g) > 0.3).float() * 1.25
    # forward: act(conv + bias) * mask into columns [off, off + K)
    ref = torch.empty(B, 1, 1, K, device="cuda")
    ops.conv_fwd(geom, x, w, ref, ops.epilogue(bias=bias, act=ops.ACT_LEAKY, slope=0.1,
                                               mask=mask_w[:, off:off + K].contiguous()))
    ep = ops.epilogue(bias=bias, act=ops.ACT_LEAKY, slope=0.1)
    ep.mask, ep.mask_ld = mask_w[:, off:off + K].data_ptr(), ld
    ops.conv_fwd(geom, x, w, wide[:, off:off + K].unflatten(1, (1, 1, K)), ep, out_ld=ld)
    assert torch.equal(wide[:, off:off + K], ref.reshape(B, K))
    assert (wide[:, :off] == 3.0).all() and (wide[:, off + K:] == 3.0).all()
    # backward from a column range
    gw = torch.randn(B, ld, device="cuda", generator=g)
    gd = gw[:, off:off + K].contiguous().reshape(B, 1, 1, K)
    gv = gw[:, off:off + K].unflatten(1, (1, 1, K))
    dx1, dx2 = torch.empty(B, H, H, C, device="cuda"), torch.empty(B, H, H, C, device="cuda")
    ops.conv_bwd_data(geom, gd, wd, dx1, ops.epilogue())
    ops.conv_bwd_data(geom, gv, wd, dx2, ops.epilogue(), in_ld=ld)
    assert torch.equal(dx1, dx2)
    dw1, dw2 = torch.empty(K, C, R, R, device="cuda"), torch.empty(K, C, R, R, device="cuda")
    db1, db2 = torch.empty(K, device="cuda"), torch.empty(K, device="cuda")
    ops.conv_bwd_weight(geom, x, gd, dw1, C, K, C * T, T, 1, db=db1)
    ops.conv_bwd_weight(geom, x, gv, dw2, C, K, C * T, T, 1, db=db2, dy_ld=ld)
    assert torch.equal(dw1, dw2) and torch.equal(db1, db2)


def test_deferred_weight_gradients_match_their_own_launches():
    """ops.FoldQueue: three weight gradients recorded (AliWgradJob / AliWgradFold) and issued as one multi-job GEMM
    launch + one multi-job slab fold == the same three launched one by one (same pixel split: bit-identical), incl. the
    fused bias gradient; a job's operands may be freed by the caller before the flush (the queue keeps them)."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(4)
    cases = [(512, 1, 1024, 1024, 1, 1, 0), (512, 8, 128, 256, 4, 2, 0), (512, 3, 256, 512, 3, 1, 0)]
    q = ops.FoldQueue(torch.device("cuda", 0))
    geoms, outs, refs = [], [], []
    for (B, H, C, K, R, st, pad) in cases:
        P = (H + 2 * pad - R) // st + 1
        geoms.append(ops.geom(B, H, H, C, P, P, K, R, R, st, pad))
    q.expect(geoms)
    target = q.split_target()
    assert target == 682
    import os
    import ali_hip
    try:
        for (B, H, C, K, R, st, pad), geom in zip(cases, geoms):
            P = geom.P
            x = torch.randn(B, H, H, C, device="cuda", generator=g)
            dy = torch.randn(B, P, P, K, device="cuda", generator=g)
            dw, db = torch.full((K, C, R, R), float("nan"), device="cuda"), torch.full((K,), float("nan"), device="cuda")
            ops.conv_bwd_weight(geom, x, dy, dw, C, K, C * R * R, R * R, 1, db=db, defer=q)
            outs.append((dw, db))
            os.environ["ALI_WGRAD_BLOCKS"] = str(target)        # the stand-alone launch with the same pixel split
            ops.reload_tuning()
            dwr, dbr = torch.empty(K, C, R, R, device="cuda"), torch.empty(K, device="cuda")
            ops.conv_bwd_weight(geom, x, dy, dwr, C, K, C * R * R, R * R, 1, db=dbr)
            os.environ.pop("ALI_WGRAD_BLOCKS")
            ops.reload_tuning()
            refs.append((dwr, dbr))
            del x, dy
        assert len(q.launches) == 3 and len(q.jobs) == 3 and torch.isnan(outs[0][0]).all()
        q.flush()
    finally:
        os.environ.pop("ALI_WGRAD_BLOCKS", None)
        ops.reload_tuning()
    for (dw, db), (dwr, dbr) in zip(outs, refs):
        assert torch.equal(dw, dwr) and torch.equal(db, dbr)


def test_multi_job_gemm_launch_equals_single_launches():
    """ops.gemm_batch / ali_gemm_launch_multi: independent forward and data-gradient GEMMs recorded (AliGemmJob) and
    issued as ONE launch per kernel variant == the same GEMMs launched one by one, bit for bit -- incl. a split-K job
    (own slabs and arrival counters per job), a cost-ordered / tail-split job, a 128x32-tile job (narrow output: its own
    variant), fused epilogues, and a first-layer launch that no multi-job kernel serves (goes out at once)."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(31)

    def rnd(*shape, scale=1.0):
        return torch.randn(*shape, device="cuda", generator=g) * scale

    cases = []   # (which, geom, a, w, out shape, epilogue factory)
    # split-K 1x1 GEMM (M = 512 rows), the dz / dxz tail of mnist.py:98-105,127-136
    cases.append((0, ops.geom(512, 1, 1, 512, 1, 1, 512, 1, 1, 1, 0), rnd(512, 1, 1, 512), rnd(512, 1, 512, scale=0.05),
                  (512, 1, 1, 512), lambda: ops.epilogue(bias=bias512, act=ops.ACT_LEAKY, slope=0.2)))
    # a conv layer with padding: cost-ordered dispatch (+ tail split at this size)
    cases.append((0, ops.geom(512, 13, 13, 128, 7, 7, 256, 3, 3, 2, 1), rnd(512, 13, 13, 128), rnd(256, 9, 128, scale=0.03),
                  (512, 7, 7, 256), lambda: ops.epilogue(act=ops.ACT_LEAKY, slope=0.1)))
    # first layer (8-channel stride): no multi-job kernel -> launched at once, still correct
    cases.append((0, ops.geom(64, 28, 28, 8, 14, 14, 64, 3, 3, 2, 1), rnd(64, 28, 28, 8), rnd(64, 9, 8, scale=0.1),
                  (64, 14, 14, 64), lambda: ops.epilogue()))
    # transposed conv forward (data-gradient GEMM, 4 sub-pixel phases) with a mask epilogue
    cases.append((1, ops.geom(512, 7, 7, 256, 3, 3, 512, 3, 3, 2, 0), rnd(512, 3, 3, 512), rnd(256, 9, 512, scale=0.02),
                  (512, 7, 7, 256), lambda: ops.epilogue(mask=mask256)))
    # narrow output (32 channels): the 128x32-tile variant
    cases.append((1, ops.geom(512, 24, 24, 32, 11, 11, 64, 4, 4, 2, 0), rnd(512, 11, 11, 64), rnd(32, 16, 64, scale=0.05),
                  (512, 24, 24, 32), lambda: ops.epilogue()))
    bias512 = rnd(512)
    mask256 = (torch.rand(512, 256, device="cuda", generator=g) > 0.2).float() * 1.25

    def run(which, geom, a, w, shape, ep):
        out = torch.full(shape, float("nan"), device="cuda")
        (ops.conv_fwd if which == 0 else ops.conv_bwd_data)(geom, a, w, out, ep())
        return out

    refs = [run(*c) for c in cases]
    with ops.gemm_batch() as batch:
        outs = [run(*c) for c in cases]
        assert len(batch.jobs) == 4                       # four recorded, the first-layer launch (case 2) went out directly
        assert all(torch.isnan(o).all() for i, o in enumerate(outs) if i != 2) and not torch.isnan(outs[2]).any()
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert not torch.isnan(o).any() and torch.equal(o, r), f"job {i}"
    # again, back to back with other inputs: slabs / counters of the jobs' workspaces are left clean
    for c in cases:
        c[2].mul_(-0.5)
    refs = [run(*c) for c in cases]
    with ops.gemm_batch():
        outs = [run(*c) for c in cases]
    assert all(torch.equal(o, r) for o, r in zip(outs, refs))


@pytest.mark.parametrize("B,C,ld", [(512, 1024, 1024), (70, 36, 40), (1030, 512, 512)])
def test_discriminator_head_gemv(B, C, ld):
    """ali_head_fwd / ali_head_wgrad: Conv2d(C, 1, 1) on a 1x1 map (mnist.py:127) as a GEMV and its weight / bias gradient."""
    ops = _ops()
    g = torch.Generator().manual_seed(B + C)
    x = torch.randn(B, ld, generator=g)
    w = torch.randn(C, generator=g) / C ** 0.5
    b = torch.randn(1, generator=g)
    gy = torch.randn(B, generator=g)
    xd = x.cuda()
    y = ops.head_fwd(xd[:, :C], w.cuda(), b.cuda(), torch.empty(B, device="cuda"))
    close(y, (x[:, :C].double() @ w.double() + b.double()).float(), what="head fwd")
    dw, db = torch.empty(C, device="cuda"), torch.empty(1, device="cuda")
    ops.head_wgrad(xd[:, :C], gy.cuda(), dw, db)
    close(dw, (gy.double() @ x[:, :C].double()).float(), what="head dw")
    close(db, gy.double().sum().float().reshape(1), rtol=1e-5, what="head db")


def test_copy_multi_and_mask_peek():
    ops = _ops()
    from ali_hip import dropout
    src = [torch.randn(512, 784, device="cuda"), torch.arange(512, device="cuda", dtype=torch.int64),
           torch.randn(3, device="cuda"), torch.randn(512, 1, device="cuda").double()]
    dst = [torch.zeros_like(t) for t in src]
    strided = torch.randn(8, 6, device="cuda")[:, ::2]
    dst_s = torch.zeros(8, 3, device="cuda")
    ops.copy_multi(list(zip(dst, src)) + [(dst_s, strided)])
    assert all(torch.equal(d, t) for d, t in zip(dst, src)) and torch.equal(dst_s, strided)
    # peek_mask: the mask a later request will return, without consuming it (injected tape, paired passes)
    tape = [torch.rand(4, 8) for _ in range(6)]
    with dropout.injected_masks(tape):
        assert torch.equal(dropout.peek_mask(2, 4, 8, 0.5, torch.device("cuda")).cpu(), tape[2])
        assert dropout.peek_mask(2, 4, 9, 0.5, torch.device("cuda")) is None      # not the request that will come
        assert dropout.peek_mask(6, 4, 8, 0.5, torch.device("cuda")) is None
        assert torch.equal(dropout.next_mask(4, 8, 0.5, torch.device("cuda")).cpu(), tape[0])
        with dropout.paired_passes(3):
            both = dropout.peek_mask(1, 8, 8, 0.5, torch.device("cuda")).cpu()
            assert torch.equal(both, torch.cat([tape[2], tape[5]]))


def test_splitk_last_block_fold_stress():
    """The in-kernel split-K fold (slabs written and read with device-scope accesses, per-tile arrival counters) under
    back-to-back launches that reuse the same slabs with different data: a stale slab or a counter left non-zero
    would show up as an O(1) error.  600 launches, three shapes, alternating inputs."""
    import os
    import ali_hip
    ops = _ops()
    g = torch.Generator().manual_seed(12)
    old = os.environ.get("ALI_SPLITK")
    try:
        for (B, C, K, S) in ((512, 1024, 1024, 4), (64, 512, 256, 8), (1024, 512, 512, 3)):
            xs = [torch.randn(B, 1, 1, C, generator=g).cuda() * s for s in (1.0, -2.0)]
            w = (torch.randn(K, 1, C, generator=g) * 0.05).cuda()
            geom = ops.geom(B, 1, 1, C, 1, 1, K, 1, 1, 1, 0)
            os.environ["ALI_SPLITK"] = "1"
            ops.reload_tuning()
            refs = [ops.conv_fwd(geom, x, w, torch.empty(B, 1, 1, K, device="cuda"), ops.epilogue()).clone() for x in xs]
            os.environ["ALI_SPLITK"] = str(S)
            ops.reload_tuning()
            y = torch.empty(B, 1, 1, K, device="cuda")
            worst = torch.zeros((), device="cuda")
            for it in range(200):
                ops.conv_fwd(geom, xs[it & 1], w, y, ops.epilogue())
                worst = torch.maximum(worst, (y - refs[it & 1]).abs().max())
            scale = max(r.abs().max().item() for r in refs)
            assert worst.item() <= 1e-5 * scale, (B, C, K, S, worst.item(), scale)
    finally:
        if old is None:
            os.environ.pop("ALI_SPLITK", None)
        else:
            os.environ["ALI_SPLITK"] = old
        ops.reload_tuning()


@pytest.mark.parametrize("B,H,W,n_rows,Cg,Cx", [(37, 28, 28, 10, 8, 8), (5, 128, 128, 3, 2, 4), (600, 28, 28, 10, 1, 8),
                                                  (3, 256, 256, 70, 4, 8)])
def test_plane_table_grad_vs_torch(B, H, W, n_rows, Cg, Cx):
    """ali_plane_table_grad == tanh' * gradient, nearest-upsampling undone, scattered by class (torch formulation in
    ali_hip.planes.plane_to_table_grad)."""
    ops = _ops()
    from ali_hip.planes import plane_to_table_grad
    g = torch.Generator().manual_seed(21)
    g0 = torch.randn(B, H, W, Cg, generator=g).cuda()
    x0 = torch.tanh(torch.randn(B, H, W, Cx, generator=g)).cuda()
    idx = torch.randint(0, n_rows, (B, 2), generator=g).to(torch.int32).cuda()
    gch, xch, col = Cg - 1, 1, 1
    plane = x0[..., xch].reshape(B, H * W)
    ref = plane_to_table_grad(g0[..., gch].reshape(B, H * W) * (1 - plane * plane), idx[:, col], n_rows, H, W)
    got = ops.plane_table_grad(g0, gch, x0, xch, idx, col, n_rows)
    close(got, ref, rtol=2e-5, what="table grad")


def test_assemble_planes_matches_torch_modules():
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    B = 5
    emb = torch.nn.Embedding(10, 256)
    seq = torch.nn.Sequential(emb, torch.nn.Unflatten(1, (1, 16, 16)), torch.nn.Upsample(size=(28, 28)),
                              torch.nn.Tanh())
    X = torch.randn(B, 28, 28, generator=g)
    idx = torch.randint(0, 10, (B,), generator=g)
    cont = torch.randn(B, 3, generator=g)
    ref = torch.cat([X[:, None], seq(idx)] + [cont[:, j].reshape(B, 1, 1, 1).repeat(1, 1, 28, 28) for j in range(3)], 1)
    out = ops.assemble_planes(X.cuda(), idx.to(torch.int32).reshape(B, 1).cuda(), [emb.weight.detach().cuda()],
                              cont.cuda(), B, 28, 28, 8)
    close(nchw(out[..., :5]), ref, rtol=1e-6, what="planes")
    assert out[..., 5:].abs().max().item() == 0
    # scale-factor variant (audio: 16 -> 128)
    seq2 = torch.nn.Sequential(emb, torch.nn.Unflatten(1, (1, 16, 16)), torch.nn.Upsample(scale_factor=8),
                               torch.nn.Tanh())
    X2 = torch.randn(B, 128, 128, generator=g)
    out2 = ops.assemble_planes(X2.cuda(), idx.to(torch.int32).reshape(B, 1).cuda(), [emb.weight.detach().cuda()],
                               None, B, 128, 128, 4)
    close(out2[..., 1], seq2(idx)[:, 0], rtol=1e-6, what="planes x8")


@pytest.mark.parametrize("B,K,P,R,pad", [(5, 64, 25, 4, 0), (3, 32, 24, 5, 0), (2, 64, 9, 3, 1), (2, 128, 6, 3, 1)])
def test_direct_one_channel_kernels(B, K, P, R, pad):
    """ali_tconv1_{fwd,dgrad,wgrad}: ConvTranspose2d(K -> 1, stride 1) forward / input gradient / weight gradient."""
    ops = _ops()
    g = torch.Generator().manual_seed(K + P)
    x = torch.randn(B, K, P, P, generator=g)
    w = torch.randn(K, 1, R, R, generator=g) * 0.2
    b = torch.randn(1, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yprev = F.leaky_relu(xr, 0.2)
    yr = torch.tanh(F.conv_transpose2d(yprev, wr, br, stride=1, padding=pad))
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    H = yr.shape[2]
    T = R * R
    xin = nhwc(yprev.detach()).cuda()
    w_tk = torch.empty(1, T, K, device="cuda")
    ops.pack_weights(w.cuda().contiguous(), w_tk, 1, T, K, K, T, 1, T)
    y = torch.empty(B, H, H, 1, device="cuda")
    ops.tconv1_fwd(xin, w_tk, b.cuda(), y, B, P, P, K, R, R, pad, 1, ops.ACT_TANH, 0.0)
    close(nchw(y), yr, what="tconv1 fwd")
    gpre = ops.act_bwd(nhwc(gy).cuda(), y, ops.ACT_TANH, 0.0)
    gx = torch.empty(B, P, P, K, device="cuda")
    ops.tconv1_dgrad(gpre, 1, w_tk, xin, ops.ACT_LEAKY, 0.2, gx, B, P, P, K, R, R, pad)
    close(nchw(gx), xr.grad, what="tconv1 dgrad (fused leaky')")
    dw = torch.empty(K, 1, R, R, device="cuda")
    ops.tconv1_wgrad(xin, gpre, 1, 1, dw, T, 1, 0, B, P, P, K, R, R, pad)
    close(dw, wr.grad, what="tconv1 wgrad")
    # strided planes: output into / small operand from one channel of an NHWC tensor
    y4 = torch.zeros(B, H, H, 4, device="cuda")
    ops.tconv1_fwd(xin, w_tk, b.cuda(), y4[..., 2], B, P, P, K, R, R, pad, 4, ops.ACT_TANH, 0.0)
    close(y4[..., 2], y[..., 0], rtol=1e-6, what="strided out")
    # per-sample factor (one column of a [B, C] Dropout2d mask) applied in the same launch
    m5 = torch.rand(B, 5, generator=g).cuda()
    ys = torch.empty(B, H, H, 1, device="cuda")
    ops.tconv1_fwd(xin, w_tk, b.cuda(), ys, B, P, P, K, R, R, pad, 1, ops.ACT_TANH, 0.0, rowscale=m5[:, 3])
    assert torch.equal(ys, y * m5[:, 3].reshape(B, 1, 1, 1))
    # several small channels at once (first-layer weight gradient dW[k][c][tap]): channel 1 carries gpre
    g4 = torch.randn(B, H, H, 4, generator=g).cuda()
    g4[..., 1] = gpre[..., 0]
    dw2 = torch.zeros(K, 3, R, R, device="cuda")
    ops.tconv1_wgrad(xin, g4, 4, 3, dw2, 3 * T, 1, T, B, P, P, K, R, R, pad)
    close(dw2[:, 1], dw[:, 0], rtol=1e-6, what="multi-channel small")
    ref0 = torch.empty(K, 1, R, R, device="cuda")
    ops.tconv1_wgrad(xin, g4[..., 0].contiguous(), 1, 1, ref0, T, 1, 0, B, P, P, K, R, R, pad)
    close(dw2[:, 0], ref0[:, 0], rtol=1e-6, what="multi-channel small, plane 0")


FULL_SIZE = [  # the MorphoMNIST layer shapes at the bench batch (BASELINE.json configs[1], bs=512/GPU)
    ("conv", 512, 64, 14, 128, 4, 2, 1), ("conv", 512, 128, 7, 256, 4, 2, 1), ("conv", 512, 32, 24, 64, 4, 2, 0),
    ("conv", 512, 64, 11, 128, 4, 1, 0), ("conv", 512, 1024, 1, 1024, 1, 1, 0), ("conv", 512, 256, 3, 512, 3, 1, 0),
    ("convT", 512, 512, 3, 256, 3, 2, 0), ("convT", 512, 128, 13, 64, 3, 2, 1), ("convT", 512, 800, 1, 512, 3, 1, 0),
]


@pytest.mark.parametrize("kind,B,C,H,K,R,stride,pad", FULL_SIZE)
def test_full_size_adjoint_identities(kind, B, C, H, K, R, stride, pad):
    """Size-independent properties at the bench batch (too big for the CPU oracle): the three kernels of a layer are
    mutually adjoint,  <dy, conv(x, w)> = <dgrad(dy, w), x> = <wgrad(x, dy), w>,  and conv is linear in x."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(17)
    T = R * R
    if kind == "conv":
        P = (H + 2 * pad - R) // stride + 1
        xs, ys, wshape = (B, H, H, C), (B, P, P, K), (K, C, R, R)
        geom = ops.geom(B, H, H, C, P, P, K, R, R, stride, pad)
    else:   # ConvTranspose2d(C -> K): described by the conv it is the data gradient of (x := its output)
        P = (H - 1) * stride - 2 * pad + R
        xs, ys, wshape = (B, H, H, C), (B, P, P, K), (C, K, R, R)
        geom = ops.geom(B, P, P, K, H, H, C, R, R, stride, pad)
    x = torch.randn(xs, device="cuda", generator=g)
    x2 = torch.randn(xs, device="cuda", generator=g)
    dy = torch.randn(ys, device="cuda", generator=g)
    w = torch.randn(wshape, device="cuda", generator=g) / (C * T) ** 0.5
    y, y2, y3 = (torch.empty(ys, device="cuda") for _ in range(3))
    dx = torch.empty(xs, device="cuda")
    dw = torch.empty(wshape, device="cuda")
    if kind == "conv":
        wf = torch.empty(K, T, C, device="cuda")
        ops.pack_weights(w, wf, K, T, C, C, C * T, 1, T)
        wd = torch.empty(C, T, K, device="cuda")
        ops.pack_weights(w, wd, C, T, K, K, T, 1, C * T)
        fwd = lambda inp, out: ops.conv_fwd(geom, inp, wf, out, ops.epilogue())            # noqa: E731
        ops.conv_bwd_data(geom, dy, wd, dx, ops.epilogue())
        ops.conv_bwd_weight(geom, x, dy, dw, C, K, C * T, T, 1)
    else:
        wf = torch.empty(K, T, C, device="cuda")
        ops.pack_weights(w, wf, K, T, C, C, T, 1, K * T)
        wd = torch.empty(C, T, K, device="cuda")
        ops.pack_weights(w, wd, C, T, K, K, K * T, 1, T)
        fwd = lambda inp, out: ops.conv_bwd_data(geom, inp, wf, out, ops.epilogue())       # noqa: E731
        ops.conv_fwd(geom, dy, wd, dx, ops.epilogue())
        ops.conv_bwd_weight(geom, dy, x, dw, K, C, K * T, T, 1)
    fwd(x, y)
    fwd(x2, y2)
    fwd(2.5 * x - 0.75 * x2, y3)
    lin = (y3 - (2.5 * y - 0.75 * y2)).abs().max().item() / (y.abs().max().item() + 1e-30)
    assert lin < 1e-5, f"linearity {lin:.2e}"
    a = (dy.double() * y.double()).sum().item()
    b = (dx.double() * x.double()).sum().item()
    c = (dw.double() * w.double()).sum().item()
    scale = (dy.double().norm() * y.double().norm()).item()
    assert abs(a - b) <= 1e-5 * scale and abs(a - c) <= 1e-5 * scale, (a, b, c, scale)


# The layers the spectrogram benches actually run (bench.py --workload audio at 256 / GPU, --workload esrf at 64 / GPU:
# BASELINE.json configs[2] and [4]): the first three Encoder / Discriminator convolutions and the last three Generator
# transposed convolutions (audio_mnist.py:186-198,228-243; esrf_acoustic.py:144-199), at the bench batch, where pick_tile
# / wgrad_tile leave the 64x64 regime.       kind, B, C(stride), C live, H, K, R, stride, pad, out_pad
SPECT_FULL = {
    "audio-E0": ("conv", 256, 8, 7, 128, 64, 5, 2, 1, 0), "audio-E1": ("conv", 256, 64, 64, 63, 128, 5, 2, 1, 0),
    "audio-E2": ("conv", 256, 128, 128, 31, 256, 5, 2, 1, 0),
    "audio-G3": ("convT", 256, 256, 256, 16, 128, 5, 2, 2, 1), "audio-G4": ("convT", 256, 128, 128, 32, 64, 5, 2, 2, 1),
    "audio-G5": ("scatter", 256, 64, 64, 64, 1, 5, 2, 2, 1),
    "esrf-E0": ("conv", 64, 4, 3, 512, 64, 5, 2, 1, 0), "esrf-E1": ("conv", 64, 64, 64, 255, 128, 5, 2, 1, 0),
    "esrf-E2": ("conv", 64, 128, 128, 127, 256, 5, 2, 1, 0),
    "esrf-G5": ("convT", 64, 128, 128, 64, 64, 5, 2, 2, 1), "esrf-G6": ("convT", 64, 64, 64, 128, 64, 5, 2, 2, 1),
    "esrf-G7": ("scatter", 64, 64, 64, 256, 1, 5, 2, 2, 1),
}


@pytest.mark.parametrize("precision", ["f32", "f16", "f16mem"])
@pytest.mark.parametrize("name", list(SPECT_FULL))
def test_spectrogram_layers_at_the_bench_batch(name, precision):
    """Parity of the launches the audio / ESRF benches make, at their real size (the tiles pick_tile / wgrad_tile choose
    there: 64x128 / 128x128 fp32, 128x128 fp16).  Convolutions are independent per image, so the forward and the data
    gradient of three images of the full-size launch (first, middle, last: edge tiles included) are compared with torch's
    CPU fp32 convolution of just those images; the weight gradient (a sum over the whole batch) is compared with the
    64x64-tile kernel -- the one every oracle-sized test pins -- and through the adjoint identity <dW, W> = <dy, y>.
    f16: operands rounded to fp16 in flight; f16mem: read from fp16 twins (what a stepper iteration does).  Tolerances:
    fp32 2e-4 of max-abs (summation order only), fp16 4e-3 (operand rounding 2^-11 accumulated over the contraction)."""
    ops = _ops()
    kind, B, C, Cl, H, K, R, stride, pad, opad = SPECT_FULL[name]
    f16 = precision != "f32"
    if f16 and C % 32 and precision != "f16":
        pytest.skip("first layers (4 / 8 channels) read no fp16 twins")     # (f16: the row-walking kernel's fp16 variant)
    tol = 4e-3 if f16 else 2e-4
    g = torch.Generator(device="cuda").manual_seed(23)
    T = R * R
    pick = sorted({0, B // 2 - 1, B - 1})
    x = torch.randn(B, H, H, C, device="cuda", generator=g)
    if Cl < C:
        x[..., Cl:] = 0
    if kind == "conv":
        P = (H + 2 * pad - R) // stride + 1
        w = torch.randn(K, Cl, R, R, device="cuda", generator=g) / (Cl * T) ** 0.5
        wp = pack_conv_fwd(ops, w.cpu(), C)
        wd = pack_conv_dgrad(ops, w.cpu(), C)
        geom = ops.geom(B, H, H, C, P, P, K, R, R, stride, pad)
        fwd = lambda out: ops.conv_fwd(geom, x, wp, out, ops.epilogue())                    # noqa: E731
        bwd = lambda dy_, out: ops.conv_bwd_data(geom, dy_, wd, out, ops.epilogue())        # noqa: E731
        wgr = lambda dy_, out: ops.conv_bwd_weight(geom, x, dy_, out, Cl, K, Cl * T, T, 1)  # noqa: E731
        ref_fwd = lambda xi: F.conv2d(xi, w.cpu(), stride=stride, padding=pad)              # noqa: E731
        ref_bwd = lambda gi: F.conv_transpose2d(gi, w.cpu(), stride=stride, padding=pad,   # noqa: E731
                                                output_padding=H - ((P - 1) * stride - 2 * pad + R))
        yshape, dwshape = (B, P, P, K), (K, Cl, R, R)
    else:
        P = (H - 1) * stride - 2 * pad + R + opad
        w = torch.randn(C, K, R, R, device="cuda", generator=g) / (C * T / stride ** 2) ** 0.5
        geom = ops.geom(B, P, P, K, H, H, C, R, R, stride, pad)        # the conv this convT is the data gradient of
        wd = torch.empty(C, T, K, device="cuda")
        ops.pack_weights(w, wd, C, T, K, K, K * T, 1, T)
        if kind == "convT":
            wp = torch.empty(K, T, C, device="cuda")
            ops.pack_weights(w, wp, K, T, C, C, T, 1, K * T)
            fwd = lambda out: ops.conv_bwd_data(geom, x, wp, out, ops.epilogue())           # noqa: E731
        else:   # one output channel: per-pixel tap contributions by a 1x1 GEMM with N = taps columns, then ali_col2im
            wp = w.reshape(C, K, T).permute(2, 1, 0).reshape(T * K, 1, C).contiguous()
            gsc = ops.geom(B, H, H, C, H, H, T * K, 1, 1, 1, 0)

            def fwd(out):
                contrib = torch.empty(B, H, H, T * K, device="cuda")
                ops.conv_fwd(gsc, x, wp, contrib, ops.epilogue())
                ops.col2im(contrib, T * K, None, out, B, H, H, P, P, K, K, R, R, stride, pad)
        bwd = lambda dy_, out: ops.conv_fwd(geom, dy_, wd, out, ops.epilogue())             # noqa: E731
        wgr = lambda dy_, out: ops.conv_bwd_weight(geom, dy_, x, out, K, C, K * T, T, 1)    # noqa: E731
        ref_fwd = lambda xi: F.conv_transpose2d(xi, w.cpu(), stride=stride, padding=pad, output_padding=opad)  # noqa: E731
        ref_bwd = lambda gi: F.conv2d(gi, w.cpu(), stride=stride, padding=pad)             # noqa: E731
        yshape, dwshape = (B, P, P, K), (C, K, R, R)
    if precision == "f16mem":
        x._ali16 = x.half()
        ops.ensure_shadow16(wp), ops.ensure_shadow16(wd)
    y = torch.full(yshape, float("nan"), device="cuda")
    with ops.precision("f16" if f16 else "f32"):
        fwd(y)
    xs = nchw(x[pick][..., :Cl].cpu())
    close(nchw(y[pick].cpu()), ref_fwd(xs), tol, f"{name} forward, images {pick}")
    # output gradient correlated with the output: the adjoint identity's inner products are then O(|dy| |y|)
    dy = (y + 0.5 * torch.randn(yshape, device="cuda", generator=g) * y.std()).contiguous()
    if precision == "f16mem":
        dy._ali16 = dy.half()
    if not (kind == "scatter" and f16):          # (the one-channel end's gradients keep fp32 arithmetic on the path)
        dx = torch.full((B, H, H, C), float("nan"), device="cuda")
        with ops.precision("f16" if f16 else "f32"):
            bwd(dy, dx)
        ref = ref_bwd(nchw(dy[pick].cpu()))
        close(nchw(dx[pick][..., :Cl].cpu()), ref, tol, f"{name} data gradient, images {pick}")
        dw = torch.full(dwshape, float("nan"), device="cuda")
        with ops.precision("f16" if f16 else "f32"):
            wgr(dy, dw)
            with ops.tuning(ALI_WBM=64, ALI_WBN=64):
                dw64 = torch.full(dwshape, float("nan"), device="cuda")
                wgr(dy, dw64)
        close(dw, dw64, 2e-5 if not f16 else 1e-4, f"{name} weight gradient vs the 64x64-tile kernel")
        a = (dy.double() * y.double()).sum().item()
        c = (dw.double() * w.double()).sum().item()
        assert abs(a - c) <= (1e-5 if not f16 else 5e-4) * abs(a), (name, a, c)


FUSED_BN_CASES = [
    # B, C, H, K, R, stride, groups, drop      (MNIST D.dx: mnist.py:108-123) + ragged / split-K / 128-row tiles
    (256, 8, 28, 32, 5, 1, 2, True), (128, 32, 24, 64, 4, 2, 2, False), (128, 64, 11, 128, 4, 1, 1, False),
    (256, 128, 8, 256, 4, 2, 2, False), (70, 32, 9, 64, 4, 2, 1, True), (512, 128, 8, 256, 4, 2, 1, False),
]


@pytest.mark.parametrize("B,C,H,K,R,stride,groups,drop", FUSED_BN_CASES)
def test_batchnorm_reductions_fused_into_the_gemm_epilogues(B, C, H, K, R, stride, groups, drop, forced_tile):
    """nn.BatchNorm2d's batch statistics accumulated by the producing Conv2d's epilogue (optionally through a Dropout2d
    mask, per pass of a batched launch) and its backward reductions accumulated by the data-gradient GEMM's epilogue
    (AliEpilogue.bn_mode 1 / 2 + ali_bn_*_from_partials) vs torch's BatchNorm2d on the CPU (mnist.py:108-123)."""
    ops = _ops()
    g = torch.Generator().manual_seed(B + 7 * K)
    x = torch.randn(B, C, H, H, generator=g)
    w = torch.randn(K, C, R, R, generator=g) / (C * R * R) ** 0.5
    b = torch.randn(K, generator=g) * 0.1
    mask = (torch.rand(B, K, generator=g) > 0.3).float() / 0.7 if drop else None
    gamma, beta = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1
    P = (H - R) // stride + 1
    Bg = B // groups
    # ---- reference: conv -> lrelu -> [drop] -> bn (batch stats per pass), then a second conv's data gradient arrives
    wr = w.clone()
    y = F.leaky_relu(F.conv2d(x, wr, b, stride=stride), 0.1)
    ym = y * mask.reshape(B, K, 1, 1) if drop else y
    bn = torch.nn.BatchNorm2d(K)
    with torch.no_grad():
        bn.weight.copy_(gamma), bn.bias.copy_(beta)
    bn.train()
    outs = [bn(ym[i * Bg:(i + 1) * Bg]) for i in range(groups)]
    # ---- HIP: conv with fused statistics
    xh = nhwc(x).cuda()
    geom = ops.geom(B, H, H, C, P, P, K, R, R, stride, 0)
    slots, tile_rows, pm = ops.conv_mtiles(geom, 0)
    if groups > 1 and (Bg * (1 if pm else P * P)) % tile_rows:
        pytest.skip("passes of a batched launch must start on a tile boundary (chain.py falls back to ali_bn_stats)")
    part = torch.full((2 * K * slots,), float("nan"), device="cuda")
    yh = torch.empty(B, P, P, K, device="cuda")
    maskc = mask.cuda() if drop else None
    ops.conv_fwd(geom, xh, pack_conv_fwd(ops, w, C), yh,
                 ops.epilogue(bias=b.cuda(), act=ops.ACT_LEAKY, slope=0.1, bn_fwd=(part, groups, maskc, slots)))
    close(nchw(yh), y, what="conv out (stored unmasked)")
    rm, rv = torch.zeros(K, device="cuda"), torch.ones(K, device="cuda")
    st = ops.bn_stats_from_partials(part, slots, groups, K, Bg * P * P, gamma.cuda(), beta.cuda(), rm, rv, 0.1, 1e-5)
    st = st.reshape(groups, 4, K)
    for gi in range(groups):
        ref = ym[gi * Bg:(gi + 1) * Bg]
        close(st[gi, 0], ref.mean(dim=(0, 2, 3)), 1e-5, f"mean pass {gi}")
        close(1 / st[gi, 1] ** 2 - 1e-5, ref.var(dim=(0, 2, 3), unbiased=False), 1e-4, f"var pass {gi}")
    close(rm, bn.running_mean, 1e-5, "running_mean")
    close(rv, bn.running_var, 1e-5, "running_var")
    t = ops.bn_apply(yh, st[0] if groups == 1 else st, maskc, None, B, P * P, K, groups=groups)
    close(nchw(t), torch.cat(outs), what="bn out")
    # ---- backward of pass 0 through bn (+drop) with the reductions fused into a data-gradient GEMM's epilogue:
    # the consumer is a second conv K -> K2 (3x3 if the map allows), whose dgrad produces g = dL/dt
    R2 = 3 if P >= 3 else 1
    K2 = 64
    w2 = torch.randn(K2, K, R2, R2, generator=g) / (K * R2 * R2) ** 0.5
    P2 = P - R2 + 1
    gy2 = torch.randn(Bg, K2, P2, P2, generator=g)
    y0 = y[:Bg].clone().requires_grad_(True)
    bn2 = torch.nn.BatchNorm2d(K)
    with torch.no_grad():
        bn2.weight.copy_(gamma), bn2.bias.copy_(beta)
    bn2.train()
    t0 = bn2(y0 * mask[:Bg].reshape(Bg, K, 1, 1) if drop else y0)
    F.conv2d(t0, w2).backward(gy2)
    geom2 = ops.geom(Bg, P, P, K, P2, P2, K2, R2, R2, 1, 0)
    slots2 = ops.conv_mtiles(geom2, 1)[0]
    part2 = torch.full((2 * K * slots2,), float("nan"), device="cuda")
    gt = torch.empty(Bg, P, P, K, device="cuda")
    x_in = yh[:Bg].contiguous()
    m0 = maskc[:Bg].contiguous() if drop else None
    ops.conv_bwd_data(geom2, nhwc(gy2).cuda(), pack_conv_dgrad(ops, w2, K), gt,
                      ops.epilogue(bn_bwd=(part2, x_in, st[0, 0], st[0, 1], m0, None, slots2)))
    dgam, dbet, gprev = ops.bn_bwd_from_partials(part2, slots2, x_in, gt, m0, None, st[0], gamma.cuda(), Bg, P * P, K,
                                                 True, -1.0)
    close(dgam, bn2.weight.grad, what="dgamma")
    close(dbet, bn2.bias.grad, what="dbeta")
    close(nchw(gprev), y0.grad, what="gx")


def test_fused_batchnorm_slot_capacity_is_checked():
    """AliEpilogue.bn_slots: a launch whose M-tile count differs from what bn_part was sized for (a stale
    ali_conv_mtiles result after a tuning reload) is refused instead of writing out of bounds; the Python
    wrapper's reload drops its cached tile counts."""
    import ali_hip
    ops = _ops()
    B, C, H, K, R = 128, 64, 11, 128, 4
    P = H - R + 1
    geom = ops.geom(B, H, H, C, P, P, K, R, R, 1, 0)
    x = torch.randn(B, H, H, C, device="cuda")
    w = torch.randn(K, R * R, C, device="cuda") * 0.05
    y = torch.empty(B, P, P, K, device="cuda")
    slots = ops.conv_mtiles(geom, 0)[0]
    part = torch.zeros(2 * K * slots, device="cuda")
    ops.conv_fwd(geom, x, w, y, ops.epilogue(bn_fwd=(part, 1, None, slots)))
    with pytest.raises(RuntimeError, match="slots"):
        ops.conv_fwd(geom, x, w, y, ops.epilogue(bn_fwd=(part, 1, None, slots + 1)))
    with ops.tuning(ALI_BM=128, ALI_BN=64):
        slots128 = ops.conv_mtiles(geom, 0)[0]
        assert slots128 != slots                       # the cache was dropped with the reload
        with pytest.raises(RuntimeError, match="slots"):
            ops.conv_fwd(geom, x, w, y, ops.epilogue(bn_fwd=(part, 1, None, slots)))
    assert ops.conv_mtiles(geom, 0)[0] == slots
    assert isinstance(ali_hip.load().ali_last_error(), bytes)


F16_CASES = [
    # kind, B, C, H, K, R, stride, pad        5x5 stride-2 family of esrf_acoustic.py:144-199 (+ 1x1 tail, split-K, phases)
    ("conv", 4, 64, 31, 128, 5, 2, 1), ("conv", 3, 128, 15, 256, 5, 2, 1), ("conv", 64, 512, 1, 512, 1, 1, 0),
    ("conv", 2, 256, 7, 512, 5, 2, 1), ("convT", 3, 128, 8, 64, 5, 2, 2), ("convT", 2, 64, 16, 32, 5, 2, 2),
    ("conv", 2, 32, 40, 64, 5, 2, 1),
]


# first Conv2d of the spectrogram stacks (audio_mnist.py:186): conv_s2_first_kernel's fp16 variant, 4 / 8 channels
F16_FIRST_CASES = [("conv", 2, 4, 70, 64, 5, 2, 1), ("conv", 2, 8, 67, 64, 5, 2, 1), ("conv", 1, 4, 96, 64, 5, 2, 2)]


@pytest.mark.parametrize("kind,B,C,H,K,R,stride,pad", F16_CASES + F16_FIRST_CASES)
def test_fp16_mfma_gemm_path(kind, B, C, H, K, R, stride, pad, forced_tile):
    """AliEpilogue.mfma_f16 (BASELINE config 5): operands rounded to fp16 on their way into LDS, fp32 accumulation.
    (i) On data that fp16 holds exactly (small integers / powers of two) the result must EQUAL the fp32 path's bit for
    bit -- this pins the fragment layout of v_mfma_f32_32x32x16_f16; (ii) on random data it must sit within fp16
    operand rounding (2^-11 per operand, accumulated over the contraction) of the fp32 result."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 31 + C + K)
    for exact in (True, False):
        if exact:
            x = torch.randint(-4, 5, (B, C, H, H), generator=g).float()
            w = torch.randint(-2, 3, (K, C, R, R) if kind == "conv" else (C, K, R, R), generator=g).float() / 8
        else:
            x = torch.randn(B, C, H, H, generator=g)
            w = torch.randn((K, C, R, R) if kind == "conv" else (C, K, R, R), generator=g) / (C * R * R) ** 0.5
        xh = nhwc(x).cuda()
        if kind == "conv":
            P = (H + 2 * pad - R) // stride + 1
            geom = ops.geom(B, H, H, C, P, P, K, R, R, stride, pad)
            wp = pack_conv_fwd(ops, w, C)
            outs = []
            for prec in ("f32", "f16"):
                y = torch.empty(B, P, P, K, device="cuda")
                with ops.precision(prec):
                    ops.conv_fwd(geom, xh, wp, y, ops.epilogue())
                outs.append(y)
        else:   # ConvTranspose2d forward == data-gradient GEMM, 4 sub-pixel phases
            Ho = (H - 1) * stride - 2 * pad + R + 1
            geom = ops.geom(B, Ho, Ho, K, H, H, C, R, R, stride, pad)
            wp = torch.empty(K, R * R, C, device="cuda")
            ops.pack_weights(w.cuda().contiguous(), wp, K, R * R, C, C, R * R, 1, K * R * R)
            outs = []
            for prec in ("f32", "f16"):
                y = torch.empty(B, Ho, Ho, K, device="cuda")
                with ops.precision(prec):
                    ops.conv_bwd_data(geom, xh, wp, y, ops.epilogue())
                outs.append(y)
        if exact:
            assert torch.equal(outs[0], outs[1]), f"{kind}: fp16 path differs on exactly representable data"
        else:
            scale = outs[0].abs().max().item()
            err = (outs[0] - outs[1]).abs().max().item()
            assert 0 < err <= 4e-3 * scale, (err, scale)


@pytest.mark.parametrize("kind,B,C,H,K,R,stride,pad", F16_CASES + [("conv", 70, 64, 9, 96, 5, 2, 1)])
def test_fp16_mfma_weight_gradient(kind, B, C, H, K, R, stride, pad, forced_tile):
    """ali_conv_bwd_weight(mfma_f16 = 1): both operands rounded to fp16 on their way into LDS (kept [pixel][channel],
    read back transposed by ds_read_b64_tr_b16), fp32 accumulation and slabs.  Bit-identical to the fp32 path on data
    fp16 holds exactly and whose sums fp32 holds exactly; within operand rounding of it on random data."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 13 + C + K)
    for exact in (True, False):
        if kind == "conv":
            P = (H + 2 * pad - R) // stride + 1
            xs, ys = (B, H, H, C), (B, P, P, K)                 # gathered operand x, dense operand dy
            geom = ops.geom(B, H, H, C, P, P, K, R, R, stride, pad)
        else:   # ConvTranspose2d weight gradient: gathered = its output gradient, dense = its input
            Ho = (H - 1) * stride - 2 * pad + R + 1
            xs, ys = (B, Ho, Ho, K), (B, H, H, C)
            geom = ops.geom(B, Ho, Ho, K, H, H, C, R, R, stride, pad)
        if exact:
            x = torch.randint(-3, 4, xs, generator=g).float().cuda()
            dy = (torch.randint(-2, 3, ys, generator=g).float() / 4).cuda()
        else:
            x, dy = torch.randn(xs, generator=g).cuda(), torch.randn(ys, generator=g).cuda()
        Cg, Cd = xs[3], ys[3]
        outs = []
        for prec in ("f32", "f16"):
            dw = torch.empty(Cd, Cg, R, R, device="cuda")
            with ops.precision(prec):
                ops.conv_bwd_weight(geom, x, dy, dw, Cg, Cd, Cg * R * R, R * R, 1)
            outs.append(dw)
        if exact:
            assert torch.equal(outs[0], outs[1]), f"{kind}: fp16 weight gradient differs on exactly representable data"
        else:
            scale = outs[0].abs().max().item()
            err = (outs[0] - outs[1]).abs().max().item()
            assert 0 < err <= 4e-3 * scale, (err, scale)


@pytest.mark.parametrize("kind,B,C,H,K,R,stride,pad", [c for c in F16_CASES if c[2] % 64 == 0])
def test_fp16_operands_in_memory_path(kind, B, C, H, K, R, stride, pad, forced_tile):
    """AliEpilogue.in16 / w16 / out16: a precision("f16") launch leaves the fp16 twin of its output, and a launch whose
    operands both carry one reads those (16-byte gathers of 8 halves, 64-deep k-tiles) -- the same products as the
    converting fp16 path (both round the same fp32 values to fp16, RNE), summed in fp32 in the same k order up to where
    a split-K launch cuts its slabs (64-deep instead of 32-deep tiles): equal to fp32 summation rounding."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 5 + C + K)
    x = torch.randn(B, H, H, C, generator=g).cuda()
    w = torch.randn((K, C, R, R) if kind == "conv" else (C, K, R, R), generator=g) / (C * R * R) ** 0.5
    if kind == "conv":
        P = (H + 2 * pad - R) // stride + 1
        geom, which, oshape = ops.geom(B, H, H, C, P, P, K, R, R, stride, pad), 0, (B, P, P, K)
        wp = pack_conv_fwd(ops, w, C)
        run = ops.conv_fwd
    else:
        Ho = (H - 1) * stride - 2 * pad + R + 1
        geom, which, oshape = ops.geom(B, Ho, Ho, K, H, H, C, R, R, stride, pad), 1, (B, Ho, Ho, K)
        wp = torch.empty(K, R * R, C, device="cuda")
        ops.pack_weights(w.cuda().contiguous(), wp, K, R * R, C, C, R * R, 1, K * R * R)
        run = ops.conv_bwd_data
    with ops.precision("f16"):
        y_cvt = torch.empty(oshape, device="cuda")
        run(geom, x, wp, y_cvt, ops.epilogue())                       # operands fp32 in memory, converted in flight
        assert ops.shadow16(y_cvt) is not None and torch.equal(ops.shadow16(y_cvt), y_cvt.half())
        x._ali16 = x.half()
        ops.ensure_shadow16(wp)
        y_mem = torch.empty(oshape, device="cuda")
        run(geom, x, wp, y_mem, ops.epilogue())                       # both operands read as fp16
    close(y_mem, y_cvt, 2e-6, "fp16 operands in memory vs converted in flight")
    # and it really read the twins: poison them
    x._ali16.zero_()
    with ops.precision("f16"):
        y_z = torch.empty(oshape, device="cuda")
        run(geom, x, wp, y_z, ops.epilogue())
    assert y_z.abs().max().item() == 0.0


@pytest.mark.parametrize("kind,B,C,H,K,R,stride,pad,epi", [
    ("conv", 5, 64, 33, 256, 5, 2, 1, "plain"),       # ragged M (5 * 16^2 rows) and one N-tile
    ("conv", 3, 128, 31, 320, 5, 2, 1, "bias_act"),   # ragged N (320 = 256 + 64), image-major rows
    ("conv", 64, 128, 9, 512, 5, 2, 1, "mask"),       # pixel-major rows (B >= 64, small map): padding taps skipped tile-wide
    ("convT", 3, 256, 9, 128, 5, 2, 2, "plain"),      # data-gradient form: four sub-pixel phases, C_out of the GEMM = 256
    ("convT", 2, 320, 12, 64, 5, 2, 1, "dact"),       # act' of the previous layer in the epilogue, odd output size
    ("conv", 2, 192, 1, 256, 1, 1, 0, "plain"),       # 1x1 on a 1x1 map: M = 2
])
def test_fp16_lds_dma_kernel(kind, B, C, H, K, R, stride, pad, epi):
    """gconv_kernel512<256,256,...,F16=3> (fp16 twins by LDS-DMA into a swizzled image, 8 waves): what pick_tile gives the
    large fp16 layers of the spectrogram benches, forced here on small shapes.  Same products and the same k order as the
    register-staged twin loop: EQUAL to it bit for bit (ALI_NO_DMA16=1 runs that one), and within fp32 summation rounding
    of torch's convolution of the fp16-rounded operands."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 3 + C + K)
    x = torch.randn(B, H, H, C, generator=g).cuda()
    if kind == "conv":
        w = torch.randn(K, C, R, R, generator=g) / (C * R * R) ** 0.5
        P = (H + 2 * pad - R) // stride + 1
        geom, oshape, nout = ops.geom(B, H, H, C, P, P, K, R, R, stride, pad), (B, P, P, K), K
        wp = pack_conv_fwd(ops, w, C)
        run = ops.conv_fwd
        ref = F.conv2d(nchw(x.cpu().half().float()), w.half().float(), stride=stride, padding=pad)
    else:
        w = torch.randn(C, K, R, R, generator=g) / (C * R * R / stride ** 2) ** 0.5
        Ho = (H - 1) * stride - 2 * pad + R + 1
        geom, oshape, nout = ops.geom(B, Ho, Ho, K, H, H, C, R, R, stride, pad), (B, Ho, Ho, K), K
        wp = torch.empty(K, R * R, C, device="cuda")
        ops.pack_weights(w.cuda().contiguous(), wp, K, R * R, C, C, R * R, 1, K * R * R)
        run = ops.conv_bwd_data
        ref = F.conv_transpose2d(nchw(x.cpu().half().float()), w.half().float(), stride=stride, padding=pad, output_padding=1)
    ref = nhwc(ref)
    bias = torch.randn(nout, generator=g).cuda() * 0.1
    mask = (torch.rand(B, nout, generator=g) > 0.3).float().cuda() * 1.25
    yprev = torch.randn(oshape, generator=g).cuda()
    yprev._ali16 = yprev.half()            # act' is then evaluated on the twin (AliEpilogue.dact_y16): same signs

    def epilogue():
        if epi == "bias_act":
            return ops.epilogue(bias=bias, act=ops.ACT_LEAKY, slope=0.2)
        if epi == "mask":
            return ops.epilogue(bias=bias, act=ops.ACT_LEAKY, slope=0.2, mask=mask)
        if epi == "dact":
            return ops.epilogue(dact_y=yprev, dact=ops.ACT_LEAKY, dslope=0.2)
        return ops.epilogue()
    if epi in ("bias_act", "mask"):
        ref = F.leaky_relu(ref + bias.cpu(), 0.2)
    if epi == "mask":
        ref = ref * mask.cpu().reshape(B, 1, 1, nout)
    if epi == "dact":
        ref = ref * torch.where(yprev._ali16.float().cpu() > 0, 1.0, 0.2)
    x._ali16 = x.half()
    ops.ensure_shadow16(wp)
    outs = []
    for env in ({"ALI_BM": 256, "ALI_BN": 256}, {"ALI_NO_DMA16": 1, "ALI_SPLITK": 1}):   # (whole k-loops in both)
        with ops.precision("f16"), ops.tuning(**env):
            y = torch.full(oshape, float("nan"), device="cuda")
            run(geom, x, wp, y, epilogue())
            outs.append(y)
            assert torch.equal(ops.shadow16(y), y.half())          # the fp16 twin for the next layer
    close(outs[0], ref, what=f"LDS-DMA fp16 kernel ({kind}, {epi})")
    assert torch.equal(outs[0], outs[1]), "LDS-DMA loop differs from the register-staged twin loop"
    x._ali16.zero_()                                                  # ... and it really read the twins
    with ops.precision("f16"), ops.tuning(ALI_BM=256, ALI_BN=256):
        z = torch.empty(oshape, device="cuda")
        run(geom, x, wp, z, ops.epilogue())
    assert z.abs().max().item() == 0.0


@pytest.mark.parametrize("kind,B,C,H,K,R,stride,pad", [c for c in F16_CASES if c[2] % 8 == 0 and c[4] >= 64])
def test_fp16_weight_gradient_from_twins(kind, B, C, H, K, R, stride, pad, forced_tile):
    """ali_conv_bwd_weight fed the fp16 twins of both operands (x16 / dy16): same products as the converting fp16 path
    (the twins ARE the rounded operands), fp32 sums up to slab grouping; the fused Conv2d bias gradient is the column
    sum of the twin.  Poisoned twins prove they are what is read."""
    if forced_tile and forced_tile[3] < 64:
        pytest.skip("the twin-reading weight-gradient loop is instantiated for >= 64 dense channels per tile (wgrad.hip: mem16)")
    ops = _ops()
    g = torch.Generator().manual_seed(B * 3 + C + K)
    if kind == "conv":
        P = (H + 2 * pad - R) // stride + 1
        xs, ys = (B, H, H, C), (B, P, P, K)
        geom = ops.geom(B, H, H, C, P, P, K, R, R, stride, pad)
    else:
        Ho = (H - 1) * stride - 2 * pad + R + 1
        xs, ys = (B, Ho, Ho, K), (B, H, H, C)
        geom = ops.geom(B, Ho, Ho, K, H, H, C, R, R, stride, pad)
    x, dy = torch.randn(xs, generator=g).cuda(), torch.randn(ys, generator=g).cuda()
    Cg, Cd = xs[3], ys[3]
    outs, dbs = [], []
    with ops.precision("f16"):
        for twins in (False, True):
            if twins:
                x._ali16, dy._ali16 = x.half(), dy.half()
            dw = torch.empty(Cd, Cg, R, R, device="cuda")
            db = torch.empty(Cd, device="cuda")
            ops.conv_bwd_weight(geom, x, dy, dw, Cg, Cd, Cg * R * R, R * R, 1, db=db)
            outs.append(dw), dbs.append(db)
        close(outs[1], outs[0], 2e-6, "weight gradient from twins vs converted in flight")
        close(dbs[1], dy.half().float().sum(dim=(0, 1, 2)), 1e-5, "bias gradient from the twin")
        close(dbs[0], dy.sum(dim=(0, 1, 2)), 1e-5, "bias gradient (fp32 operand)")
        x._ali16.zero_()
        dw = torch.empty(Cd, Cg, R, R, device="cuda")
        ops.conv_bwd_weight(geom, x, dy, dw, Cg, Cd, Cg * R * R, R * R, 1)
        assert dw.abs().max().item() == 0.0


@pytest.mark.parametrize("K,C,R,cpad", [(64, 5, 3, 8), (128, 64, 4, 64), (96, 771, 3, 800), (33, 40, 5, 40), (512, 512, 1, 512),
                                         (1024, 512, 5, 512), (4100, 2052, 1, 2052)])   # the last two: tiled transposes
def test_pack_weights_tiled_transposes_and_fp16_twin(K, C, R, cpad):
    """ali_pack_weights_multi: the Conv2d forward pack [K][T][Cpad] and data-gradient pack [Cpad][T][K] (tiled LDS
    transposes; element-wise fallback for the shapes that are not) vs torch permutes, zero padding channels included,
    and the fp16 twin written by the same launch."""
    ops = _ops()
    w = torch.randn(K, C, R, R, generator=torch.Generator().manual_seed(K + C)).cuda()
    T = R * R
    fwd = torch.full((K, T, cpad), float("nan"), device="cuda")
    ops.ensure_shadow16(fwd)
    ops.pack_weights(w, fwd, K, T, C, cpad, C * T, 1, T)
    ref = torch.zeros(K, T, cpad, device="cuda")
    ref[:, :, :C] = w.permute(0, 2, 3, 1).reshape(K, T, C)
    assert torch.equal(fwd, ref)
    assert torch.equal(ops.shadow16(fwd), ref.half())
    dg = torch.full((cpad, T, K), float("nan"), device="cuda")
    with ops.batched_packs():
        ops.pack_weights(w, dg, C, T, K, K, T, 1, C * T)
    ref = w.permute(1, 2, 3, 0).reshape(C, T, K)
    assert torch.equal(dg[:C], ref) and torch.isnan(dg[C:]).all()        # rows >= C belong to the caller (kept zero)
