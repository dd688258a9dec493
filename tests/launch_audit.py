"""In-situ audit of the GEMM launches of a real iteration (tests only; used through ``ali_hip.ops.launch_hook``).

A LeakyReLU input within fp32 summation noise of zero takes either slope depending on summation order, so at the
reference's width d=64 (hundreds of such units per iteration) a whole-iteration gradient can only be compared with the
oracle's to ~1e-2 -- loose enough to hide a wrong tap in one layer.  The audit removes that: every ``ali_conv_fwd`` /
``ali_conv_bwd_data`` / ``ali_conv_bwd_weight`` launch the iteration makes is re-computed right after it ran, with
torch's CPU fp32 convolution, from the very operands the kernel read (so no sign pattern can differ), epilogue included
(bias, activation, Dropout2d mask, act' of the previous layer), and held to the kernel tests' bound: 2e-4 of the
result's max-abs.  fp16-MFMA launches are compared with the same convolution of the fp16-ROUNDED operands (products of
fp16 values are exact in fp32: only the summation order differs), so they are held to the same bound.
"""
import torch
import torch.nn.functional as F

ACT_NONE, ACT_LEAKY, ACT_TANH = 0, 1, 2


def _geom(g):
    return tuple(int(getattr(g, n)) for n in ("B", "H", "W", "C", "P", "Q", "K", "R", "S", "stride", "pad"))


def _act(v, act, slope):
    if act == ACT_LEAKY:
        return torch.where(v > 0, v, v * slope)
    if act == ACT_TANH:
        return torch.tanh(v)
    return v


def _dact(y, act, slope):
    if act == ACT_LEAKY:
        return torch.where(y > 0, torch.ones_like(y), torch.full_like(y, slope))
    if act == ACT_TANH:
        return 1 - y * y
    return torch.ones_like(y)


class LaunchAudit:
    def __init__(self, tol=2e-4, verbose=False):
        self.tol = tol
        self.verbose = verbose
        self.checked = {"fwd": 0, "bwd_data": 0, "wgrad": 0}
        self.f16_checked = 0
        self.worst = 0.0
        self.log = []

    # ------------------------------------------------------------------
    def _cmp(self, got, ref, what, tol=None):
        tol = self.tol if tol is None else tol
        got, ref = got.double(), ref.double()
        assert got.shape == ref.shape, (what, got.shape, ref.shape)
        assert not torch.isnan(got).any(), f"{what}: NaN in the kernel's result"
        scale = ref.abs().max().item() + 1e-30
        err = (got - ref).abs().max().item()
        self.worst = max(self.worst, err / scale)
        self.log.append((what, err / scale))
        if self.verbose:
            print(f"audit {what}: {err / scale:.2e}")
        assert err <= tol * scale, f"launch audit: {what}: max err {err:.3e} vs scale {scale:.3e}"

    @staticmethod
    def _runs_f16(g, which, ep):
        """does this launch multiply fp16-rounded operands (include/ali_hip.h: ali_conv_uses_f16)"""
        import ctypes
        import ali_hip
        return bool(ali_hip.load().ali_conv_uses_f16(ctypes.byref(g), which, ctypes.byref(ep)))

    # ------------------------------------------------------------------
    def gemm(self, kind, g, a, w, out, ep, in_ld, out_ld):
        torch.cuda.synchronize()
        B, H, W, C, P, Q, K, R, S, stride, pad = _geom(g)
        refs = getattr(ep, "refs", {})
        if kind == "fwd":          # a = x [B,H,W,C], w = [K][R*S][C], out = y [B,P,Q,K]
            f16 = self._runs_f16(g, 0, ep)
            x = a.detach().reshape(B, H, W, C).cpu()
            wt = w.detach().cpu().reshape(K, R, S, C).permute(0, 3, 1, 2)
            if f16:
                x, wt = x.half().float(), wt.half().float()
            ref = F.conv2d(x.permute(0, 3, 1, 2), wt, stride=stride, padding=pad).permute(0, 2, 3, 1)
            n_out, oshape = K, (B, P, Q, K)
        else:                      # a = dy [B,P,Q,K], w = [C][R*S][K], out = dx [B,H,W,C]
            f16 = self._runs_f16(g, 1, ep)
            dy = a.detach().reshape(B, P, Q, K).cpu()
            wt = w.detach().cpu().reshape(C, R, S, K).permute(3, 0, 1, 2)          # [K][C][R][S]
            if f16:
                dy, wt = dy.half().float(), wt.half().float()
            opad = (H - ((P - 1) * stride - 2 * pad + R), W - ((Q - 1) * stride - 2 * pad + S))
            ref = F.conv_transpose2d(dy.permute(0, 3, 1, 2), wt, stride=stride, padding=pad,
                                     output_padding=opad).permute(0, 2, 3, 1)
            n_out, oshape = C, (B, H, W, C)
        if refs.get("bias") is not None:
            ref = ref + refs["bias"].detach().cpu().reshape(1, 1, 1, -1)[..., :n_out]
        ref = _act(ref, int(ep.act), float(ep.slope))
        if refs.get("mask") is not None:
            ref = ref * refs["mask"].detach().cpu()[:, :n_out].reshape(oshape[0], 1, 1, n_out)
        if refs.get("dact_y") is not None:
            src = refs["dact_y16"].float() if refs.get("dact_y16") is not None else refs["dact_y"]   # (fp16 path: the twin)
            yprev = src.detach().reshape(oshape).cpu()
            ref = ref * _dact(yprev, int(ep.dact), float(ep.dslope))
        got = out.detach().reshape(oshape).cpu()
        self._cmp(got, ref, f"{kind} {(B, H, W, C, P, Q, K, R, stride, pad)}{' f16' if f16 else ''}")
        self.checked[kind] += 1
        self.f16_checked += int(f16)

    # ------------------------------------------------------------------
    def scatter(self, x, w_nc, bias, out, dims, act, slope):
        """ali_tconv_scatter: ConvTranspose2d(C -> NC) of x with the tap matrix w_nc [(r*S+s)*NC + c][C], exact fp32"""
        torch.cuda.synchronize()
        B, H, W, C, Hout, Wout, NC, ostride, R, S, stride, pad = dims
        xs = x.detach().reshape(B, H, W, C).cpu().permute(0, 3, 1, 2)
        wt = w_nc.detach().cpu().reshape(R, S, NC, C).permute(3, 2, 0, 1)                    # [C][NC][R][S]
        opad = (Hout - ((H - 1) * stride - 2 * pad + R), Wout - ((W - 1) * stride - 2 * pad + S))
        ref = F.conv_transpose2d(xs, wt, stride=stride, padding=pad, output_padding=opad).permute(0, 2, 3, 1)
        if bias is not None:
            ref = ref + bias.detach().cpu().reshape(1, 1, 1, -1)[..., :NC]
        ref = _act(ref, int(act), float(slope))
        got = torch.as_strided(out.detach(), (B, Hout, Wout, NC), (Hout * Wout * ostride, Wout * ostride, ostride, 1),
                               out.storage_offset()).cpu()
        self._cmp(got, ref, f"scatter {dims}")
        self.checked["fwd"] += 1

    # ------------------------------------------------------------------
    def wgrad(self, g, x, dy, dst, cg_log, cd_log, strides, db, dy_ld):
        geom = _geom(g)
        from ali_hip import ops
        f16 = bool(ops._PRECISION["f16"]) and geom[3] % 4 == 0 and geom[6] % 4 == 0

        def done():
            torch.cuda.synchronize()
            B, H, W, C, P, Q, K, R, S, stride, pad = geom
            xs = x.detach().reshape(B, H, W, C).cpu()
            dys = dy.detach().reshape(B, P, Q, K).cpu()
            dy_sum = dys.sum(dim=(0, 1, 2))
            if f16:
                xs, dys = xs.half().float(), dys.half().float()
            ref = torch.nn.grad.conv2d_weight(xs.permute(0, 3, 1, 2), (K, C, R, S), dys.permute(0, 3, 1, 2),
                                              stride=stride, padding=pad)[:cd_log, :cg_log].reshape(cd_log, cg_log, R * S)
            got = torch.as_strided(dst.detach(), (cd_log, cg_log, R * S), strides, dst.storage_offset()).cpu()
            # the pixel reduction of the spectrogram layers is 1e5-1e6 terms long: summation-order noise ~ sqrt(n) * 2^-24
            self._cmp(got, ref, f"wgrad {geom}{' f16' if f16 else ''}", tol=max(self.tol, 5e-4 if f16 else 0))
            if db is not None:   # (fp16: the column sums of the fp32 operand or of its fp16 twin, whichever the launch read)
                self._cmp(db.detach().cpu(), dy_sum[:cd_log], f"wgrad bias {geom}", tol=4e-3 if f16 else 1e-4)
            self.checked["wgrad"] += 1
            self.f16_checked += int(f16)
        return done
