"""debug: fp16 shadow-operand path vs in-flight conversion: E+G gradients of one EG phase (ESRF d=8)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("imagecfgen-pytorch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
torch.set_num_threads(8)
import ali_oracle as orc
from test_gpu_modules import paired_models, to_dev
from ali_hip import ops
from ali_hip.step import AliStepper
fam, d, B = sys.argv[1] if len(sys.argv) > 1 else "esrf", 8, 2
(Eo, Go, Do), (E, G, D), images, c, z = paired_models(fam, d=d, B=B)
for m in (Eo, Go, Do, E, G, D): m.train()
oe, od = orc.build_optimizers(Eo, Go, Do, fam)
st = AliStepper(E, G, D, betas=(0.5, 0.9), precision="f16")
res = {}
orig = ops.shadow16
for mode in ("cvt", "mem"):
    ops.shadow16 = (lambda t: None) if mode == "cvt" else orig
    import ali_hip.chain as ch
    with torch.no_grad(), ops.precision("f16"):
        st.load_state(Eo, Go, Do, oe, od)
        cx = st._begin(images.cuda(), to_dev(c), z.cuda())
        st._eg_grads(cx)
    res[mode] = (st.opt_eg.grad.double().cpu().clone(), cx["out"]["loss_eg"].item())
ops.shadow16 = orig
g0, g1 = res["cvt"][0], res["mem"][0]
print("loss", res["cvt"][1], res["mem"][1])
print("rel L2 mem vs cvt", ((g1 - g0).norm() / g0.norm()).item())
off = 0
for nm, m in (("E", E), ("G", G)):
    for k, p in m.named_parameters():
        n = p.numel()
        a, b = g0[off:off + n], g1[off:off + n]
        print(f"{nm}.{k:28s} rel {((a-b).norm()/(a.norm()+1e-300)).item():.2e}")
        off += n

# ---- which launch reads a twin that is not the rounded fp32 tensor?
real = ops._f16_operands
bad = []
def checked(g, which, x, w_packed, y, ep):
    x16, w16 = ops.shadow16(x), ops.shadow16(w_packed)
    if x16 is not None and w16 is not None:
        ex = bool(torch.equal(x16, x.half()))
        ew = bool(torch.equal(w16, w_packed.half()))
        if not (ex and ew):
            bad.append((which, tuple(x.shape), tuple(w_packed.shape), ex, ew,
                        float((x16.float() - x).abs().max()), float((w16.float() - w_packed).abs().max())))
    return real(g, which, x, w_packed, y, ep)
ops._f16_operands = checked
with torch.no_grad(), ops.precision("f16"):
    st.load_state(Eo, Go, Do, oe, od)
    cx = st._begin(images.cuda(), to_dev(c), z.cuda())
    st._eg_grads(cx)
print("launches reading a stale twin:", len(bad))
for b in bad[:20]:
    print(b)

# ---- per launch: fp16-in-memory result vs converting path on the same inputs
ops._f16_operands = real
import ali_hip.chain as chain
mism = []
def wrap(name, which):
    orig_fn = getattr(ops, name)
    def f(g, x, w, y, ep):
        has = ops.shadow16(x) is not None and ops.shadow16(w) is not None
        out = orig_fn(g, x, w, y, ep)
        if has:
            y2 = torch.empty_like(y)
            sx, sw = x._ali16, w._ali16
            del x._ali16, w._ali16
            ep2 = ops.AliEpilogue()
            for fld, _ in ops.AliEpilogue._fields_:
                setattr(ep2, fld, getattr(ep, fld))
            ep2.in16 = ep2.w16 = ep2.out16 = None
            orig_fn(g, x, w, y2, ep2)
            x._ali16, w._ali16 = sx, sw
            err = float((y - y2).abs().max()); sc = float(y2.abs().max())
            if err > 1e-5 * sc:
                mism.append((name, [getattr(g, n) for n, _ in ops.AliConvGeom._fields_], err, sc,
                             bool(ep.mask), bool(ep.dact_y), bool(ep.bias), ep.act))
        return out
    setattr(ops, name, f)
wrap("conv_fwd", 0); wrap("conv_bwd_data", 1)
with torch.no_grad(), ops.precision("f16"):
    st.load_state(Eo, Go, Do, oe, od)
    cx = st._begin(images.cuda(), to_dev(c), z.cuda())
    st._eg_grads(cx)
print("mismatching launches:", len(mism))
for m in mism[:30]:
    print(m)
