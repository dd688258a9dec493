"""GPU parity of the caller-facing paths around the training iteration:

* ``FinetuneStepper`` -- the three ``finetune_*_bigan.py`` flavours (pickled modules / ``load_state_dict`` / whole
  batch dict as attributes) against the oracle executing the reference's statements, eager and HIP-graph replay;
* ``GeneratorSampler`` -- the ``*_generator_score.py`` loops against the ORACLE's Generator loop;
* ``WaveformData`` + the three ``train`` entry points fed tensors: spectrogram statistics pass, ``spect_to_img`` and
  the ALI loop on the device (audio_mnist.py:343-420, whalecalls.py:426-499, esrf_acoustic.py:298-379);
* ``load_model`` and periodic resumable checkpoints (mnist.py:302-313).
"""
import copy

import numpy as np
import pytest
import torch

import ali_oracle as orc
from test_gpu_modules import close, paired_models, to_dev

pytestmark = pytest.mark.gpu


def _reference_finetune(E, G, x, a, lr, steps):
    """finetune_audio_mnist_bigan.py:62-91 / finetune_whale_bigan.py:52-77 (metric mse), verbatim statements."""
    E.train(), G.eval()
    opt = torch.optim.Adam(E.parameters(), lr=lr)
    rec, lat = [], []
    for _ in range(steps):
        opt.zero_grad()
        codes = E(x, a)
        xr = G(codes, a)
        rec_loss = torch.square(x - xr).mean()
        latent = torch.square(codes).mean()
        (rec_loss + latent).backward()
        opt.step()
        rec.append(rec_loss.item()), lat.append(latent.item())
    return rec, lat


def _check_encoder_update(Eo, E, before, lr, steps):
    so = Eo.state_dict()
    wo = torch.cat([(so[k] - before[k]).reshape(-1).double() for k in so])
    wp = torch.cat([(v.cpu() - before[k]).reshape(-1).double() for k, v in E.state_dict().items()])
    err = (wp - wo).abs()
    assert err.max().item() <= 2.2 * lr * steps
    assert (err > 0.1 * lr).double().mean().item() < 1e-2, (err > 0.1 * lr).double().mean().item()
    assert err.mean().item() <= 0.02 * lr * steps, err.mean().item()


@pytest.mark.parametrize("capture", [False, True])
def test_finetune_stepper_audio_state_dict_flavour(capture):
    """finetune_audio_mnist_bigan.py:57-91: modules rebuilt from ``E_state_dict`` / ``G_state_dict``, attributes
    ``{k: batch[k].float()}``, x = spect_to_img(images) of shape [B,1,128,128]."""
    import image_scms.audio_mnist as pm
    from ali_hip.step import FinetuneStepper
    (Eo, Go, _), _, images, c, _ = paired_models("audio", d=8, B=2)
    model_dict = {"E_state_dict": copy.deepcopy(Eo.state_dict()), "G_state_dict": copy.deepcopy(Go.state_dict())}
    E, G = pm.Encoder(8).cuda(), pm.Generator(8).cuda()
    E.load_state_dict(model_dict["E_state_dict"])
    G.load_state_dict(model_dict["G_state_dict"])
    E.train(), G.eval()
    before = copy.deepcopy(Eo.state_dict())
    lr, steps = 1e-4, 3
    rec, lat = _reference_finetune(Eo, Go, images, c, lr, steps)
    ft = FinetuneStepper(E, G, lr=lr, capture=capture)
    out = []
    for _ in range(steps):                               # (a replayed graph returns the same output tensors)
        r = ft.step(images.cuda(), to_dev(c))
        out.append((r["rec"].item(), r["latent"].item()))
    np.testing.assert_allclose([o[0] for o in out], rec, rtol=2e-4)
    np.testing.assert_allclose([o[1] for o in out], lat, rtol=2e-4)
    _check_encoder_update(Eo, E, before, lr, steps)
    if capture:
        assert len(ft._graphs) == 1


@pytest.mark.parametrize("capture", [False, True])
def test_finetune_stepper_whale_whole_batch_dict(capture):
    """finetune_whale_bigan.py:52-77: the WHOLE batch dict (extra keys 'audio', 'path', 'time'; int one-hots) is passed
    as the attributes, and x stays [B,256,256], so ``square(x - xr)`` broadcasts against xr [B,1,256,256] to all B*B
    pairs -- reproduced as executed."""
    from ali_hip.step import FinetuneStepper
    (Eo, Go, _), (E, G, _), images, c, _ = paired_models("whale", d=8, B=2)
    x = images.reshape(2, 256, 256)
    batch = {"audio": x, "call_type": c["call_type"].int(), "path": ["a.wav", "b.wav"],
             "time": torch.tensor([[0.0, 2.0], [1.0, 3.0]])}
    before = copy.deepcopy(Eo.state_dict())
    lr, steps = 1e-5, 2
    rec, lat = _reference_finetune(Eo, Go, x, batch, lr, steps)
    E.train(), G.eval()
    ft = FinetuneStepper(E, G, lr=lr, capture=capture)
    dev_batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    out = []
    for _ in range(steps):
        r = ft.step(x.cuda(), dev_batch)
        out.append((r["rec"].item(), r["latent"].item()))
    np.testing.assert_allclose([o[0] for o in out], rec, rtol=2e-4)
    np.testing.assert_allclose([o[1] for o in out], lat, rtol=2e-4)
    _check_encoder_update(Eo, E, before, lr, steps)


@pytest.mark.parametrize("family,d,B,R", [("mnist", 64, 4, 1), ("mnist", 64, 4, 3), ("audio", 8, 2, 3)])
def test_generator_sampler_vs_oracle_loop(family, d, B, R):
    """mnist_generator_score.py:69-74 (one round) and audiomnist_generator_score.py:83-88 (mean over ``mc_rounds``
    generations, G.eval()) executed by the ORACLE's Generator on the CPU vs the batched, graph-replayed sampler."""
    from ali_hip.step import GeneratorSampler
    (_, Go, _), (_, G, _), _, c, _ = paired_models(family, d=d, B=B)
    Go.eval(), G.eval()
    zs = torch.randn(R, B, 512, 1, 1, generator=torch.Generator().manual_seed(17))
    with torch.no_grad():
        gen = 0
        for r in range(R):
            gen = gen + Go(zs[r], c)
        gen = gen / R
    sampler = GeneratorSampler(G)
    for _ in range(2):                                   # second call: pure graph replay
        out = sampler(zs.cuda(), to_dev(c))
    close(out, gen.reshape(out.shape), what=f"{family} sampler R={R}")


def _cpu_log_spectrogram(wave, n_fft, win_length, hop_length=None, pad=0):
    """(torchaudio.transforms.Spectrogram(...)(x) + 1e-6).log() restated with torch.stft (torchaudio's documented
    mapping: two-sided zero pad, centred reflect-padded frames, periodic Hann window, power 2)."""
    hop_length = hop_length or win_length // 2
    x = torch.nn.functional.pad(wave.double(), (pad, pad))
    spec = torch.stft(x, n_fft, hop_length=hop_length, win_length=win_length,
                      window=torch.hann_window(win_length, dtype=torch.float64), center=True, pad_mode="reflect",
                      return_complex=True).abs().pow(2.0)
    return (spec + 1e-6).log().float()


@pytest.mark.parametrize("mod_name,L", [("audio_mnist", 8000), ("whalecalls", 6000), ("esrf_acoustic", 40000)])
def test_waveform_data_stream_vs_cpu_restatement(mod_name, L):
    """Statistics pass (audio_mnist.py:347-359) and ``spect_to_img`` (:361-363) over a tensor-in stream, spectrograms
    from the device front-end, vs the same statements on torch.stft spectrograms."""
    import importlib
    from image_scms import _spect
    pm = importlib.import_module(f"image_scms.{mod_name}")
    g = torch.Generator().manual_seed(3)
    n, bs = 6, 4
    wave = torch.randn(n, L, generator=g) * torch.linspace(0.2, 2.0, n).reshape(n, 1)
    data = _spect.WaveformData(wave, {}, **pm.STFT, device="cuda")
    stream = lambda: data.stream(batch_size=bs, shuffle=False)  # noqa: E731
    mean, std, nb = _spect.spectrogram_statistics(stream, "cuda")
    ref = _cpu_log_spectrogram(wave, **pm.STFT)
    H, W = pm.Encoder.image_hw
    assert ref.shape == (n, H, W) and nb == 2
    m_ref = s_ref = 0
    for lo in range(0, n, bs):
        m_ref = m_ref + ref[lo:lo + bs].mean(dim=(0, 1)).reshape(1, 1, -1)
        s_ref = s_ref + ref[lo:lo + bs].square().mean(dim=(0, 1)).reshape(1, 1, -1)
    m_ref, s_ref = m_ref / nb, s_ref / nb
    std_ref = torch.sqrt(torch.clamp_min(s_ref - m_ref.square(), 0.0))
    close(mean, m_ref, 1e-5, "spect_mean")
    # frames inside the zero ``pad`` margin are constant (log 1e-6): their variance is pure fp32 cancellation noise
    # in the reference's E[X^2] - E[X]^2 formula -- compared only where the variance is real
    real = (std_ref.reshape(-1) > 0.05)
    assert real.sum() >= W - 8 and torch.isfinite(std).all()
    close(std.reshape(-1).cpu()[real], std_ref.reshape(-1)[real], 1e-4, "spect_std")
    data.fuse_spect_to_img(mean, std)
    img = torch.cat([b["audio"] for b in stream()]).cpu()
    assert torch.isfinite(img).all() and img.abs().max() <= 1.0
    img_ref = torch.clip((ref - m_ref) / (std_ref + 1e-6), -3, 3) / 3.0
    close(img[..., real], img_ref[..., real], 2e-4, "spect_to_img")


@pytest.mark.parametrize("mod_name,L,n,bs", [("audio_mnist", 8000, 6, 4), ("whalecalls", 6000, 4, 2),
                                              ("esrf_acoustic", 40000, 2, 2)])
def test_train_entry_points_run_on_tensors(mod_name, L, n, bs, tmp_path):
    """``audio_mnist.train`` / ``whalecalls.train`` / ``esrf_acoustic.train`` -- the reference's own entry points and
    default widths (d=64) -- given a tensor data source: raw waveforms -> device spectrograms -> statistics ->
    spect_to_img -> graph-replayed ALI iterations (ragged last batch for audio), periodic checkpoint."""
    import importlib
    import ali_hip
    from ali_hip import ops
    from image_scms import _spect
    pm = importlib.import_module(f"image_scms.{mod_name}")
    ali_hip.manual_seed(2)
    torch.manual_seed(2)
    np.random.seed(2)
    g = torch.Generator().manual_seed(8)
    wave = torch.randn(n, L, generator=g)
    attrs = {}
    for k, v in pm.ATTRIBUTE_DIMS.items():
        if k in ("path", "time"):
            continue
        if k == "closest_boat":
            attrs[k] = torch.rand(n, 1, generator=g) * 2 - 1
        else:
            attrs[k] = torch.nn.functional.one_hot(torch.randint(0, v, (n,), generator=g), v).float()
    data = _spect.WaveformData(wave, attrs, **pm.STFT, device="cuda", runs=np.arange(n) + 100)
    ops.set_workspace_bytes(1 << 30)
    try:
        ck = tmp_path / "ck.tar"
        E, G, D, oD, oE = pm.train(data, n_epochs=1, device="cuda", batch_size=bs, save_images_every=None,
                                   checkpoint_every=1, checkpoint_path=str(ck))
    finally:
        ops.set_workspace_bytes(256 << 20)
    assert all(torch.isfinite(p).all() for m in (E, G, D) for p in m.parameters())
    assert oD.state_dict()["step"] == 2 * ((n + bs - 1) // bs) and oE.state_dict()["step"] == (n + bs - 1) // bs
    sd = torch.load(ck)
    assert set(sd["E_state_dict"]) == set(E.state_dict()) and sd["optimizer_D"]["step"] == oD.state_dict()["step"]
    for k, v in E.state_dict().items():
        assert torch.equal(sd["E_state_dict"][k], v.cpu()), k
    torch.manual_seed(2)
    fresh = pm.Encoder()
    fresh.apply(pm.init_weights)
    moved = sum(float((p.detach().cpu() - q.detach()).abs().sum()) for p, q in zip(E.parameters(), fresh.parameters()))
    assert moved > 0


def test_periodic_checkpoint_load_model_and_resume(tmp_path):
    """mnist.train(checkpoint_every=...) writes the state-dict format ``mnist.load_model`` reads (mnist.py:302-313);
    the file also resumes an ``AliStepper`` (weights, BatchNorm buffers, Adam moments and step counts)."""
    import ali_hip
    import image_scms.mnist as pm
    from ali_hip.step import AliStepper
    ali_hip.manual_seed(4)
    torch.manual_seed(4)
    np.random.seed(4)
    x, a = orc.synth_morphomnist(128, seed=3)
    ck = tmp_path / "mnist-ck.tar"
    E, G, D, oD, oE = pm.train(x, a, n_epochs=2, device="cuda", save_images_every=None, batch_size=64,
                               checkpoint_every=2, checkpoint_path=str(ck))
    E2, G2, D2, raw = pm.load_model(str(ck), return_raw=True)
    for m, m2 in ((E, E2), (G, G2), (D, D2)):
        for (k, v), (_, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
            assert torch.equal(v.cpu(), v2), k
    assert raw["optimizer_D"]["step"] == 8 and raw["optimizer_E"]["step"] == 4
    st = AliStepper(E2.cuda(), G2.cuda(), D2.cuda(), capture=False)
    st.load_state_dict(raw)
    # (checkpoints hold the Adam moments in the parameters' own element order, whatever layout the stepper keeps them in)
    assert torch.equal(st.opt_d.logical(st.opt_d.m_views).cpu(), raw["optimizer_D"]["exp_avg"]) and int(st.opt_d.step_t) == 8
    assert torch.equal(st.opt_eg.flat, oE.flat) and torch.equal(st.opt_d.v, oD.v)
