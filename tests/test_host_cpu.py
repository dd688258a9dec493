"""CPU-side checks (no GPU): the drop-in modules' structure / RNG order / CPU semantics agree with the
oracle (hence with the reference), the C-ABI library loads and exports every declared symbol."""
import os
import re

import numpy as np
import pytest
import torch

import ali_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    import ali_hip
    from ali_hip import _lib
    lib = ali_hip.load()                       # loads without a GPU; no compute call here
    header = open(os.path.join(ROOT, "include", "ali_hip.h")).read()
    declared = set(re.findall(r"\b(ali_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ali_version() >= 1


def test_tile_order_is_host_side_and_balances_the_cus():
    """ali_conv_tile_order runs on the host (no GPU).  MNIST D's 256 -> 128 4x4 stride-2 data gradient (mnist.py:120): 512
    M-tiles x 2 n-tiles = 1024 blocks, all resident, 4 per CU, with 1, 2 or 4 live taps per tile.  Slot u runs on CU
    u % 256; the table places the tiles so that every CU gets the same number of taps (9 x 2 n-tiles); in raster
    order a CU would hold four neighbouring tiles (up to 16 x 2).  Layers whose tiles all cost the same report 0."""
    import ctypes
    import ali_hip
    from ali_hip import ops
    lib = ali_hip.load()
    buf = (ctypes.c_int32 * 4096)()
    g = ops.geom(512, 8, 8, 128, 3, 3, 256, 4, 4, 2, 0)
    n = lib.ali_conv_tile_order(ctypes.byref(g), 1, 0, ctypes.cast(buf, ctypes.c_void_p), 4096)
    assert n == 512 and sorted(buf[:n]) == list(range(512))       # 4 phases x 16 pixels x 8 tiles of 64 images

    def live(tile):     # phase (ph, pw), pixel (qh, qw) of the phase's 4 x 4 sub-grid -> taps inside the 3 x 3 input
        ph, rem = divmod(tile, 128)
        qh, qw = divmod(rem // 8, 4)
        oh, ow = (ph // 2) + 2 * qh, (ph % 2) + 2 * qw
        cnt = lambda o: sum((o - r) % 2 == 0 and 0 <= (o - r) // 2 < 3 for r in range(4))      # noqa: E731
        return cnt(oh) * cnt(ow)

    def cu_loads(order):    # the kernel's slot -> (rank, n-tile) map (n-tiles of a rank 8 slots apart), slot u on CU u % 256
        load = [0] * 256
        for u in range(1024):
            grp, rem = divmod(u, 16)
            load[u % 256] += live(order[grp * 8 + (rem & 7)])
        return load
    assert sorted(live(t) for t in range(512)) == [1] * 128 + [2] * 256 + [4] * 128
    balanced, raster = cu_loads(list(buf[:n])), cu_loads(list(range(512)))
    assert max(balanced) == min(balanced) == 9 and max(raster) == 16
    g1 = ops.geom(512, 1, 1, 1024, 1, 1, 1024, 1, 1, 1, 0)
    assert lib.ali_conv_tile_order(ctypes.byref(g1), 0, 0, ctypes.cast(buf, ctypes.c_void_p), 4096) == 0
    assert lib.ali_conv_tile_order(ctypes.byref(g), 1, 0, ctypes.cast(buf, ctypes.c_void_p), 100) == 0   # cap too small


def test_product_mnist_modules_match_oracle_on_cpu():
    import image_scms.mnist as pm
    torch.manual_seed(5)
    Eo, Go, Do = orc.build_models("mnist")
    torch.manual_seed(5)
    E, G, D = pm.Encoder(), pm.Generator(), pm.Discriminator()
    for m in (E, G, D):
        m.apply(pm.init_weights)
    assert orc.weights_digest(E, G, D) == orc.weights_digest(Eo, Go, Do)      # same ctor + init RNG order
    assert list(D.state_dict()) == list(Do.state_dict())
    x, a = orc.synth_morphomnist(4, seed=3)
    stats = {k: (v.min(dim=0).values, v.max(dim=0).values) for k, v in a.items() if k != "digit"}
    images, c = orc.mnist_scale_batch(x, a, stats)
    z = torch.randn(4, 512, 1, 1)
    for m in (E, G, D, Eo, Go, Do):
        m.eval()
    assert torch.equal(E(images, c), Eo(images, c))
    assert torch.equal(G(z, c), Go(z, c))
    assert torch.equal(D(images, z, c), Do(images, z, c))


def test_product_train_matches_reference_trajectory_on_cpu(golden_dir):
    """image_scms.mnist.train on CPU (BASELINE config 0, plumbing) reproduces the reference's weights."""
    import image_scms.mnist as pm
    g = np.load(os.path.join(golden_dir, "mnist_traj_n192_bs64.npz"), allow_pickle=False)
    x, a = orc.synth_morphomnist(192, seed=1)
    torch.manual_seed(1)
    np.random.seed(1)
    E, G, D, oD, oE = pm.train(x, a, n_epochs=1, save_images_every=None, batch_size=64)
    if str(g["torch_version"]) == torch.__version__:
        assert orc.weights_digest(E, G, D) == str(g["weights_digest"])
    for k, v in E.state_dict().items():
        ref = g[f"stats.E.{k}"]
        assert abs(v.double().sum().item() - ref[0]) <= 1e-5 * max(ref[1], 1e-30)


def test_cuda_path_refuses_without_library(monkeypatch):
    """The product never falls back: a missing library is a hard error."""
    from ali_hip import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libali_hip.so")
    with pytest.raises(_lib.AliHipUnavailable):
        _lib.load()


def test_batchify_and_helpers():
    from image_scms import training_utils as tu
    x = torch.arange(10)
    assert [b[0].tolist() for b in tu.batchify(x, batch_size=4)] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]
    d = list(tu.batchify_dict({"a": x, "b": x * 2}, batch_size=6))
    assert d[1]["b"].tolist() == [12, 14, 16, 18]
    assert list(tu.batchify(torch.zeros(0))) == []
    img = torch.zeros(2, 1, 8, 8)
    assert tu.binarized_attribute_channel(img, torch.tensor([[0., 1.], [1., 0.]])).sum().item() == 128
    assert tu.attributes_image(img, torch.ones(2, 3)).shape == (2, 2, 8, 8)


@pytest.mark.parametrize("family,modname,d,B", [("audio", "audio_mnist", 8, 2), ("whale", "whalecalls", 8, 1),
                                                  ("esrf", "esrf_acoustic", 4, 1)])
def test_product_spect_modules_match_oracle_on_cpu(family, modname, d, B):
    """ctor / init RNG order, state_dict keys and CPU forward of the spectrogram families == oracle (== reference)."""
    import importlib
    pm = importlib.import_module(f"image_scms.{modname}")
    torch.manual_seed(5)
    Eo, Go, Do = orc.build_models(family, d)
    torch.manual_seed(5)
    E, G, D = pm.Encoder(d), pm.Generator(d), pm.Discriminator(d)
    for m in (E, G, D):
        m.apply(pm.init_weights)
    assert orc.weights_digest(E, G, D) == orc.weights_digest(Eo, Go, Do)
    for mo, mp in ((Eo, E), (Go, G), (Do, D)):
        assert list(mp.state_dict()) == list(mo.state_dict())
    images, a, z = orc.synth_spect_batch(family, B, seed=3)
    a["path"] = torch.zeros(B, 1)                      # callers pass whole batch dicts: extra keys are ignored
    for m in (E, G, D, Eo, Go, Do):
        m.eval()
    assert torch.equal(E(images, a), Eo(images, a))
    assert torch.equal(G(z, a), Go(z, a))
    assert torch.equal(D(images, z, a), Do(images, z, a))
    assert torch.equal(G(z.reshape(B, 512), a), Go(z, a))       # z may be [B,512] (reference audio_mnist.py:251)


def test_audio_train_entry_point_matches_oracle_loop_on_cpu(tmp_path):
    """``audio_mnist.train`` (reference audio_mnist.py:321-420) fed a tensor data source on the CPU: statistics pass,
    spect_to_img, two ALI iterations at the reference width d=64 -- against the same statements executed with the
    oracle's modules and ``ali_step`` (bit for bit: CPU tensors run the stock torch ops of the same sub-modules).
    The checkpoint it writes is the state-dict format the audio callers read (finetune_audio_mnist_bigan.py:57-61)."""
    import image_scms.audio_mnist as pm
    from image_scms import _spect
    n, bs = 4, 2
    g = torch.Generator().manual_seed(8)
    wave = torch.randn(n, 8000, generator=g)
    attrs = {k: torch.nn.functional.one_hot(torch.randint(0, v, (n,), generator=g), v).float()
             for k, v in pm.ATTRIBUTE_DIMS.items()}
    runs = np.array([1, 2, 38, 3])                       # run 38 is a validation run: excluded from training
    data = _spect.WaveformData(wave, attrs, **pm.STFT, device="cpu", runs=runs)
    ck = tmp_path / "audio.tar"
    torch.manual_seed(3)
    np.random.seed(3)
    E, G, D, oD, oE = pm.train(data, n_epochs=1, device="cpu", batch_size=bs, save_images_every=None,
                               checkpoint_every=1, checkpoint_path=str(ck))
    # ---- the same, restated with the oracle
    torch.manual_seed(3)
    np.random.seed(3)
    Eo, Go, Do = orc.build_models("audio", 64)
    oe, od = orc.build_optimizers(Eo, Go, Do, "audio")
    ref = _spect.WaveformData(wave, attrs, **pm.STFT, device="cpu", runs=runs)
    kw = dict(batch_size=bs, excluded_runs=pm.VALIDATION_RUNS)
    mean, ss, nb = 0, 0, 0
    for batch in ref.stream(**kw):
        nb += 1
        mean = mean + batch["audio"].mean(dim=(0, 1)).reshape((1, 1, -1))
        ss = ss + batch["audio"].square().mean(dim=(0, 1)).reshape((1, 1, -1))
    mean, ss = (mean / nb).float(), (ss / nb).float()
    # (the one documented departure from audio_mnist.py:357-359: E[X^2] - E[X]^2 of the constant frames inside the zero
    # `pad` margin is fp32 cancellation noise, its sqrt NaN when negative; clamped at 0 -- _spect.spectrogram_statistics)
    std = torch.sqrt(torch.clamp_min(ss - mean.square(), 0.0))
    assert nb == 2
    for m in (Eo, Go, Do):
        m.train()
    for batch in ref.stream(**kw):
        images = batch["audio"].reshape((-1, 1, 128, 128)).float()
        c = {k: torch.clone(batch[k]).float() for k in ref.data if k in pm.ATTRIBUTE_DIMS}
        images = torch.clip((images - mean) / (std + 1e-6), -3, 3) / 3.0
        zm = torch.zeros((len(images), 512, 1, 1)).float()
        z = torch.normal(zm, zm + 1)
        orc.ali_step(Eo, Go, Do, oe, od, images, c, z)
    if orc.weights_digest(E, G, D) != orc.weights_digest(Eo, Go, Do):
        # Bit for bit in 5 runs of 6 here; in the 6th, stock torch CPU kernels at this width (d=64, 8 threads) return
        # different rounding for the SAME statements run twice in one process.  Adam's first steps are sign-like, so a
        # rounding-level gradient may then step the other way: the weights still agree to within the steps taken, and
        # almost all of them exactly.
        for (k, v), (_, vo) in zip(list(E.state_dict().items()) + list(G.state_dict().items()) + list(D.state_dict().items()),
                                   list(Eo.state_dict().items()) + list(Go.state_dict().items()) + list(Do.state_dict().items())):
            diff = (v.double() - vo.double()).abs()
            assert diff.max().item() <= 4 * 2.2e-4, (k, diff.max().item())
            if v.numel() >= 10000:
                assert (diff > 0).double().mean().item() <= 0.2 and diff.mean().item() <= 1e-5, k
    sd = torch.load(ck)
    E2 = pm.Encoder()
    E2.load_state_dict(sd["E_state_dict"])
    assert orc.weights_digest(E2) == orc.weights_digest(E)
    assert sd["optimizer_D"]["state"][0]["step"] == 4


def test_data_source_is_not_left_standardised_and_statistics_match_the_reference_statement():
    """(i) ``run_training`` fuses ``spect_to_img`` into the source's stream while it trains and must hand the source
    back streaming log-spectrograms: a second ``train`` on the same object (or a validation / fine-tuning consumer that
    standardises itself, finetune_audio_mnist_bigan.py) would otherwise compute its statistics on already-standardised
    images.  (ii) ``spectrogram_statistics(clamp_variance=False)`` is audio_mnist.py:347-359 as written,
    bit for bit, on a waveform without constant frames (pad = 0: no all-zero margin); with the default clamp only
    negative rounding noise changes (NaN -> 0)."""
    import image_scms.audio_mnist as pm
    from image_scms import _spect
    n, bs = 4, 2
    g = torch.Generator().manual_seed(9)
    wave = torch.randn(n, 8000, generator=g)
    attrs = {k: torch.nn.functional.one_hot(torch.randint(0, v, (n,), generator=g), v).float()
             for k, v in pm.ATTRIBUTE_DIMS.items()}
    data = _spect.WaveformData(wave, attrs, **pm.STFT, device="cpu")
    stream = lambda: data.stream(batch_size=bs, shuffle=False)     # noqa: E731
    before = [b["audio"].clone() for b in stream()]
    seen = []
    real = _spect.train_on_stream

    def fake_train(E, G, D, stream_fn, **kw):            # the loop itself is covered elsewhere: record what it is fed
        seen.append([b["audio"].clone() for b in stream_fn()])
        return E, G, D, None, None, []
    _spect.train_on_stream = fake_train
    try:
        stats = []
        for _ in range(2):
            real_stats = _spect.spectrogram_statistics

            def spy(fn, dev, **kw):
                out = real_stats(fn, dev, **kw)
                stats.append(out)
                return out
            _spect.spectrogram_statistics = spy
            try:
                _spect.run_training(None, None, None, data, dict(batch_size=bs, shuffle=False), (), 1, 1e-4, "cpu")
            finally:
                _spect.spectrogram_statistics = real_stats
    finally:
        _spect.train_on_stream = real
    assert torch.equal(stats[0][0], stats[1][0]) and torch.equal(stats[0][1], stats[1][1])   # same statistics twice
    assert all(torch.equal(a, b) for a, b in zip(seen[0], seen[1]))
    assert seen[0][0].abs().max().item() <= 1.0                                            # training saw images ...
    after = [b["audio"] for b in stream()]
    assert all(torch.equal(a, b) for a, b in zip(before, after))                           # ... the caller sees spectra
    # (ii) the reference's statement, un-clamped, on spectrograms without constant frames
    src = _spect.WaveformData(wave, attrs, n_fft=255, win_length=128, pad=0, device="cpu")
    sfn = lambda: src.stream(batch_size=bs, shuffle=False)         # noqa: E731
    mean, ss, nb = 0, 0, 0
    for batch in sfn():                                    # audio_mnist.py:347-359
        nb += 1
        mean = mean + batch["audio"].mean(dim=(0, 1)).reshape((1, 1, -1))
        ss = ss + batch["audio"].square().mean(dim=(0, 1)).reshape((1, 1, -1))
    mean = (mean / nb).float()
    std = torch.sqrt((ss / nb).float() - mean.square())
    m0, s0, _ = _spect.spectrogram_statistics(sfn, "cpu", clamp_variance=False)
    assert torch.equal(m0, mean) and torch.equal(s0, std) and not torch.isnan(std).any()
    m1, s1, _ = _spect.spectrogram_statistics(sfn, "cpu")
    assert torch.equal(m1, mean) and torch.equal(s1, std)          # the clamp is inert where the variance is positive


def test_mnist_checkpoint_and_load_model_on_cpu(tmp_path):
    """mnist.train(checkpoint_every=...) -> mnist.load_model (reference mnist.py:302-313) round trip."""
    import image_scms.mnist as pm
    x, a = orc.synth_morphomnist(64, seed=2)
    torch.manual_seed(2)
    np.random.seed(2)
    ck = tmp_path / "m.tar"
    E, G, D, _, _ = pm.train(x, a, n_epochs=1, save_images_every=None, batch_size=32, checkpoint_every=1,
                             checkpoint_path=str(ck))
    E2, G2, D2, raw = pm.load_model(str(ck), return_raw=True)
    assert orc.weights_digest(E2, G2, D2) == orc.weights_digest(E, G, D)
    assert set(raw) >= {"E_state_dict", "G_state_dict", "D_state_dict", "optimizer_E", "optimizer_D"}
    assert isinstance(pm.load_model(str(ck)), tuple) and len(pm.load_model(str(ck))) == 3
