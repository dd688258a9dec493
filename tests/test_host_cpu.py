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
