"""AudioMNIST spectrogram ALI -- drop-in for the reference's ``image_scms/audio_mnist.py``
(constants :18-30, init_weights :33-38, Encoder :173-210, Generator :213-256, Discriminator :259-318,
train :321-482).  128x128 log-spectrogram images, six categorical attributes rendered as tanh'd embedding planes.
Models and the training loop run on the HIP kernels for CUDA tensors; the dataset adapter (zip + torchaudio
front-end, :41-170) is the step before the hot path and is not re-implemented."""
import numpy as np
import torch
import torch.nn as nn

from . import _spect
from ._spect import init_weights  # noqa: F401  (std=0.001, reference :33)
from .training_utils import AdversariallyLearnedInference  # noqa: F401

np.random.seed(42)   # the reference seeds numpy at import time (:17)
VALIDATION_RUNS = [38, 7, 42, 10, 14, 18, 20, 22, 28]

LATENT_DIM = 512
ATTRIBUTE_COUNT = 47
IMAGE_SHAPE = (128, 128)
ATTRIBUTE_DIMS = {
    "country_of_origin": 13,
    "native_speaker": 2,
    "accent": 15,
    "digit": 10,
    "age": 5,
    "gender": 2
}
_KEYS = tuple(sorted(ATTRIBUTE_DIMS.keys()))

AudioMNISTData = _spect.data_adapter_unavailable("AudioMNISTData", "torchaudio, librosa")


class _Family:
    image_hw = IMAGE_SHAPE
    cat_keys = _KEYS
    cont_key = None

    def plane_module(self, k):
        return self.embedding_dict[k]

    def table(self, k):
        return self.embedding_dict[k]


class Encoder(_Family, _spect.SpectEncoder):
    def __init__(self, d=64):
        super().__init__()
        self.embedding_dict = nn.ModuleDict({k: _spect.plane_embedding(v, 8) for k, v in ATTRIBUTE_DIMS.items()})
        self.layers = _spect.conv_stack(len(ATTRIBUTE_DIMS) + 1, [1, 2, 4, 8, 16, None], d)


class Generator(_Family, _spect.SpectGenerator):
    def __init__(self, d=64):
        super().__init__()
        self.embedding_dict = nn.ModuleDict({k: nn.Embedding(v, 256) for k, v in ATTRIBUTE_DIMS.items()})
        self.layers = _spect.deconv_stack(LATENT_DIM + 256 * len(ATTRIBUTE_DIMS), [8, 4, 2, 1, None], d)


class Discriminator(_Family, _spect.SpectDiscriminator):
    def __init__(self, d=64):
        super().__init__()
        self.embedding_dict = nn.ModuleDict({k: _spect.plane_embedding(v, 8) for k, v in ATTRIBUTE_DIMS.items()})
        self.dz = _spect.dz_stack()
        self.dx = _spect.conv_stack(len(ATTRIBUTE_DIMS) + 1, [1, 2, 4, 8, 16, None], d)
        self.dxz = _spect.dxz_stack()


STFT = dict(n_fft=255, win_length=128, pad=96)     # AudioMNISTData.audio_to_spectrogram (reference :59-61)


def train(path_to_zip,
          n_epochs=200,
          l_rate=1e-4,
          device='cpu',
          save_images_every=2,
          batch_size=128,
          image_output_path='',
          checkpoint_every=None,
          checkpoint_path=None):
    """Reference signature (:321-327) and loop (:343-420): statistics pass over the training stream, ``spect_to_img``
    standardisation, ALI iterations with Adam betas (0.5, 0.9).

    ``path_to_zip`` is either the AudioMNIST zip (needs the reference's ``AudioMNISTData`` reader: torchaudio +
    librosa, not part of this package -- raises ImportError) or a data source with the adapter's interface, e.g.
    ``_spect.WaveformData(waveforms, attrs, **STFT, device=device, runs=...)``: then the whole loop, spectrograms
    included, runs on the device.  The demo-image / wav dump of the reference (:422-480, matplotlib + Griffin-Lim) is
    not part of the path."""
    E, G, D = Encoder().to(device), Generator().to(device), Discriminator().to(device)
    for m in (E, G, D):
        m.apply(init_weights)
    data = path_to_zip if _spect.is_data_source(path_to_zip) else AudioMNISTData(path_to_zip, device=device)
    keys = [k for k in data.data if k in ATTRIBUTE_DIMS]
    return _spect.run_training(E, G, D, data, dict(batch_size=batch_size, excluded_runs=VALIDATION_RUNS), keys,
                               n_epochs, l_rate, device, checkpoint_every=checkpoint_every,
                               checkpoint_path=checkpoint_path)
