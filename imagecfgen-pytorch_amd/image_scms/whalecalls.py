"""Whale-call spectrogram BiGAN -- drop-in for the reference's ``image_scms/whalecalls.py``
(constants :14-20, init_weights :23-28, Encoder :230-271, Generator :274-321, Discriminator :324-387,
train :390-569).  256x256 images, one categorical attribute (call_type; 'time' and 'path' are carried by the
batches but never used by the models)."""
import torch
import torch.nn as nn

from . import _spect
from ._spect import init_weights  # noqa: F401

ATTRIBUTE_DIMS = {
    "call_type": 3,
    "path": 1,
    "time": 2
}
IMAGE_SHAPE = (256, 256)
LATENT_DIM = 512
_USED = {k: v for k, v in ATTRIBUTE_DIMS.items() if k not in ["time", "path"]}
_KEYS = tuple(sorted(_USED.keys()))

WhaleCallData = _spect.data_adapter_unavailable("WhaleCallData", "torchaudio, scipy.io wav/mat files")


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
        self.embedding_dict = nn.ModuleDict({k: _spect.plane_embedding(v, 16) for k, v in _USED.items()})
        self.layers = _spect.conv_stack(2, [1, 2, 4, 8, 16, 16, None], d)


class Generator(_Family, _spect.SpectGenerator):
    def __init__(self, d=64):
        super().__init__()
        self.embedding_dict = nn.ModuleDict({k: nn.Embedding(v, 256) for k, v in _USED.items()})
        self.layers = _spect.deconv_stack(LATENT_DIM + 256, [16, 8, 4, 2, 1, None], d)


class Discriminator(_Family, _spect.SpectDiscriminator):
    def __init__(self, d=64):
        super().__init__()
        self.embedding_dict = nn.ModuleDict({k: _spect.plane_embedding(v, 16) for k, v in _USED.items()})
        self.dz = _spect.dz_stack()
        self.dx = _spect.conv_stack(2, [1, 2, 2, 4, 8, 16, None], d)
        self.dxz = _spect.dxz_stack()


STFT = dict(n_fft=511, win_length=128, hop_length=24, pad=64)     # WhaleCallData.audio_to_spectrogram (reference :52-55)


def train(nocall_directory,
          gunshot_directory=None,
          upcall_directory=None,
          n_epochs=200,
          l_rate=1e-4,
          device='cpu',
          save_images_every=2,
          batch_size=32,
          image_output_path='',
          filter_length=None,
          checkpoint_every=None,
          checkpoint_path=None):
    """Reference signature (:390-399) and loop (:426-499): statistics pass, ``spect_to_img``, BiGAN iterations with
    the attributes cast to int (:455).  ``nocall_directory`` may be a data source with the adapter's interface
    (``_spect.WaveformData(..., **STFT)``) instead of the three NARW recording directories, which need the reference's
    ``WhaleCallData`` reader (torchaudio, scipy wav/mat files; raises ImportError here)."""
    E, G, D = Encoder().to(device), Generator().to(device), Discriminator().to(device)
    for m in (E, G, D):
        m.apply(init_weights)
    if _spect.is_data_source(nocall_directory):
        data = nocall_directory
    else:
        data = WhaleCallData(nocall_directory, gunshot_directory, upcall_directory, device=device,
                             filter_length=filter_length)
    return _spect.run_training(E, G, D, data, dict(batch_size=batch_size), _KEYS, n_epochs, l_rate, device,
                               attr_cast=torch.int32, checkpoint_every=checkpoint_every,
                               checkpoint_path=checkpoint_path)
