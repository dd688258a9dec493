"""ESRF hydrophone spectrogram ALI -- drop-in for the reference's ``image_scms/esrf_acoustic.py``
(continuous_feature_map :13-14, constants :17-21, init_weights :24-29, Encoder :134-170, Generator :173-205,
Discriminator :208-260, train :263-447).  512x512 images; attributes: ``has_boat`` (2 classes, embedding plane)
and the continuous ``closest_boat`` (broadcast plane)."""
import torch
import torch.nn as nn

from . import _spect
from ._spect import init_weights  # noqa: F401


def continuous_feature_map(c: torch.Tensor, size: tuple = (512, 512)):
    return c.reshape((c.size(0), 1, 1, 1)).repeat(1, 1, *size)


ATTRIBUTE_DIMS = {
    "closest_boat": 1,
    "has_boat": 2
}
LATENT_DIM = 512

EsrfStation = _spect.data_adapter_unavailable("EsrfStation", "torchaudio, pandas label tables and the ESRF wav files")


class _Family:
    image_hw = (512, 512)
    cat_keys = ("has_boat",)
    cont_key = "closest_boat"

    def plane_module(self, k):
        return self.has_boat_embedding

    def table(self, k):
        return self.has_boat_embedding


class Encoder(_Family, _spect.SpectEncoder):
    def __init__(self, d=64):
        super().__init__()
        self.has_boat_embedding = _spect.plane_embedding(2, 32)
        self.layers = _spect.conv_stack(len(ATTRIBUTE_DIMS) + 1, [1, 2, 4, 8, 16, 32, 64, None], d)


class Generator(_Family, _spect.SpectGenerator):
    def __init__(self, d=64):
        super().__init__()
        self.has_boat_embedding = nn.Embedding(2, 256)
        self.layers = _spect.deconv_stack(LATENT_DIM + 257, [16, 8, 4, 2, 1, 1, None], d)


class Discriminator(_Family, _spect.SpectDiscriminator):
    def __init__(self, d=64):
        super().__init__()
        self.has_boat_embedding = _spect.plane_embedding(2, 32)
        self.dx = _spect.conv_stack(len(ATTRIBUTE_DIMS) + 1, [1, 2, 4, 8, 16, 32, 64, None], d)
        self.dz = _spect.dz_stack()
        self.dxz = _spect.dxz_stack()


STFT = dict(n_fft=1023, win_length=256, hop_length=79, pad=200)   # EsrfStation.audio_to_spectrogram (reference :36-39)


def train(path_to_wavs,
          path_to_labels: str = None,
          n_epochs: int = 200,
          l_rate: float = 1e-4,
          device: str = 'cpu',
          save_images_every: int = 2,
          batch_size: int = 64,
          image_output_path: str = '',
          validation_split=0.2,
          start_model_path=None,
          checkpoint_every=None,
          checkpoint_path=None):
    """Reference signature (:263-272) and loop (:298-379).  ``start_model_path`` warm-starts from a pickled-module
    checkpoint (:280-284; optimiser state is not restored, as in the reference).  ``path_to_wavs`` may be a data
    source with the adapter's interface (``_spect.WaveformData(..., **STFT)``) instead of the wav directory, which
    needs the reference's ``EsrfStation`` reader (torchaudio, pandas label tables; raises ImportError here)."""
    E, G, D = Encoder().to(device), Generator().to(device), Discriminator().to(device)
    for m in (E, G, D):
        m.apply(init_weights)
    if start_model_path is not None:
        ckpt = torch.load(start_model_path, map_location=device, weights_only=False)
        E, G, D = ckpt["E"].to(device), ckpt["G"].to(device), ckpt["D"].to(device)
    if _spect.is_data_source(path_to_wavs):
        data = path_to_wavs
    else:
        data = EsrfStation(path_to_wavs, path_to_labels, device=device, validation_split=validation_split)
    return _spect.run_training(E, G, D, data, dict(batch_size=batch_size, mode='train'), tuple(ATTRIBUTE_DIMS),
                               n_epochs, l_rate, device, checkpoint_every=checkpoint_every,
                               checkpoint_path=checkpoint_path)
