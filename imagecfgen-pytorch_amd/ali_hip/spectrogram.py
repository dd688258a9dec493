"""Spectrogram front-end on the device: the transform the reference's dataset adapters apply to every
batch of raw audio before the hot path,

    (torchaudio.transforms.Spectrogram(n_fft, win_length, hop_length, pad)(x) + 1e-6).log()      [B, n_fft//2+1, frames]

(``audio_mnist.py:59-61,116``: n_fft=255, win 128, pad 96; ``whalecalls.py:52-55``: 511 / 128 / hop 24 / pad 64;
``esrf_acoustic.py:36-39``: 1023 / 256 / hop 79 / pad 200) and, fused, ``spect_to_img`` (``audio_mnist.py:361-363``).

The Hann window has ``win_length`` non-zero samples inside the ``n_fft`` frame, so a frame's one-sided DFT is a
[2F x win_length] cos|sin matrix applied to those samples: the frames (a strided, overlapping view of the padded
signal) go through the ordinary fp32-MFMA GEMM (``ali_conv_fwd``, 1x1) and ``ali_spect_post`` does power, log,
standardise, clip and the [B,T,F] -> [B,F,T] transposition.  An exact k-ordered fp32 DFT, no FFT butterflies: the
matrices are small (128..256 x 256..1024) and the GEMM is far from being the bottleneck of an epoch.

torchaudio is not importable in the build container: parity is pinned to ``torch.stft`` with torchaudio's documented
parameter mapping (tests/test_gpu_kernels.py), not to torchaudio itself.
"""
import math

import torch
import torch.nn.functional as F

from . import ops


class SpectrogramFrontEnd:
    def __init__(self, n_fft, win_length=None, hop_length=None, pad=0, device="cuda"):
        self.n_fft = n_fft
        self.win = win_length or n_fft
        self.hop = hop_length or self.win // 2
        self.pad = pad
        self.F = n_fft // 2 + 1
        if self.win % 32:
            raise ValueError("win_length must be a multiple of 32 (channel stride of the GEMM's fast path)")
        self.left = (n_fft - self.win) // 2                       # torch.stft centres the window in the frame
        n = torch.arange(self.win, dtype=torch.float64) + self.left
        f = torch.arange(self.F, dtype=torch.float64)
        ang = 2.0 * math.pi * f[:, None] * n[None, :] / n_fft
        w = torch.hann_window(self.win, periodic=True, dtype=torch.float64)
        mat = torch.cat([torch.cos(ang) * w, -torch.sin(ang) * w], dim=0)          # [2F, win]
        self.weight = mat.float().reshape(2 * self.F, 1, self.win).contiguous().to(device)

    def frames(self, wave):
        """[B, L] -> contiguous [B, T, win]: the window's support of every STFT frame (centre=True, reflect)."""
        x = F.pad(wave.float(), (self.pad, self.pad))
        x = F.pad(x[:, None, :], (self.n_fft // 2, self.n_fft // 2), mode="reflect")[:, 0]
        T = 1 + (x.shape[1] - self.n_fft) // self.hop
        return x[:, self.left:].unfold(1, self.win, self.hop)[:, :T].contiguous()

    @torch.no_grad()
    def __call__(self, wave, mean=None, std=None, stds_kept=3.0):
        """wave [B, L] (CUDA).  Returns the log-spectrogram [B, F, T]; with ``mean`` / ``std`` (shape [T], the
        reference's per-last-index statistics) the standardised, clipped image ``spect_to_img`` would give."""
        fr = self.frames(wave)
        B, T, _ = fr.shape
        y = torch.empty(B, T, 1, 2 * self.F, dtype=torch.float32, device=fr.device)
        ops.conv_fwd(ops.geom(B, T, 1, self.win, T, 1, 2 * self.F, 1, 1, 1, 0), fr.reshape(B, T, 1, self.win),
                     self.weight, y, ops.epilogue())
        out = torch.empty(B, self.F, T, dtype=torch.float32, device=fr.device)
        if mean is not None:
            mean, std = mean.reshape(-1).float().contiguous(), std.reshape(-1).float().contiguous()
        return ops.spect_post(y, B, T, self.F, out, mean, std, stds_kept)
