import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "imagecfgen-pytorch_amd")]
import torch
from ali_hip.spectrogram import SpectrogramFrontEnd
for name, (n_fft, win, hop, pad, L, B) in {"audio": (255, 128, None, 96, 8000, 256), "whale": (511, 128, 24, 64, 5952, 128),
                                           "esrf": (1023, 256, 79, 200, 40000, 64)}.items():
    fe = SpectrogramFrontEnd(n_fft, win, hop, pad)
    x = torch.randn(B, L, device="cuda")
    for _ in range(3): out = fe(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): out = fe(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    xc = x.cpu()
    import time
    t0 = time.time()
    for _ in range(3):
        ref = (torch.stft(torch.nn.functional.pad(xc, (pad, pad)), n_fft=n_fft, hop_length=hop or win // 2, win_length=win,
                          window=torch.hann_window(win), center=True, pad_mode="reflect", return_complex=True).abs().pow(2) + 1e-6).log()
    cpu_ms = (time.time() - t0) / 3 * 1e3
    print(f"{name}: B={B} L={L} -> {tuple(out.shape)}  {ms*1e3:.0f} us/batch ({B/ms*1e3:.0f} clips/s); torch.stft on the host CPU {cpu_ms:.1f} ms")
