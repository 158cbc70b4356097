"""Oracle: log-Mel front-end.  TEST INFRASTRUCTURE ONLY.

Restates, on CPU / fp32:
  * nnAudio-0.2.0 ``create_fourier_kernels`` and ``mel`` (NOT vendored under
    /root/reference; call sites model/Spectrogram.py:133-141 and :421) --
    PARITY UNPINNED, see oracle/__init__.py;
  * ``STFT.forward`` (model/Spectrogram.py:187-231) and
    ``MelSpectrogram.forward`` (model/Spectrogram.py:443-461);
  * the log + imagewise min-max normalisation of
    ``UNet_Onset.run_on_batch`` (model/UNet_onset.py:419-423, :432-442) and
    ``Normalization('imagewise')`` (model/utils.py:94-100).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE = 16000          # model/constants.py:4
HOP_LENGTH = 512             # model/constants.py:5  (16000*32//1000)
N_FFT = 2048                 # model/constants.py:25 WINDOW_LENGTH, Spectrogram.py:396 n_fft default
N_MELS = 229                 # model/constants.py:13
MEL_FMIN = 30.0              # model/constants.py:14
MEL_FMAX = 8000.0            # model/constants.py:15
N_FREQ = N_FFT // 2 + 1


def hann_periodic(n=N_FFT):
    """scipy.signal.get_window('hann', n, fftbins=True) restated: 0.5-0.5cos(2*pi*k/n)."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def fourier_kernels(n_fft=N_FFT):
    """nnAudio ``create_fourier_kernels(n_fft, freq_scale='no', window='hann')``.

    Returns (wsin, wcos, window) with wsin/wcos already multiplied by the
    window as model/Spectrogram.py:162-164 does; shapes [n_fft//2+1, 1, n_fft]
    and [n_fft]; float32 tables computed in float64.
    """
    n = np.arange(n_fft, dtype=np.float64)
    k = np.arange(n_fft // 2 + 1, dtype=np.float64)
    ang = 2.0 * np.pi * k[:, None] * n[None, :] / n_fft
    win = hann_periodic(n_fft).astype(np.float32)
    ksin = np.sin(ang).astype(np.float32)
    kcos = np.cos(ang).astype(np.float32)
    wsin = torch.from_numpy(ksin) * torch.from_numpy(win)
    wcos = torch.from_numpy(kcos) * torch.from_numpy(win)
    return wsin.unsqueeze(1), wcos.unsqueeze(1), torch.from_numpy(win)


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3.0
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    big = f >= min_log_hz
    mels = np.where(big, min_log_mel + np.log(np.maximum(f, 1e-12) / min_log_hz) / logstep, mels)
    return mels


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3.0
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    big = m >= min_log_mel
    freqs = np.where(big, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)
    return freqs


def mel_filterbank(sr=SAMPLE_RATE, n_fft=N_FFT, n_mels=N_MELS, fmin=MEL_FMIN, fmax=MEL_FMAX):
    """nnAudio/librosa-0.7 ``mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm=1)`` -> [n_mels, n_fft//2+1] fp32."""
    n_freq = n_fft // 2 + 1
    fftfreqs = np.linspace(0.0, sr / 2.0, n_freq)
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    mel_f = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, n_freq), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    w *= enorm[:, None]
    return torch.from_numpy(w.astype(np.float32))


def frontend_buffers():
    """The four registered buffers the reference keeps in its state_dict
    (model/Spectrogram.py:167-178, :435)."""
    wsin, wcos, win = fourier_kernels()
    return {
        'spectrogram.mel_basis': mel_filterbank(),
        'spectrogram.stft.wsin': wsin,
        'spectrogram.stft.wcos': wcos,
        'spectrogram.stft.window_mask': win.view(1, -1, 1),
    }


def melspec_power(audio, bufs):
    """MelSpectrogram.forward (model/Spectrogram.py:443-461) over STFT.forward
    (:187-231, 'Magnitude' output, trainable=False): reflect-pad n_fft/2, two
    conv1d with the windowed sin/cos kernels at stride hop, sqrt(re^2+im^2),
    **2, mel_basis @ spec.   audio [B, L] -> [B, 229, 1 + L//512]."""
    x = audio
    if x.dim() == 1:                       # broadcast_dim (nnAudio.utils)
        x = x[None, None, :]
    elif x.dim() == 2:
        x = x[:, None, :]
    elif x.dim() != 3:
        raise ValueError("Only support input with shape = (batch, len) or shape = (len)")
    x = F.pad(x, (N_FFT // 2, N_FFT // 2), mode='reflect')
    imag = F.conv1d(x, bufs['spectrogram.stft.wsin'], stride=HOP_LENGTH)
    real = F.conv1d(x, bufs['spectrogram.stft.wcos'], stride=HOP_LENGTH)
    mag = torch.sqrt(real.pow(2) + imag.pow(2))
    return torch.matmul(bufs['spectrogram.mel_basis'], mag ** 2.0)


def log_normalise(mel, log=True):
    """model/UNet_onset.py:420-423: log(spec+1e-5), per-clip min-max
    (model/utils.py:94-100), transpose to time-major, add the channel dim.
    mel [B, 229, T] -> [B, 1, T, 229]."""
    s = torch.log(mel + 1e-5) if log else mel
    b = s.shape[0]
    flat = s.reshape(b, -1)
    mx = flat.max(1, keepdim=True)[0].unsqueeze(1)
    mn = flat.min(1, keepdim=True)[0].unsqueeze(1)
    s = (s - mn) / (mx - mn)
    return s.transpose(-1, -2).unsqueeze(1)


def frontend(audio, bufs, log=True):
    """audio [B, L] (the caller has already dropped the last sample,
    model/UNet_onset.py:419,432) -> normalised log-mel [B, 1, T, 229]."""
    return log_normalise(melspec_power(audio, bufs), log)
