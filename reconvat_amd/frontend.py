"""Log-Mel front-end modules with the reference's attribute / state_dict layout.

``MelSpectrogram`` mirrors nnAudio's class as the reference uses it (model/UNet_onset.py:354-356,
model/Spectrogram.py:396-461): buffers ``mel_basis [229,1025]`` and ``stft.{wsin,wcos} [1025,1,2048]``,
``stft.window_mask [1,2048,1]`` are kept so a reference checkpoint loads with ``load_state_dict`` -- but
the computation is the fused FFT kernel (csrc/mel.hip), driven by small derived tables (window, FFT
twiddles, the sparse rows of ``mel_basis``) that are rebuilt whenever the buffers are (re)loaded.

The three helpers nnAudio 0.2.0 provides (not vendored in the reference) are restated here from their
published definitions: periodic Hann window, windowed DFT kernels, librosa-0.7 Slaney mel filterbank.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .constants import SAMPLE_RATE, HOP_LENGTH, N_BINS, MEL_FMIN, MEL_FMAX, WINDOW_LENGTH


def _hann(n):
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def _slaney_hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3.0)
    log = 15.0 + np.log(np.maximum(f, 1e-12) / 1000.0) / (np.log(6.4) / 27.0)
    return np.where(f >= 1000.0, log, lin)


def _slaney_mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    lin = m * (200.0 / 3.0)
    log = 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0))
    return np.where(m >= 15.0, log, lin)


def slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax):
    nf = n_fft // 2 + 1
    freqs = np.linspace(0.0, sr / 2.0, nf)
    edges = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(fmin), _slaney_hz_to_mel(fmax), n_mels + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - freqs[None, :]
    lower = -ramps[:-2] / width[:-1, None]
    upper = ramps[2:] / width[1:, None]
    basis = np.maximum(0.0, np.minimum(lower, upper))
    basis *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return basis.astype(np.float32)


class STFT(nn.Module):
    """Holds the reference's STFT buffers (state_dict compatibility only)."""

    def __init__(self, n_fft=WINDOW_LENGTH):
        super().__init__()
        n = np.arange(n_fft, dtype=np.float64)
        k = np.arange(n_fft // 2 + 1, dtype=np.float64)
        ang = 2.0 * np.pi * k[:, None] * n[None, :] / n_fft
        win = torch.from_numpy(_hann(n_fft).astype(np.float32))
        self.register_buffer('wsin', (torch.from_numpy(np.sin(ang).astype(np.float32)) * win).unsqueeze(1))
        self.register_buffer('wcos', (torch.from_numpy(np.cos(ang).astype(np.float32)) * win).unsqueeze(1))
        self.register_buffer('window_mask', win.view(1, -1, 1))


class MelSpectrogram(nn.Module):
    def __init__(self, sr=SAMPLE_RATE, n_fft=WINDOW_LENGTH, n_mels=N_BINS, hop_length=HOP_LENGTH, fmin=MEL_FMIN,
                 fmax=MEL_FMAX):
        super().__init__()
        if n_fft != 2048 or hop_length != 512 or n_mels > 256:
            raise ValueError('the fused front-end kernel (csrc/mel.hip: four frames per workgroup sharing one audio chunk, one thread '
                             f'per mel band) is built for n_fft = 2048, hop_length = 512, n_mels <= 256; got n_fft={n_fft}, '
                             f'hop_length={hop_length}, n_mels={n_mels} (the reference scripts use 2048 / 512 / 229)')
        self.n_fft, self.hop, self.n_mels = n_fft, hop_length, n_mels
        self.stft = STFT(n_fft)
        self.register_buffer('mel_basis', torch.from_numpy(slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax)))
        self._tables = None
        self._tables_key = None

    def tables(self):
        """Derived kernel tables on the buffers' device (rebuilt if the buffers changed)."""
        mb, wm = self.mel_basis, self.stft.window_mask
        key = (mb.device, mb._version, wm._version, mb.data_ptr())
        if self._tables is not None and self._tables_key == key:
            return self._tables
        basis = mb.detach().cpu().numpy()
        nz = basis != 0
        start = nz.argmax(1).astype(np.int32)
        last = (basis.shape[1] - 1 - nz[:, ::-1].argmax(1)).astype(np.int32)
        length = np.where(nz.any(1), last - start + 1, 0).astype(np.int32)
        if int(length.max()) > 32:
            raise ValueError(f'the fused front-end kernel keeps 32 filter taps per mel band in registers; the widest band of this '
                             f'filterbank spans {int(length.max())} FFT bins (fewer mel bands / a higher fmax than the reference '
                             "scripts' 229 bands over 30 .. 8000 Hz widen the bands)")
        # taps past a band's end are stored as zeros and still multiplied: like the reference's dense `mel_basis @ spec`
        # (0 * inf = nan there poisons EVERY band), a non-finite power bin is not contained to its own bands -- finite audio assumed
        ld = 32
        w = np.zeros((basis.shape[0], ld), dtype=np.float32)
        for i in range(basis.shape[0]):
            w[i, :length[i]] = basis[i, start[i]:start[i] + length[i]]
        k = np.arange(self.n_fft // 2, dtype=np.float64)
        tw = np.stack([np.cos(2 * np.pi * k / self.n_fft), -np.sin(2 * np.pi * k / self.n_fft)], 1).astype(np.float32)
        dev = mb.device
        self._tables = {
            'window': wm.detach().reshape(-1).contiguous().float(),
            'twiddle': torch.from_numpy(tw).to(dev),
            'mel_start': torch.from_numpy(start).to(dev),
            'mel_len': torch.from_numpy(length).to(dev),
            'mel_w': torch.from_numpy(w).to(dev),
        }
        self._tables_key = key
        return self._tables

    @staticmethod
    def _as_batch(x):
        if x.dim() == 1:
            return x[None, :]
        if x.dim() == 3 and x.shape[1] == 1:
            return x[:, 0, :]
        if x.dim() == 2:
            return x
        raise ValueError("Only support input with shape = (batch, len) or shape = (len)")

    def forward(self, x):
        """Mel power spectrogram [B, n_mels, T] like nnAudio's MelSpectrogram.forward."""
        x = self._as_batch(x)
        if x.shape[-1] < self.n_fft // 2:
            raise AssertionError("Signal length shorter than reflect padding length (n_fft // 2).")
        return ops.melspec(x, self.tables(), do_log=False, normalise=False, hop=self.hop).transpose(1, 2)

    def lognorm(self, x, log=True, normalise=True):
        """Fused path: (log-)mel, per-clip min-max normalised, time-major [B, 1, T, n_mels]."""
        x = self._as_batch(x)
        return ops.melspec(x, self.tables(), do_log=log, normalise=normalise, hop=self.hop).unsqueeze(1)


class Normalization:
    """model/utils.py:82-106 ('imagewise' and 'framewise' min-max) for callers that use
    ``model.normalize.transform`` directly; the hot path uses the fused kernel instead."""

    def __init__(self, mode='framewise'):
        if mode not in ('framewise', 'imagewise'):
            print('please choose the correct mode')
        self.mode = mode

    def transform(self, x):
        if self.mode == 'framewise':
            x_max = x.max(1, keepdim=True)[0]
            x_min = x.min(1, keepdim=True)[0]
            out = (x - x_min) / (x_max - x_min)
            out[torch.isnan(out)] = 0
            return out
        flat = x.reshape(x.shape[0], -1)
        x_max = flat.max(1, keepdim=True)[0].unsqueeze(1)
        x_min = flat.min(1, keepdim=True)[0].unsqueeze(1)
        return (x - x_min) / (x_max - x_min)
