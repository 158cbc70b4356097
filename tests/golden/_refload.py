"""Import the reference implementation (/root/reference) in THIS container.

Used only by ``make_golden.py`` (never on the GPU box, never by the product).
The reference needs packages that are not installed here (nnAudio, sacred,
mir_eval, soundfile, mido, tensorboard); they are replaced by inert stubs in
``sys.modules`` -- none of them is on the arithmetic path except nnAudio, whose
``Spectrogram`` module IS the reference's own vendored model/Spectrogram.py
(loaded by path) and whose three un-vendored helpers come from the oracle's
restatement (oracle/frontend.py; "parity unpinned" there).
"""
import importlib.util
import sys
import types

REF = '/root/reference'


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference():
    sys.dont_write_bytecode = True          # /root/reference is read-only
    if 'model' in sys.modules and getattr(sys.modules['model'], '__file__', '').startswith(REF):
        return sys.modules['model']
    sys.path.insert(0, '/root/repo')
    from oracle import frontend as fe
    import numpy as np

    _stub('soundfile')
    _stub('mido', Message=object, MidiFile=object, MidiTrack=object)
    me = _stub('mir_eval')
    for sub in ('multipitch', 'transcription', 'transcription_velocity', 'util'):
        setattr(me, sub, _stub('mir_eval.' + sub, evaluate=None, precision_recall_f1_overlap=None,
                               midi_to_hz=None, hz_to_midi=None))
    import torch.utils  # noqa
    _stub('torch.utils.tensorboard', SummaryWriter=object)
    import matplotlib
    matplotlib.use = lambda *a, **k: None

    def create_fourier_kernels(n_fft, win_length=None, freq_bins=None, fmin=50, fmax=6000, sr=44100,
                               freq_scale='linear', window='hann', verbose=True):
        assert freq_scale == 'no' and window == 'hann' and (win_length in (None, n_fft))
        n = np.arange(n_fft, dtype=np.float64)
        k = np.arange(n_fft // 2 + 1, dtype=np.float64)
        ang = 2.0 * np.pi * k[:, None] * n[None, :] / n_fft
        ksin = np.sin(ang).astype(np.float32)[:, None, :]
        kcos = np.cos(ang).astype(np.float32)[:, None, :]
        bins2freq = [i * sr / n_fft for i in range(n_fft // 2 + 1)]
        binslist = list(range(n_fft // 2 + 1))
        return ksin, kcos, bins2freq, binslist, fe.hann_periodic(n_fft).astype(np.float32)

    def broadcast_dim(x):
        if x.dim() == 2:
            return x[:, None, :]
        if x.dim() == 1:
            return x[None, None, :]
        if x.dim() == 3:
            return x
        raise ValueError("Only support input with shape = (batch, len) or shape = (len)")

    def mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False, norm=1, dtype=np.float32):
        assert not htk and norm == 1
        return fe.mel_filterbank(sr, n_fft, n_mels, fmin, fmax if fmax is not None else sr / 2).numpy()

    nn_pkg = _stub('nnAudio')
    nn_pkg.__path__ = []
    _stub('nnAudio.utils', create_fourier_kernels=create_fourier_kernels, broadcast_dim=broadcast_dim)
    sys.modules['nnAudio.utils'].__all__ = ['create_fourier_kernels', 'broadcast_dim']
    _stub('nnAudio.librosa_functions', mel=mel)
    sys.modules['nnAudio.librosa_functions'].__all__ = ['mel']
    spec = importlib.util.spec_from_file_location('nnAudio.Spectrogram', REF + '/model/Spectrogram.py')
    mod = importlib.util.module_from_spec(spec)
    sys.modules['nnAudio.Spectrogram'] = mod
    spec.loader.exec_module(mod)
    nn_pkg.Spectrogram = mod

    sys.path.insert(0, REF)
    import model
    return model
