"""Signal and label geometry of the hot path.

The names are part of the drop-in surface (callers do ``from model import *`` and use them), the values are the ones the
reference fixes in model/constants.py:4-25: 16 kHz audio, 32 ms hop -> 512 samples, 2048-sample Hann window, 229 Slaney
mel bands between 30 Hz and Nyquist, the 88 piano keys as MIDI 21..108.
"""

SAMPLE_RATE = 16000                     # Hz
WINDOW_LENGTH = 2048                    # samples per STFT frame (also n_fft)

_HOP_MS = 32
HOP_LENGTH = (SAMPLE_RATE // 1000) * _HOP_MS          # 512 samples between frames
ONSET_LENGTH = OFFSET_LENGTH = HOP_LENGTH             # an onset / offset label spans one hop ...
HOPS_IN_ONSET = HOPS_IN_OFFSET = 1                    # ... i.e. exactly one frame

N_BINS = 229                            # mel bands
MEL_FMIN, MEL_FMAX = 30, SAMPLE_RATE // 2

MIN_MIDI, MAX_MIDI = 21, 108            # A0 .. C8

assert HOP_LENGTH == 512 and MAX_MIDI - MIN_MIDI + 1 == 88
