"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- numpy restatement of the reference's data item rule.

``PianoRollAudioDataset.__getitem__`` (reference model/dataset.py:35-69): a track is
dict(audio int16 [T], label uint8 [n_steps, 88] with 3 = onset, 2 = frame, 1 = offset, velocity uint8 [n_steps, 88]);
with a sequence length the item is the crop starting at

    step_begin = RandomState(seed).randint(T - sequence_length) // HOP_LENGTH       (:41, one draw per item, in call order)
    begin      = step_begin * HOP_LENGTH

audio[begin : begin + sequence_length] / 32768 as float32 (:62), and the label rows step_begin .. + sequence_length //
HOP_LENGTH decoded into onset = (label == 3), offset = (label == 1), frame = (label > 1) as float32 (:63-65),
velocity / 128 (:66).  Integer / byte work: the product's device-side cropper must match this BIT-EXACTLY.
Pinned by tests/golden/dataset.npz (the reference class itself run on in-memory tracks).
"""
import numpy as np

HOP_LENGTH = 512          # reference model/constants.py


def synthetic_tracks(n=3, seed=123, min_len=40000, max_len=70000):
    """Deterministic in-memory tracks (numpy RandomState is stable across versions): what the golden script feeds
    the reference and what the tests feed the oracle / the HIP kernel."""
    rng = np.random.RandomState(seed)
    tracks = []
    for i in range(n):
        t = int(rng.randint(min_len, max_len))
        steps = (t - 1) // HOP_LENGTH + 1
        tracks.append({'path': f'track{i}.flac',
                       'audio': rng.randint(-32768, 32768, size=t).astype(np.int16),
                       'label': rng.randint(0, 4, size=(steps, 88)).astype(np.uint8),
                       'velocity': rng.randint(0, 128, size=(steps, 88)).astype(np.uint8)})
    return tracks


def draw_begin(random_state, audio_length, sequence_length):
    """model/dataset.py:41,48 -- returns (step_begin, begin)."""
    step_begin = int(random_state.randint(audio_length - sequence_length)) // HOP_LENGTH
    return step_begin, step_begin * HOP_LENGTH


def crop_item(track, step_begin, sequence_length):
    """model/dataset.py:43-66 for a given step_begin."""
    n_steps = sequence_length // HOP_LENGTH
    begin = step_begin * HOP_LENGTH
    audio = track['audio'][begin:begin + sequence_length].astype(np.float32) / np.float32(32768.0)
    label = track['label'][step_begin:step_begin + n_steps, :]
    vel = track['velocity'][step_begin:step_begin + n_steps, :]
    return {'audio': audio, 'start_idx': begin,
            'onset': (label == 3).astype(np.float32), 'offset': (label == 1).astype(np.float32),
            'frame': (label > 1).astype(np.float32), 'velocity': vel.astype(np.float32) / np.float32(128.0)}
