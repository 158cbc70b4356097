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


def ingest_corpus(root):
    """A tiny synthetic corpus in `root` laid out like MAPS and MusicNet (16 kHz mono 16-bit wav + tsv note lists + metadata):
    shared by tests/golden/make_golden.py (fed to the reference classes) and tests/test_dataset_ingest.py (fed to the product)."""
    import os
    import pickle
    from scipy.io import wavfile
    rng = np.random.RandomState(5)

    def track(path_wav, path_tsv, seconds):
        n = int(seconds * 16000) + int(rng.randint(0, 700))
        os.makedirs(os.path.dirname(path_wav), exist_ok=True)
        os.makedirs(os.path.dirname(path_tsv), exist_ok=True)
        wavfile.write(path_wav, 16000, rng.randint(-20000, 20000, size=n).astype(np.int16))
        rows = []
        for _ in range(int(rng.randint(5, 40))):
            on = float(rng.uniform(0, seconds))
            # half-frame onsets (round-half-even), notes running past the end of the audio, overlapping notes on one key
            if rng.rand() < 0.3:
                on = (int(on * 31.25) + 0.5) / 31.25
            off = on + float(rng.uniform(0.01, 1.5))
            rows.append((on, off, int(rng.randint(21, 109)), int(rng.randint(1, 128))))
        np.savetxt(path_tsv, np.array(rows), fmt='%.6f', delimiter='\t', header='onset,offset,note,velocity')

    maps = os.path.join(root, 'MAPS')
    pieces = ['alb_se2', 'bk_xmas1', 'chpn_op25', 'deb_clai', 'grieg_butterfly', 'liz_et6']
    for g in ('AkPnBcht', 'ENSTDkAm'):
        for pc in pieces:
            track(os.path.join(maps, 'flac', f'MAPS_MUS-{pc}_{g}.wav'), os.path.join(maps, 'tsvs', f'MAPS_MUS-{pc}_{g}.tsv'), 1.2)
    with open(os.path.join(root, 'overlapping.pkl'), 'wb') as fh:
        pickle.dump(['chpn_op25', 'liz_et6'], fh)
    mn = os.path.join(root, 'MusicNet')
    ensembles = ['Solo Violin', 'Solo Violin', 'Violin and Harpsichord', 'Accompanied Violin', 'Accompanied Violin',
                 'String Quartet', 'String Quartet', 'String Quartet', 'String Sextet', 'Viola Quintet', 'Solo Cello', 'Solo Cello',
                 'Accompanied Cello', 'Accompanied Clarinet', 'Clarinet Quintet', 'Pairs Clarinet-Horn-Bassoon',
                 'Clarinet-Cello-Piano Trio', 'Wind Octet', 'Wind Octet', 'Wind Quintet', 'Wind Quintet', 'Solo Piano', 'Solo Flute']
    ids = [2200 + 7 * i for i in range(len(ensembles))] + [2203, 2204]
    ensembles = ensembles + ['Solo Flute', 'Solo Flute']
    os.makedirs(mn, exist_ok=True)
    with open(os.path.join(mn, 'train_metadata.csv'), 'w') as fh:
        fh.write('id,composer,composition,movement,ensemble\n')
        for i, e in zip(ids, ensembles):
            fh.write(f'{i},X,Y,Z,{e}\n')
    for i in ids:
        track(os.path.join(mn, 'train_data', f'{i}.wav'), os.path.join(mn, 'tsv_train_labels', f'{i}.tsv'), 0.7)
    for i in (2106, 2191, 2298, 2628, 1819, 2416, 2303, 2382):
        track(os.path.join(mn, 'test_data', f'{i}.wav'), os.path.join(mn, 'tsv_test_labels', f'{i}.tsv'), 0.7)
    return root

