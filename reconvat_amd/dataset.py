"""Host-side corpora for the training scripts: which files make up a corpus, how one track becomes
(int16 audio, uint8 label roll, uint8 velocity roll), and the host fallback for drawing one training item.

Layering (the device feed `reconvat_amd/feed.py::DeviceCorpus` consumes the same `tracks`):

  file rules  : `MAPS`, `MAESTRO`, `MusicNet`, `Guqin` name the (audio, tsv) pairs of a group -- tables below restate the
                reference's selection rules (model/dataset.py:145-420); `CachedFolder` takes any folder of `.pt` caches.
  ingest      : `ingest_track` = the `.pt` cache contract of model/dataset.py:85-142: cache hit -> torch.load, else decode
                the audio (wav natively, flac through `soundfile` when that package exists), paint the tsv notes into the
                roll (`paint_roll`, 3 = onset, 2 = sustain, 1 = offset) and write the cache.
  item rule   : `crop_item` (model/dataset.py:35-69): `step_begin = RandomState(seed).randint(T - L) // 512`, audio / 2^15,
                masks `label == 3`, `== 1`, `> 1`, velocity / 128 -- bit-exact against the oracle / reference golden
                (tests/test_feed.py); the GPU feed implements the same rule in `rv_crop_segments`.

`SyntheticSegments` produces seeded random segments of the same shapes -- what the benchmark and the plumbing runs use,
since none of the corpora can be downloaded here.
"""
import os
import pickle
from glob import glob

import numpy as np
import torch
from torch.utils.data import Dataset

from .constants import HOP_LENGTH, SAMPLE_RATE, MIN_MIDI, MAX_MIDI, HOPS_IN_ONSET, HOPS_IN_OFFSET

N_KEYS = MAX_MIDI - MIN_MIDI + 1
_MASKS = (('onset', lambda roll: roll == 3), ('offset', lambda roll: roll == 1), ('frame', lambda roll: roll > 1))


# ------------------------------------------------------------------------------------------------------------------
# ingest: audio + tsv -> cached track
# ------------------------------------------------------------------------------------------------------------------
def read_audio_int16(path):
    """16-bit PCM samples of a mono 16 kHz file as int16 [T]."""
    if path.endswith('.wav'):
        from scipy.io import wavfile
        sr, pcm = wavfile.read(path)
        if pcm.dtype != np.int16:
            raise ValueError(f'{path}: expected 16-bit PCM, got {pcm.dtype}')
    else:
        try:
            import soundfile
        except ImportError as e:
            raise FileNotFoundError(
                f'{path}: no `.pt` cache next to it and decoding FLAC needs the `soundfile` package, which is not installed. '
                'Convert the corpus to 16 kHz wav (read natively) or create the caches with Preprocessing.ipynb / the '
                'reference (model/dataset.py:85-142).') from e
        pcm, sr = soundfile.read(path, dtype='int16')
    if sr != SAMPLE_RATE:
        raise ValueError(f'{path}: sample rate {sr}, expected {SAMPLE_RATE}')
    if pcm.ndim != 1:
        raise ValueError(f'{path}: expected mono audio, got shape {pcm.shape}')
    return np.ascontiguousarray(pcm)


def paint_roll(notes, n_steps):
    """tsv rows (onset s, offset s, MIDI note, velocity) -> (label, velocity) uint8 [n_steps, 88]; later rows overwrite
    earlier ones exactly like the reference's sequential slice assignments (model/dataset.py:123-138)."""
    label = np.zeros((n_steps, N_KEYS), dtype=np.uint8)
    velocity = np.zeros((n_steps, N_KEYS), dtype=np.uint8)
    notes = np.atleast_2d(np.asarray(notes, dtype=np.float64))
    if notes.size == 0:
        return label, velocity
    to_step = SAMPLE_RATE / HOP_LENGTH
    # python round() is round-half-to-even, as is np.rint
    head = np.rint(notes[:, 0] * to_step).astype(np.int64)
    tail = np.minimum(n_steps, np.rint(notes[:, 1] * to_step).astype(np.int64))
    attack_end = np.minimum(n_steps, head + HOPS_IN_ONSET)
    release_end = np.minimum(n_steps, tail + HOPS_IN_OFFSET)
    keys = notes[:, 2].astype(np.int64) - MIN_MIDI
    for h, a, t, r, k, v in zip(head, attack_end, tail, release_end, keys, notes[:, 3]):
        column, vcolumn = label[:, k], velocity[:, k]
        for lo, hi, code in ((h, a, 3), (a, t, 2), (t, r, 1)):
            column[lo:hi] = code
        vcolumn[h:t] = v
    return label, velocity


def ingest_track(audio_path, tsv_path, refresh=False):
    cache = os.path.splitext(audio_path)[0] + '.pt' if audio_path.endswith(('.flac', '.wav')) else audio_path
    if os.path.exists(cache) and not refresh:
        return torch.load(cache)
    pcm = read_audio_int16(audio_path)
    n_steps = (len(pcm) - 1) // HOP_LENGTH + 1
    label, velocity = paint_roll(np.loadtxt(tsv_path, delimiter='\t', skiprows=1), n_steps)
    track = dict(path=audio_path, audio=torch.from_numpy(pcm), label=torch.from_numpy(label), velocity=torch.from_numpy(velocity))
    torch.save(track, cache)
    return track


# ------------------------------------------------------------------------------------------------------------------
# item rule
# ------------------------------------------------------------------------------------------------------------------
def crop_item(track, step_begin, sequence_length, device='cpu'):
    """One item of a track: the window [step_begin, step_begin + L/512) of the rolls and the matching L samples
    (`sequence_length=None`: the whole track, as the evaluation sets use it)."""
    audio, roll, vel = track['audio'], track['label'], track['velocity']
    item = {'path': track['path']}
    if sequence_length is not None:
        first = step_begin * HOP_LENGTH
        steps = slice(step_begin, step_begin + sequence_length // HOP_LENGTH)
        audio, roll, vel = audio[first:first + sequence_length], roll[steps], vel[steps]
        item['start_idx'] = first
    roll = roll.to(device)
    item['audio'] = audio.to(device).to(torch.float32) * (1.0 / 32768.0)          # exact: power-of-two scale
    item['label'] = roll
    for name, rule in _MASKS:
        item[name] = rule(roll).to(torch.float32)
    item['velocity'] = vel.to(device).to(torch.float32) * (1.0 / 128.0)
    return item


class PianoRollAudioDataset(Dataset):
    """In-memory corpus with the reference's surface (`.data`, `.sequence_length`, `.random`, `len`, `[i]`)."""

    def __init__(self, path, groups=None, sequence_length=None, seed=42, refresh=False, device='cpu'):
        self.path, self.sequence_length, self.device, self.refresh = path, sequence_length, device, refresh
        self.groups = list(groups) if groups is not None else self.available_groups()
        self.random = np.random.RandomState(seed)
        self.data = [self.load(*pair) for group in self.groups for pair in self.files(group)]

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        track = self.data[index]
        step_begin = None
        if self.sequence_length is not None:
            step_begin = int(self.random.randint(len(track['audio']) - self.sequence_length)) // HOP_LENGTH
        return crop_item(track, step_begin, self.sequence_length, self.device)

    @classmethod
    def available_groups(cls):
        raise NotImplementedError

    def files(self, group):
        raise NotImplementedError

    def load(self, audio_path, tsv_path):
        return ingest_track(audio_path, tsv_path, self.refresh)


def _audio_files(pattern):
    """Audio files matching `pattern` (written for .flac); a track that only exists as .wav or as a .pt cache counts."""
    stem = pattern[:-len('.flac')]
    found = {os.path.splitext(f)[0]: f for ext in ('.pt', '.wav', '.flac') for f in glob(stem + ext)}
    return sorted(found[s] if not found[s].endswith('.pt') else s + '.flac' for s in found)


class MAPS(PianoRollAudioDataset):
    """model/dataset.py:182-214: group = piano / recording-condition suffix; `overlap=False` drops every piece that also
    occurs in the test pianos (names listed in ./overlapping.pkl); `supersmall` keeps the 4th remaining file only."""
    GROUPS = ['AkPnBcht', 'AkPnBsdf', 'AkPnCGdD', 'AkPnStgb', 'ENSTDkAm', 'ENSTDkCl', 'SptkBGAm', 'SptkBGCl', 'StbgTGd2']

    def __init__(self, path='./MAPS', groups=None, sequence_length=None, overlap=True, seed=42, refresh=False, device='cpu',
                 supersmall=False, overlap_list='overlapping.pkl'):
        self.overlap, self.supersmall, self.overlap_list = overlap, supersmall, overlap_list
        super().__init__(path, groups if groups is not None else ['ENSTDkAm', 'ENSTDkCl'], sequence_length, seed, refresh, device)

    @classmethod
    def available_groups(cls):
        return list(cls.GROUPS)

    def files(self, group):
        audio = _audio_files(os.path.join(self.path, 'flac', f'*_{group}.flac'))
        if not self.overlap:
            with open(self.overlap_list, 'rb') as fh:
                banned = pickle.load(fh)
            audio = sorted(a for a in audio if not any(name in a for name in banned))
            if self.supersmall:
                audio = audio[3:4]
        return sorted((a, a.replace('/flac/', '/tsvs/').rsplit('.', 1)[0] + '.tsv') for a in audio)


class MAESTRO(PianoRollAudioDataset):
    """model/dataset.py:145-180: split groups come from maestro-v2.0.0.json, any other group is a (year) folder.  A note
    list that only exists as .midi is converted to tsv once (reconvat_amd.midi.parse_midi)."""
    SPLITS = ['train', 'validation', 'test']

    def __init__(self, path='../../public_data/MAESTRO/', groups=None, sequence_length=None, seed=42, refresh=False, device='cpu'):
        super().__init__(path, groups if groups is not None else ['train'], sequence_length, seed, refresh, device)

    @classmethod
    def available_groups(cls):
        return list(cls.SPLITS)

    def files(self, group):
        import json
        if group in self.SPLITS:
            with open(os.path.join(self.path, 'maestro-v2.0.0.json')) as fh:
                rows = [r for r in json.load(fh) if r['split'] == group]
            pairs = sorted((os.path.join(self.path, r['audio_filename'].replace('.wav', '.flac')),
                            os.path.join(self.path, r['midi_filename'])) for r in rows)
            pairs = [(a if os.path.exists(a) or os.path.exists(a[:-5] + '.pt') else a[:-5] + '.wav', m) for a, m in pairs]
        else:
            audio = _audio_files(os.path.join(self.path, group, '*.flac'))
            pairs = list(zip(audio, sorted(glob(os.path.join(self.path, group, '*.midi')))))
            if not pairs:
                raise RuntimeError(f'Group {group} is empty')
        out = []
        for audio, midi in pairs:
            tsv = midi.replace('.midi', '.tsv').replace('.mid', '.tsv')
            if not os.path.exists(tsv) and os.path.exists(midi):
                from .midi import parse_midi
                np.savetxt(tsv, parse_midi(midi), fmt='%.6f', delimiter='\t', header='onset,offset,note,velocity')
            out.append((audio, tsv))
        return out


# MusicNet group rules (model/dataset.py:238-342) as data.  A rule is (mode, selector):
#   ('ids', [...])                         fixed recording ids
#   ('ensemble', names, slice)             per ensemble name (substring match on the metadata's `ensemble` column, in the
#                                          metadata's row order): the ids[slice] of that ensemble, concatenated
_STRING = ['Solo Violin', 'Violin and Harpsichord', 'Accompanied Violin', 'String Quartet', 'String Sextet', 'Viola Quintet',
           'Solo Cello', 'Accompanied Cello']
_WIND = ['Accompanied Clarinet', 'Clarinet Quintet', 'Pairs Clarinet-Horn-Bassoon', 'Clarinet-Cello-Piano Trio', 'Wind Octet',
         'Wind Quintet']
MUSICNET_RULES = {
    'train_string_l': ('train', ('ensemble', _STRING, slice(0, 1))),
    'train_string_ul': ('train', ('ensemble', _STRING, slice(1, None))),
    'train_violin_l': ('train', ('ensemble', ['Solo Violin', 'Accompanied Violin'], slice(None))),
    'train_violin_ul': ('train', ('ensemble', ['String Quartet', 'String Sextet'], slice(None))),
    'test_violin': ('test', ('ids', ['2106', '2191', '2298', '2628'])),
    'train_wind_l': ('train', ('ensemble', _WIND, slice(0, 1))),
    'train_wind_ul': ('train', ('ensemble', _WIND, slice(1, None))),
    'test_wind': ('test', ('ids', ['1819', '2416'])),
    'train_flute_l': ('train', ('ids', ['2203'])),
    'train_flute_ul': ('train', ('ensemble+ids', _WIND, slice(None), ['2203'])),
    'test_flute': ('train', ('ids', ['2204'])),
}


class MusicNet(PianoRollAudioDataset):
    def __init__(self, path='./MusicNet', groups=None, sequence_length=None, seed=42, refresh=False, device='cpu'):
        super().__init__(path, groups if groups is not None else ['train'], sequence_length, seed, refresh, device)

    @classmethod
    def available_groups(cls):
        return ['train', 'test']

    def _ensemble_ids(self, names, which):
        import pandas as pd
        meta = pd.read_csv(os.path.join(self.path, 'train_metadata.csv'))
        ids = []
        for name in names:
            ids.extend(meta[meta['ensemble'].str.contains(name)]['id'].values[which].tolist())
        return ids

    def _pairs(self, ids, split):
        audio, tsvs = [], []
        for i in ids:
            audio.extend(_audio_files(os.path.join(self.path, f'{split}_data', f'{i}.flac')))
            tsvs.extend(glob(os.path.join(self.path, f'tsv_{split}_labels', f'{i}.tsv')))
        return list(zip(sorted(audio), sorted(tsvs)))

    def files(self, group):
        if group == 'small test':
            audio = sorted(a for i in ('2303', '2382', '1819') for a in _audio_files(os.path.join(self.path, 'test_data', i + '.flac')))
            return list(zip(audio, sorted(glob(os.path.join(self.path, 'tsv_test_labels', '*.tsv')))))
        if group in MUSICNET_RULES:
            split, rule = MUSICNET_RULES[group]
            ids = list(rule[1]) if rule[0] == 'ids' else self._ensemble_ids(rule[1], rule[2])
            if rule[0] == 'ensemble+ids':
                ids += list(rule[3])
            return self._pairs(ids, split)
        # any other group name is an ensemble substring over the training split
        audio = sorted(a for i in self._ensemble_ids([group], slice(None))
                       for a in _audio_files(os.path.join(self.path, 'train_data', f'{i}.flac')))
        return list(zip(audio, sorted(glob(os.path.join(self.path, 'tsv_train_labels', '*.tsv')))))


class Guqin(PianoRollAudioDataset):
    """model/dataset.py:345-404."""
    PIECES = {'train_l': ['jiou', 'siang', 'ciou', 'yi', 'yu', 'feng', 'yang'], 'train_ul': [], 'test': ['gu', 'guan', 'liang']}

    def __init__(self, path='./Guqin', groups=None, sequence_length=None, seed=42, refresh=False, device='cpu'):
        super().__init__(path, groups if groups is not None else ['train'], sequence_length, seed, refresh, device)

    @classmethod
    def available_groups(cls):
        return list(cls.PIECES)

    def files(self, group):
        if group not in self.PIECES:
            raise ValueError('Please choose a valid group')
        audio = sorted(a for p in self.PIECES[group] for a in _audio_files(os.path.join(self.path, 'audio', p + '.flac')))
        tsvs = sorted(t for p in self.PIECES[group] for t in glob(os.path.join(self.path, 'tsv_label', p + '.tsv')))
        return list(zip(audio, tsvs))


class CachedFolder(PianoRollAudioDataset):
    """Any folder of ``*.pt`` track caches: group = sub-directory."""

    def __init__(self, path, groups=('.',), sequence_length=None, seed=42, device='cpu'):
        super().__init__(path, list(groups), sequence_length, seed, False, device)

    def files(self, group):
        return [(p, None) for p in sorted(glob(os.path.join(self.path, group, '**', '*.pt'), recursive=True))]


class SyntheticSegments(Dataset):
    """Seeded random segments: audio ~ U(-0.1, 0.1), ~5 % frame / ~1 % onset labels."""

    def __init__(self, n_items=64, sequence_length=327680, seed=0, device='cpu'):
        self.n, self.sequence_length, self.seed, self.device = n_items, sequence_length, seed, device
        self._tracks = None

    def __len__(self):
        return self.n

    @property
    def data(self):
        """The same kind of material as whole TRACKS in the PianoRollAudioDataset layout (int16 audio, uint8 label roll with
        3 = onset, 2 = frame; each 16 hops longer than a segment), so that the training scripts can keep a synthetic corpus in
        HBM and crop it with the device feed (reconvat_amd/feed.py) exactly like a real one."""
        if self._tracks is None:
            tracks = []
            steps = self.sequence_length // HOP_LENGTH + 16
            for index in range(self.n):
                g = torch.Generator().manual_seed(self.seed * 1000003 + index)
                audio = ((torch.rand(steps * HOP_LENGTH, generator=g) * 0.2 - 0.1) * 32768.0).round().to(torch.int16)
                u = torch.rand(steps, N_KEYS, generator=g)
                label = (2 * (u > 0.95).to(torch.uint8) + (u > 0.99).to(torch.uint8))
                tracks.append({'path': f'synthetic/{index}', 'audio': audio, 'label': label,
                               'velocity': torch.zeros_like(label)})
            self._tracks = tracks
        return self._tracks

    def __getitem__(self, index):
        if not 0 <= index < self.n:
            raise IndexError(index)                        # `for item in dataset` stops here, like a list-backed dataset
        g = torch.Generator().manual_seed(self.seed * 1000003 + index)
        steps = self.sequence_length // HOP_LENGTH
        audio = torch.rand(self.sequence_length, generator=g) * 0.2 - 0.1
        u = torch.rand(steps, N_KEYS, generator=g)
        frame = (u > 0.95).float()
        onset = (u > 0.99).float()
        return {'path': f'synthetic/{index}', 'audio': audio.to(self.device), 'onset': onset.to(self.device),
                'offset': torch.zeros_like(frame).to(self.device), 'frame': frame.to(self.device),
                'velocity': torch.zeros_like(frame).to(self.device), 'label': (2 * frame + onset).to(torch.uint8)}


def prepare_VAT_dataset(sequence_length, validation_length, refresh, device, small=False, supersmall=False,
                        dataset='MAPS', rank=0):
    """model/helper_functions.py:51-117 (same corpora per `train_on` value); 'Synthetic' is the extra option used for
    plumbing runs and benchmarks.  `rank` offsets the crop seed so that data-parallel ranks draw different windows."""
    seed = 42 + rank
    if dataset == 'Synthetic':
        n = 4 if supersmall else (16 if small else 64)
        l_set = SyntheticSegments(n, sequence_length, seed=1 + 100 * rank, device=device)
        ul_set = SyntheticSegments(4 * n, sequence_length, seed=2 + 100 * rank, device=device)
        val = SyntheticSegments(4, validation_length, seed=3, device=device)
        return l_set, ul_set, val, val
    if dataset == 'MAPS':
        groups = ['AkPnBcht'] if small else ['AkPnBcht', 'AkPnBsdf', 'AkPnCGdD', 'AkPnStgb', 'SptkBGAm', 'SptkBGCl', 'StbgTGd2']
        l_set = MAPS(groups=groups, sequence_length=sequence_length, overlap=False, device=device, refresh=refresh,
                     supersmall=supersmall and small, seed=seed)
        ul_set = MAESTRO(groups=['train'], sequence_length=sequence_length, device=device, seed=seed)
        test_pianos = ['ENSTDkAm', 'ENSTDkCl']
        val = MAPS(groups=test_pianos, sequence_length=validation_length, overlap=True, device=device, refresh=refresh)
        full = MAPS(groups=test_pianos, sequence_length=None, device=device, refresh=refresh)
        return l_set, ul_set, val, full
    musicnet = {'Violin': ('violin', 'violin'), 'String': ('string', 'violin'), 'Wind': ('wind', 'wind'), 'Flute': ('flute', 'flute')}
    if dataset in musicnet:
        train_key, test_key = musicnet[dataset]
        l_set = MusicNet(groups=[f'train_{train_key}_l'], sequence_length=sequence_length, device=device, seed=seed)
        ul_set = MusicNet(groups=[f'train_{train_key}_ul'], sequence_length=sequence_length, device=device, seed=seed)
        val = MusicNet(groups=[f'test_{test_key}'], sequence_length=validation_length, device=device)
        full = MusicNet(groups=[f'test_{test_key}'], sequence_length=None, device=device)
        return l_set, ul_set, val, full
    if dataset == 'Guqin':
        l_set = Guqin(groups=['train_l'], sequence_length=sequence_length, device=device, refresh=refresh, seed=seed)
        ul_set = Guqin(groups=['train_ul'], sequence_length=sequence_length, device=device, refresh=refresh, seed=seed)
        val = Guqin(groups=['test'], sequence_length=validation_length, device=device, refresh=refresh)
        full = Guqin(groups=['test'], sequence_length=None, device=device, refresh=refresh)
        return l_set, ul_set, val, full
    raise ValueError(f"train_on must be one of MAPS, Violin, String, Wind, Flute, Guqin, Synthetic (got {dataset!r})")
