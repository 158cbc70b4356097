"""Data feed for the training scripts.

``PianoRollAudioDataset`` restates the reference's in-memory dataset contract (model/dataset.py:19-142):
every track is a dict(path, audio int16 [T], label uint8 [n_steps, 88] with 3 = onset, 2 = frame,
1 = offset, velocity uint8) cached as a ``.pt`` file next to the audio; ``__getitem__`` draws
``step_begin = RandomState(seed).randint(len - seq_len) // 512`` and returns float audio / 32768 and the
onset / offset / frame masks (model/dataset.py:35-69).  Decoding .flac/.wav/.tsv into that cache needs
`soundfile`, which is not on the MI355X image: tracks must already be cached (the reference's own first
run, or Preprocessing.ipynb, produces the .pt files).

``SyntheticSegments`` produces seeded random segments of the same shapes -- what the benchmark and the CI
plumbing run use, since none of the corpora can be downloaded here.
"""
import os
from glob import glob

import numpy as np
import torch
from torch.utils.data import Dataset

from .constants import HOP_LENGTH, SAMPLE_RATE, MIN_MIDI, MAX_MIDI


class PianoRollAudioDataset(Dataset):
    def __init__(self, path, groups=None, sequence_length=None, seed=42, refresh=False, device='cpu'):
        self.path = path
        self.groups = groups if groups is not None else self.available_groups()
        self.sequence_length = sequence_length
        self.device = device
        self.random = np.random.RandomState(seed)
        self.refresh = refresh
        self.data = []
        for group in self.groups:
            for input_files in self.files(group):
                self.data.append(self.load(*input_files))

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        data = self.data[index]
        result = dict(path=data['path'])
        if self.sequence_length is not None:
            audio_length = len(data['audio'])
            step_begin = self.random.randint(audio_length - self.sequence_length) // HOP_LENGTH
            n_steps = self.sequence_length // HOP_LENGTH
            begin = step_begin * HOP_LENGTH
            result['audio'] = data['audio'][begin:begin + self.sequence_length].to(self.device)
            result['label'] = data['label'][step_begin:step_begin + n_steps, :].to(self.device)
            result['velocity'] = data['velocity'][step_begin:step_begin + n_steps, :].to(self.device)
            result['start_idx'] = begin
        else:
            result['audio'] = data['audio'].to(self.device)
            result['label'] = data['label'].to(self.device)
            result['velocity'] = data['velocity'].to(self.device).float()
        result['audio'] = result['audio'].float().div_(32768.0)
        result['onset'] = (result['label'] == 3).float()
        result['offset'] = (result['label'] == 1).float()
        result['frame'] = (result['label'] > 1).float()
        result['velocity'] = result['velocity'].float().div_(128.0)
        return result

    @classmethod
    def available_groups(cls):
        raise NotImplementedError

    def files(self, group):
        raise NotImplementedError

    def load(self, audio_path, tsv_path):
        saved = audio_path.replace('.flac', '.pt').replace('.wav', '.pt')
        if os.path.exists(saved) and not self.refresh:
            return torch.load(saved)
        raise FileNotFoundError(
            f'{saved} not found: decoding {audio_path} needs the `soundfile` package, which is not installed here. '
            'Create the .pt caches with the reference (model/dataset.py:85-142) or Preprocessing.ipynb first.')


class MAPS(PianoRollAudioDataset):
    """model/dataset.py:145-182 (groups are the MAPS piano/recording-condition folders)."""

    def __init__(self, path='./MAPS', groups=None, sequence_length=None, overlap=True, seed=42, refresh=False,
                 device='cpu', supersmall=False):
        self.overlap = overlap
        self.supersmall = supersmall
        super().__init__(path, groups if groups is not None else ['ENSTDkAm', 'ENSTDkCl'], sequence_length, seed,
                         refresh, device)

    @classmethod
    def available_groups(cls):
        return ['AkPnBcht', 'AkPnBsdf', 'AkPnCGdD', 'AkPnStgb', 'ENSTDkAm', 'ENSTDkCl', 'SptkBGAm', 'SptkBGCl', 'StbgTGd2']

    def files(self, group):
        flacs = sorted(glob(os.path.join(self.path, 'flac', '*_%s.flac' % group)) +
                       glob(os.path.join(self.path, 'flac', '*_%s.pt' % group)))
        flacs = sorted({f.replace('.pt', '.flac') for f in flacs})
        if self.supersmall:
            flacs = flacs[:1]
        tsvs = [f.replace('/flac/', '/tsv/matched/').replace('.flac', '.tsv') for f in flacs]
        return zip(flacs, tsvs)


class CachedFolder(PianoRollAudioDataset):
    """Any folder of ``*.pt`` track caches (MAESTRO / MusicNet exports): group = sub-directory."""

    def __init__(self, path, groups=('.',), sequence_length=None, seed=42, device='cpu'):
        super().__init__(path, list(groups), sequence_length, seed, False, device)

    def files(self, group):
        pts = sorted(glob(os.path.join(self.path, group, '**', '*.pt'), recursive=True))
        return [(p.replace('.pt', '.flac'), None) for p in pts]


class SyntheticSegments(Dataset):
    """Seeded random segments: audio ~ U(-0.1, 0.1), ~5 % frame / ~1 % onset labels."""

    def __init__(self, n_items=64, sequence_length=327680, seed=0, device='cpu'):
        self.n, self.sequence_length, self.seed, self.device = n_items, sequence_length, seed, device

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1000003 + index)
        steps = self.sequence_length // HOP_LENGTH
        audio = torch.rand(self.sequence_length, generator=g) * 0.2 - 0.1
        u = torch.rand(steps, MAX_MIDI - MIN_MIDI + 1, generator=g)
        frame = (u > 0.95).float()
        onset = (u > 0.99).float()
        return {'path': f'synthetic/{index}', 'audio': audio.to(self.device), 'onset': onset.to(self.device),
                'offset': torch.zeros_like(frame).to(self.device), 'frame': frame.to(self.device),
                'velocity': torch.zeros_like(frame).to(self.device), 'label': (2 * frame + onset).to(torch.uint8)}


def prepare_VAT_dataset(sequence_length, validation_length, refresh, device, small=False, supersmall=False,
                        dataset='MAPS', rank=0):
    """model/helper_functions.py:51-117 for the corpora that can exist on this machine; 'Synthetic' is the
    extra option used for plumbing runs and benchmarks."""
    if dataset == 'Synthetic':
        n = 4 if supersmall else (16 if small else 64)
        l_set = SyntheticSegments(n, sequence_length, seed=1 + 100 * rank, device=device)
        ul_set = SyntheticSegments(4 * n, sequence_length, seed=2 + 100 * rank, device=device)
        val = SyntheticSegments(4, validation_length, seed=3, device=device)
        return l_set, ul_set, val, val
    if dataset == 'MAPS':
        groups = ['AkPnBcht'] if small else ['AkPnBcht', 'AkPnBsdf', 'AkPnCGdD', 'AkPnStgb', 'SptkBGAm', 'SptkBGCl', 'StbgTGd2']
        l_set = MAPS(groups=groups, sequence_length=sequence_length, overlap=False, device=device, refresh=refresh,
                     supersmall=supersmall, seed=42 + rank)
        ul_set = CachedFolder('./MAESTRO', ('.',), sequence_length, seed=42 + rank, device=device)
        val = MAPS(groups=['ENSTDkAm', 'ENSTDkCl'], sequence_length=validation_length, overlap=True, device=device,
                   refresh=refresh)
        full = MAPS(groups=['ENSTDkAm', 'ENSTDkCl'], sequence_length=None, device=device, refresh=refresh)
        return l_set, ul_set, val, full
    if dataset in ('Violin', 'String', 'Wind', 'Flute', 'Guqin'):
        key = dataset.lower()
        l_set = CachedFolder('./MusicNet', (f'train_{key}_l',), sequence_length, seed=42 + rank, device=device)
        ul_set = CachedFolder('./MusicNet', (f'train_{key}_ul',), sequence_length, seed=42 + rank, device=device)
        val = CachedFolder('./MusicNet', (f'test_{key}',), validation_length, device=device)
        full = CachedFolder('./MusicNet', (f'test_{key}',), None, device=device)
        return l_set, ul_set, val, full
    raise ValueError(f"train_on must be one of MAPS, Violin, String, Wind, Flute, Guqin, Synthetic (got {dataset!r})")
