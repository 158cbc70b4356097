"""Data-feed throughput: DeviceCorpus (rv_crop_segments) vs the reference-style host path (per-item slicing, float
conversion, DataLoader collate, host->device copy) on the same synthetic corpus, batch 8 x 327 680 samples."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch.utils.data import DataLoader
from reconvat_amd.feed import DeviceCorpus
from reconvat_amd.dataset import PianoRollAudioDataset

dev = torch.device('cuda:0')
rng = np.random.RandomState(5)
tracks = []
for i in range(32):
    t = int(rng.randint(2_000_000, 3_000_000))
    steps = (t - 1) // 512 + 1
    tracks.append({'path': f'track{i}.flac', 'audio': rng.randint(-32768, 32768, size=t).astype(np.int16),
                   'label': rng.randint(0, 4, size=(steps, 88)).astype(np.uint8),
                   'velocity': rng.randint(0, 128, size=(steps, 88)).astype(np.uint8)})


class Mem(PianoRollAudioDataset):
    @classmethod
    def available_groups(cls):
        return ['g']

    def files(self, group):
        return [(i, None) for i in range(len(tracks))]

    def load(self, i, _):
        t = tracks[i]
        return dict(path=t['path'], audio=torch.from_numpy(t['audio']), label=torch.from_numpy(t['label']),
                    velocity=torch.from_numpy(t['velocity']))


dc = DeviceCorpus(tracks, 327680, 8, dev)
for _ in range(3):
    dc.batch(range(8))
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 50
for i in range(n):
    b = dc.batch([(i * 8 + j) % 32 for j in range(8)])
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / n
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(20):
    dc.batch(range(8))
e1.record(); e1.synchronize()
host = Mem('.', sequence_length=327680, device='cpu')
it = iter(DataLoader(host, 8, shuffle=True, drop_last=True))
t0 = time.perf_counter()
m = 0
for batch in it:
    for k in ('audio', 'onset', 'frame'):
        batch[k] = batch[k].to(dev)
    m += 1
torch.cuda.synchronize()
t_host = (time.perf_counter() - t0) / m
sec = 8 * 327680 / 16000
print(f'device feed : {t_dev * 1e3:7.3f} ms/batch wall ({sec / t_dev:9.0f} audio-s/s), {e0.elapsed_time(e1) / 20 * 1e3:.1f} us GPU time per batch')
print(f'host path   : {t_host * 1e3:7.3f} ms/batch wall ({sec / t_host:9.0f} audio-s/s)')
