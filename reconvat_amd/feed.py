"""Device-side data feed: the corpus stays in HBM as int16 audio / uint8 rolls and every batch is
cropped and decoded by ONE pair of kernel launches (rv_crop_segments) instead of per-item host slicing, float
conversion and host->device copies.

``DeviceCorpus`` takes the tracks of a ``PianoRollAudioDataset`` (reference model/dataset.py:19-142 contract) and
reproduces ``DataLoader(dataset, batch_size, shuffle=True, drop_last=True)`` + ``__getitem__`` semantics:
  * item order: a fresh ``torch.randperm`` per epoch (the DataLoader's RandomSampler);
  * crop position of every item: ``RandomState(seed).randint(T - L) // 512`` drawn in item order
    (model/dataset.py:41) -- bit-identical batches to the host path for the same seeds (tests/test_feed_gpu.py);
  * with ``world > 1`` rank r keeps tracks r, r + world, ... (per-rank shard, disjoint data on every GPU).
A 8 x 327 680-sample batch is 5 MB of int16 in, 10 MB of fp32 + 9 MB of label masks out: ~10 us of HBM streaming.
There is no CPU fallback: the corpus tensors must live on a HIP device.
"""
import numpy as np
import torch

from .constants import HOP_LENGTH
from ._lib import call, ptr, stream, need_gpu


def _pad_to(n, m):
    return (n + m - 1) // m * m


class DeviceCorpus:
    def __init__(self, tracks, sequence_length, batch_size, device, seed=42, rank=0, world=1, sampler_seed=0):
        tracks = list(tracks)[rank::world]
        if not tracks:
            raise ValueError('DeviceCorpus: no tracks for this rank')
        self.sequence_length = int(sequence_length)
        if self.sequence_length % HOP_LENGTH:
            raise ValueError('sequence_length must be a multiple of HOP_LENGTH (512)')
        self.batch_size = int(batch_size)
        self.device = torch.device(device)
        self.n_keys = int(tracks[0]['label'].shape[1])
        self.paths = [t['path'] for t in tracks]
        self.lengths = np.array([len(t['audio']) for t in tracks], dtype=np.int64)
        if len(tracks) < self.batch_size:
            # an epoch of zero batches would make cycle(loader) spin forever
            raise ValueError(f'DeviceCorpus: rank {rank} of {world} holds {len(tracks)} track(s), fewer than batch_size='
                             f'{self.batch_size}; lower the batch size or the number of ranks')
        if (self.lengths <= self.sequence_length).any():
            raise ValueError('every track must be longer than sequence_length (the reference draws randint(T - L))')
        # concatenate; track starts padded to 8 samples / 16 label bytes so that aligned crops use 16-byte accesses
        a_off, l_off, na, nl = [], [], 0, 0
        for t in tracks:
            a_off.append(na)
            l_off.append(nl)
            na += _pad_to(len(t['audio']), 8)
            nl += _pad_to(t['label'].shape[0] * self.n_keys, 16)
        audio = torch.zeros(na, dtype=torch.int16)
        label = torch.zeros(nl, dtype=torch.uint8)
        velocity = torch.zeros(nl, dtype=torch.uint8)
        for t, ao, lo in zip(tracks, a_off, l_off):
            a = torch.as_tensor(t['audio'])
            audio[ao:ao + a.numel()] = a
            lab = torch.as_tensor(t['label']).reshape(-1)
            label[lo:lo + lab.numel()] = lab
            vel = torch.as_tensor(t['velocity']).reshape(-1)
            velocity[lo:lo + vel.numel()] = vel.to(torch.uint8)
        self.audio, self.label, self.velocity = audio.to(self.device), label.to(self.device), velocity.to(self.device)
        need_gpu(self.audio)
        self.a_off = np.array(a_off, dtype=np.int64)
        self.l_off = np.array(l_off, dtype=np.int64)
        self.random = np.random.RandomState(seed)                  # the reference's crop stream (one draw per item)
        self.sampler = torch.Generator().manual_seed(sampler_seed)  # item order (RandomSampler analogue)
        # crop offsets travel through a ring of pinned staging buffers, each guarded by the event of its last upload:
        # a batch drawn while an earlier upload is still queued never overwrites offsets the device has not read yet
        self._ring = [torch.empty((2, self.batch_size), dtype=torch.int64).pin_memory() for _ in range(4)]
        self._ring_events = [None] * len(self._ring)
        self._ring_pos = 0

    def __len__(self):
        return len(self.paths) // self.batch_size                  # batches per epoch (drop_last=True)

    def draw(self, indices):
        """Crop positions for the given items, in order (model/dataset.py:41,48): (step_begin [B], begin [B])."""
        steps = np.array([int(self.random.randint(self.lengths[i] - self.sequence_length)) // HOP_LENGTH for i in indices],
                         dtype=np.int64)
        return steps, steps * HOP_LENGTH

    def batch(self, indices):
        """One decoded batch on the device for the given track indices (len == batch_size)."""
        indices = [int(i) for i in indices]
        b = len(indices)
        steps, begins = self.draw(indices)
        slot = self._ring_pos
        self._ring_pos = (slot + 1) % len(self._ring)
        if self._ring_events[slot] is not None:
            self._ring_events[slot].synchronize()
        staging = self._ring[slot]
        staging[0, :b] = torch.from_numpy(self.a_off[indices] + begins)
        staging[1, :b] = torch.from_numpy(self.l_off[indices] + steps * self.n_keys)
        dev_begins = staging[:, :b].to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._ring_events[slot] = ev
        n_steps = self.sequence_length // HOP_LENGTH
        out = {'audio': torch.empty((b, self.sequence_length), device=self.device, dtype=torch.float32)}
        for k in ('onset', 'offset', 'frame', 'velocity'):
            out[k] = torch.empty((b, n_steps, self.n_keys), device=self.device, dtype=torch.float32)
        call('rv_crop_segments', ptr(self.audio), ptr(self.label), ptr(self.velocity), ptr(dev_begins[0]), ptr(dev_begins[1]),
             b, self.sequence_length, n_steps, self.n_keys, ptr(out['audio']), ptr(out['onset']), ptr(out['offset']),
             ptr(out['frame']), ptr(out['velocity']), stream())
        out['path'] = [self.paths[i] for i in indices]
        out['start_idx'] = torch.from_numpy(begins)
        return out

    def __iter__(self):
        """One epoch: a random permutation of the tracks in batches of batch_size, last partial batch dropped."""
        perm = torch.randperm(len(self.paths), generator=self.sampler).tolist()
        for i in range(0, len(perm) - self.batch_size + 1, self.batch_size):
            yield self.batch(perm[i:i + self.batch_size])


def device_loader(dataset, batch_size, device, rank=0, world=1, seed=42):
    """DeviceCorpus over the in-memory tracks of a PianoRollAudioDataset (``dataset.data``); with world > 1 every rank
    keeps a disjoint shard (tracks rank, rank + world, ...) and its own item-order stream."""
    return DeviceCorpus(dataset.data, dataset.sequence_length, batch_size, device, seed=seed, rank=rank, world=world,
                        sampler_seed=rank)
