"""Data feed: the oracle's item rule against the golden produced by the reference's own
PianoRollAudioDataset (CPU), and the device-side cropper rv_crop_segments / DeviceCorpus against the oracle (GPU,
bit-exact: integer / byte work)."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), 'golden')


def test_oracle_item_rule_matches_reference_golden():
    from oracle import dataset as od
    g = np.load(os.path.join(G, 'dataset.npz'))
    tracks = od.synthetic_tracks()
    rs = np.random.RandomState(42)
    seq = 16384
    for n, idx in enumerate(g['order']):
        step_begin, begin = od.draw_begin(rs, len(tracks[idx]['audio']), seq)
        assert begin == g['start_idx'][n]
        item = od.crop_item(tracks[idx], step_begin, seq)
        assert float(item['audio'].astype(np.float64).sum()) == g['audio_sum'][n]
        assert float(item['frame'].sum()) == g['frame_sum'][n] and float(item['onset'].sum()) == g['onset_sum'][n]
        assert float(item['velocity'].astype(np.float64).sum()) == g['velocity_sum'][n]
        if n < 3:
            for k in ('audio', 'onset', 'offset', 'frame', 'velocity'):
                assert np.array_equal(item[k], g[f'{n}_{k}']), (n, k)
    assert item['audio'].dtype == np.float32 and item['onset'].shape == (32, 88)


def test_product_host_dataset_matches_oracle(tmp_path):
    """reconvat_amd.dataset.PianoRollAudioDataset (host path) == oracle on the same tracks and seed."""
    from oracle import dataset as od
    from reconvat_amd.dataset import CachedFolder
    tracks = od.synthetic_tracks()
    os.makedirs(tmp_path / 'g')
    for i, t in enumerate(tracks):
        torch.save(dict(path=t['path'], audio=torch.from_numpy(t['audio']), label=torch.from_numpy(t['label']),
                        velocity=torch.from_numpy(t['velocity'])), tmp_path / 'g' / f'{i}.pt')
    ds = CachedFolder(str(tmp_path), ('g',), sequence_length=16384, seed=42)
    rs = np.random.RandomState(42)
    for idx in (0, 1, 2, 2, 1):
        item = ds[idx]
        sb, begin = od.draw_begin(rs, len(tracks[idx]['audio']), 16384)
        want = od.crop_item(tracks[idx], sb, 16384)
        assert item['start_idx'] == begin
        for k in ('audio', 'onset', 'offset', 'frame', 'velocity'):
            assert np.array_equal(item[k].numpy(), want[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize('seq,batch', [(16384, 3), (512, 2), (32768, 5)])
def test_device_corpus_bit_exact(dev, seq, batch):
    from oracle import dataset as od
    from reconvat_amd.feed import DeviceCorpus
    tracks = od.synthetic_tracks(n=5, seed=7)
    dc = DeviceCorpus(tracks, seq, batch, dev, seed=42)
    rs = np.random.RandomState(42)
    order = [4, 0, 3, 3, 1][:batch]
    out = dc.batch(order)
    for b, idx in enumerate(order):
        sb, begin = od.draw_begin(rs, len(tracks[idx]['audio']), seq)
        want = od.crop_item(tracks[idx], sb, seq)
        assert int(out['start_idx'][b]) == begin
        for k in ('audio', 'onset', 'offset', 'frame', 'velocity'):
            assert np.array_equal(out[k][b].cpu().numpy(), want[k]), (b, k)      # bit-exact
    assert out['path'] == [tracks[i]['path'] for i in order]


@pytest.mark.gpu
def test_crop_edges_and_unaligned(dev):
    """First and last legal crop of a track, a crop that is not 16-byte aligned in the corpus, nullable outputs."""
    from oracle import dataset as od
    from reconvat_amd import _lib
    from reconvat_amd._lib import call, ptr, stream
    t = od.synthetic_tracks(n=1, seed=3, min_len=50001, max_len=50002)[0]
    audio = torch.from_numpy(np.concatenate([np.zeros(3, np.int16), t['audio']])).to(dev)      # odd base offset
    label = torch.from_numpy(np.concatenate([np.zeros(5, np.uint8), t['label'].reshape(-1)])).to(dev)
    seq, n_steps = 4096, 8
    last_step = (len(t['audio']) - seq - 1) // 512
    steps = np.array([0, last_step, 7], dtype=np.int64)
    ab = torch.from_numpy(3 + steps * 512).to(dev)
    lb = torch.from_numpy(5 + steps * 88).to(dev)
    oa = torch.empty(3, seq, device=dev)
    on, fr = torch.empty(3, n_steps, 88, device=dev), torch.empty(3, n_steps, 88, device=dev)
    call('rv_crop_segments', ptr(audio), ptr(label), None, ptr(ab), ptr(lb), 3, seq, n_steps, 88, ptr(oa), ptr(on), None, ptr(fr),
         None, stream())
    for b, s in enumerate(steps):
        want = od.crop_item(t, int(s), seq)
        assert np.array_equal(oa[b].cpu().numpy(), want['audio'])
        assert np.array_equal(on[b].cpu().numpy(), want['onset']) and np.array_equal(fr[b].cpu().numpy(), want['frame'])
    assert _lib.load().rv_crop_segments(ptr(audio), ptr(label), None, ptr(ab), ptr(lb), 0, seq, n_steps, 88, ptr(oa), ptr(on),
                                        None, ptr(fr), None, stream()) != 0          # empty batch is an error


@pytest.mark.gpu
def test_device_corpus_full_size_epoch_and_sharding(dev):
    """BASELINE-size items (327 680 samples, batch 8): checksum against the host path; epoch iteration drops the
    partial batch; ranks get disjoint tracks."""
    from oracle import dataset as od
    from reconvat_amd.feed import DeviceCorpus
    tracks = od.synthetic_tracks(n=9, seed=11, min_len=400000, max_len=500000)
    dc = DeviceCorpus(tracks, 327680, 8, dev, seed=42)
    assert len(dc) == 1
    batches = list(dc)
    assert len(batches) == 1 and batches[0]['audio'].shape == (8, 327680) and batches[0]['frame'].shape == (8, 640, 88)
    b0 = batches[0]
    for j, path in enumerate(b0['path']):
        idx = int(path[5:-5])
        sb = int(b0['start_idx'][j]) // 512
        want = od.crop_item(tracks[idx], sb, 327680)
        assert np.array_equal(b0['audio'][j].cpu().numpy(), want['audio'])
        assert np.array_equal(b0['frame'][j].cpu().numpy(), want['frame'])
    r0 = DeviceCorpus(tracks, 327680, 2, dev, rank=0, world=2)
    r1 = DeviceCorpus(tracks, 327680, 2, dev, rank=1, world=2)
    assert set(r0.paths).isdisjoint(r1.paths) and len(r0.paths) + len(r1.paths) == 9


@pytest.mark.gpu
def test_device_loader_shards_ranks_and_refuses_empty_epochs(dev):
    """What cli.run_training builds for a 2-rank run: disjoint track sets per rank, rank-specific item order, and a shard
    smaller than one batch is an error (an epoch of zero batches would make cycle(loader) spin forever)."""
    from types import SimpleNamespace
    from oracle import dataset as od
    from reconvat_amd.feed import device_loader
    tracks = od.synthetic_tracks(n=8, seed=5)
    ds = SimpleNamespace(data=tracks, sequence_length=16384)
    loaders = [device_loader(ds, 2, dev, rank=r, world=2, seed=42 + r) for r in range(2)]
    seen = [set(p for b in ld for p in b['path']) for ld in loaders]
    assert seen[0].isdisjoint(seen[1]) and len(seen[0] | seen[1]) == 8
    with pytest.raises(ValueError, match='fewer than batch_size'):
        device_loader(ds, 8, dev, rank=0, world=2)
    # many batches in flight without a sync: the pinned offset ring must not be overwritten before the device read it
    ld = device_loader(ds, 2, dev, seed=42)
    outs = [ld.batch([i % 8, (i + 3) % 8]) for i in range(12)]
    torch.cuda.synchronize()
    rs = np.random.RandomState(42)
    for i, out in enumerate(outs):
        for j, idx in enumerate([i % 8, (i + 3) % 8]):
            sb, _ = od.draw_begin(rs, len(tracks[idx]['audio']), 16384)
            assert np.array_equal(out['audio'][j].cpu().numpy(), od.crop_item(tracks[idx], sb, 16384)['audio']), (i, j)
