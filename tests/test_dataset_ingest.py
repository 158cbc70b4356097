"""Corpus ingestion (reference model/dataset.py:85-142 `load` and the MAPS / MusicNet group -> file rules
:182-342): the product's `reconvat_amd.dataset` on the synthetic corpus of `oracle.dataset.ingest_corpus` against what the
REFERENCE classes produced on the same corpus (tests/golden/ingest.npz, written by make_golden.py g_ingest).  Byte / integer
work: exact."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def corpus(tmp_path_factory):
    from oracle import dataset as od
    root = str(tmp_path_factory.mktemp('corpus'))
    od.ingest_corpus(root)
    return root


def _check(g, tag, ds, root):
    assert [os.path.splitext(os.path.relpath(d['path'], root))[0] for d in ds.data] == list(g[tag + '_paths']), tag
    weights = torch.arange(1, 89)
    for i, d in enumerate(ds.data):
        assert d['audio'].dtype == torch.int16 and d['label'].dtype == torch.uint8 and d['velocity'].dtype == torch.uint8
        assert d['label'].shape == (g[tag + '_steps'][i], 88)
        assert int(d['label'].long().sum()) == g[tag + '_label_sum'][i]
        assert int((d['label'].long() * weights).sum()) == g[tag + '_label_w'][i]
        assert int(d['velocity'].long().sum()) == g[tag + '_vel_sum'][i]
        assert int(d['audio'].long().sum()) == g[tag + '_audio_sum'][i]


def test_maps_groups_overlap_rule_and_rolls(corpus):
    from reconvat_amd.dataset import MAPS
    g = np.load(os.path.join(G, 'ingest.npz'))
    pkl = os.path.join(corpus, 'overlapping.pkl')
    maps = os.path.join(corpus, 'MAPS')
    _check(g, 'maps_small', MAPS(maps, ['AkPnBcht'], overlap=False, refresh=True, overlap_list=pkl), corpus)
    _check(g, 'maps_supersmall', MAPS(maps, ['AkPnBcht'], overlap=False, supersmall=True, refresh=True, overlap_list=pkl), corpus)
    test = MAPS(maps, ['ENSTDkAm'], overlap=True, refresh=True)
    _check(g, 'maps_test', test, corpus)
    assert np.array_equal(test.data[0]['label'].numpy(), g['maps_test_label0'])            # the whole roll, bit-exact
    assert np.array_equal(test.data[0]['velocity'].numpy(), g['maps_test_velocity0'])
    # second construction hits the .pt caches written by the first
    again = MAPS(maps, ['ENSTDkAm'], overlap=True)
    assert torch.equal(again.data[0]['label'], test.data[0]['label']) and torch.equal(again.data[0]['audio'], test.data[0]['audio'])


@pytest.mark.parametrize('group', ['train_string_l', 'train_string_ul', 'train_violin_l', 'train_violin_ul', 'test_violin',
                                   'train_wind_l', 'train_wind_ul', 'test_wind', 'train_flute_l', 'train_flute_ul', 'test_flute'])
def test_musicnet_group_rules(corpus, group):
    from reconvat_amd.dataset import MusicNet
    g = np.load(os.path.join(G, 'ingest.npz'))
    _check(g, 'mn_' + group, MusicNet(os.path.join(corpus, 'MusicNet'), [group], refresh=True), corpus)


def test_prepare_vat_dataset_string_config(corpus, monkeypatch):
    """BASELINE config 5 (`train_on=String`): labelled = first recording of each string ensemble, unlabelled = the rest,
    validation = the four test_violin recordings (model/helper_functions.py:77-86)."""
    from reconvat_amd.dataset import prepare_VAT_dataset
    monkeypatch.chdir(corpus)
    g = np.load(os.path.join(G, 'ingest.npz'))
    l_set, ul_set, val, full = prepare_VAT_dataset(4096, 4096, False, 'cpu', dataset='String')
    assert len(l_set) == len(g['mn_train_string_l_paths']) and len(ul_set) == len(g['mn_train_string_ul_paths'])
    assert len(val) == len(full) == 4
    item = l_set[0]
    assert item['audio'].shape == (4096,) and item['frame'].shape == (8, 88)
    assert full[0]['audio'].shape[0] == len(full.data[0]['audio'])


def test_flac_without_decoder_is_a_clear_error(tmp_path):
    from reconvat_amd.dataset import read_audio_int16
    (tmp_path / 'a.flac').write_bytes(b'fLaC')
    with pytest.raises(FileNotFoundError, match='soundfile'):
        read_audio_int16(str(tmp_path / 'a.flac'))
