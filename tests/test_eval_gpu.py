"""Evaluation / inference surface driven by the HIP model on the device (reference
model/evaluate_functions.py:20-127, model/self_attention_VAT.py:1205-1314, transcribe_files.py:12-41): whole songs whose
frame count is neither a multiple of 16 nor one of the training sizes, against the CPU oracle's eval-mode forward on the same
inputs (posteriorgrams within 1e-3), and the note / frame metrics computed from both."""
import os

import numpy as np
import pytest
import torch

import parity_tol
from conftest import rel_err

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
DS = ((2, 2), (2, 2))


def build(kind, recon, dev, training=False):
    import reconvat_amd as ra
    from oracle import fixture as fx
    cls = ra.UNet_Onset if kind == 'onset' else ra.UNet
    m = cls(*DS, log=True, reconstruction=recon, mode='imagewise', spec='Mel', XI=1e-6, eps=2.0)
    m.load_state_dict(fx.fixture_params(kind, recon))
    return m.to(dev).train(training)


def song(frames, tag):
    """A deterministic 'whole song': `frames` hops of audio (+1 sample, the models drop the last one) with labels."""
    from oracle import fixture as fx
    onset, frame = fx.fixture_labels(1, frames, tag)
    return {'path': tag, 'audio': fx.fixture_audio(1, frames * 512, tag), 'onset': onset, 'frame': frame}


def test_transcribe_whole_song_vs_oracle(dev):
    """UNet.transcribe on a 2 077-frame clip (odd at every U-Net level: 2077 -> 1038 -> 519 -> 259 -> 129)."""
    from oracle import fixture as fx, model as om
    frames = 2077
    s = song(frames, 'song_a')
    m = build('frame', True, dev)
    with torch.no_grad():
        pred = m.transcribe({'audio': s['audio'].to(dev)})
    assert list(pred.keys()) == ['onset', 'frame'] and pred['frame'].shape == (1, frames, 88)
    params = fx.fixture_params('frame', True)
    with torch.no_grad():
        spec = om._spec(params, s['audio'])
        _, roll, _, _ = om.forward_frame(params, False, spec, True)
    err = (pred['frame'].cpu() - roll).abs().max().item()
    assert err < 1e-3, err
    assert torch.isfinite(pred['frame']).all()


@pytest.mark.parametrize('frames', [1111, 1500])
def test_evaluate_wo_velocity_hip_model_vs_oracle_posteriorgrams(dev, frames):
    """evaluate_wo_velocity with the HIP UNet_Onset (eval mode, reconstruction pass included) on a whole song, against the
    same metric code fed with the oracle's posteriorgrams.  Thresholded metrics can flip on a posteriorgram value that sits
    within rounding of 0.5, so the metric comparison allows a small absolute slack; the posteriorgrams themselves are 1e-3."""
    from oracle import fixture as fx, model as om
    from reconvat_amd.evaluate import evaluate_wo_velocity
    s = song(frames, f'song_{frames}')
    m = build('onset', True, dev)
    dev_item = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in s.items()}
    with torch.no_grad():
        got = evaluate_wo_velocity([dev_item], m, reconstruction=True, onset=True, VAT=True)
    params = fx.fixture_params('onset', True)

    class OracleModel:
        def run_on_batch(self, label, batch_ul=None, VAT=False):
            with torch.no_grad():
                return om.run_on_batch_onset(params, False, label, None, False, True)

    want = evaluate_wo_velocity([s], OracleModel(), reconstruction=True, onset=True, VAT=True)
    assert set(got) == set(want)
    for k in want:
        a, b = float(got[k][0]), float(want[k][0])
        if k.startswith('loss/'):
            assert abs(a - b) <= 1e-3 * max(abs(b), 1e-6), (k, a, b)
        else:
            assert abs(a - b) <= 0.02, (k, a, b)
    # posteriorgrams directly
    with torch.no_grad():
        pred, _, _ = m.run_on_batch(dev_item, None, False)
        po, _, _ = om.run_on_batch_onset(params, False, s, None, False, True)
    for k in ('frame', 'onset', 'frame2', 'onset2'):
        assert (pred[k].cpu() - po[k]).abs().max().item() < 1e-3, k
    assert rel_err(pred['reconstruction'], po['reconstruction']) < 1e-3


def test_run_on_batch_application_golden(dev):
    """UNet.run_on_batch_application against the reference's own outputs (tests/golden/application.npz)."""
    from oracle import fixture as fx
    from test_model_gpu import _batches, close_digest
    g = np.load(os.path.join(G, 'application.npz'))
    spread = np.load(os.path.join(G, 'lds_spread.npz'))
    bl, bul = _batches(dev)
    for training in (True, False):
        m = build('frame', True, dev, training)
        seq = [fx.fixture_noise((2, 1, 64, 229), 'd0_ul').to(dev), fx.fixture_noise((2, 1, 64, 229), 'd0_l').to(dev)]
        m.vat_loss.noise = lambda t, seq=seq: seq.pop(0).clone()
        pred, losses, spec = m.run_on_batch_application(bl, bul, True)
        key = f't{int(training)}'
        assert list(losses.keys()) == list(g[key + '_keys']) and list(pred.keys()) == list(g[key + '_pred_keys'])
        for (k, v), ref in zip(losses.items(), g[key + '_losses']):
            parity_tol.check('frame_T64', k, v.detach(), ref, 'application:' + key)
        assert tuple(pred['frame'].shape) == tuple(g[key + '_frame_shape'])
        close_digest(pred['frame'], g[key + '_frame'], 1e-3, 256)
        if training:
            close_digest(pred['ul_frame'], g[key + '_ul_frame'], 1e-3, 256)
            close_digest(pred['ul_frame2'], g[key + '_ul_frame2'], 1e-3, 256)
    with pytest.raises(UnboundLocalError):
        m.run_on_batch_application(bl, None, True)
    m = build('frame', True, dev)
    with torch.no_grad():
        tr = m.transcribe(bl)
    assert list(tr.keys()) == list(g['transcribe_keys'])
    close_digest(tr['frame'], g['transcribe_frame'], 1e-3, 256)
    assert spread is not None


def test_eval_model_collects_every_loss_key(dev):
    import reconvat_amd as ra
    from test_model_gpu import _batches
    bl, _ = _batches(dev)

    class Loader(list):
        batch_size = 2
    m = build('onset', True, dev, training=True)
    metrics = ra.eval_model(m, 3, Loader([bl, bl]), VAT_start=0, VAT=True)
    assert not m.training
    assert set(metrics) == {'loss/test_reconstruction', 'loss/test_frame', 'loss/test_frame2', 'loss/test_onset', 'loss/test_onset2',
                            'loss/test_LDS_l_frame', 'loss/test_LDS_l_onset', 'loss/test_r_norm_l'}
    assert all(len(v) == 2 for v in metrics.values())
    metrics = ra.eval_model(m, 0, Loader([bl]), VAT_start=5, VAT=True)         # before VAT_start: no VAT pass
    assert metrics['loss/test_LDS_l_frame'] == [0.0]


def test_transcribe_files_to_midi(dev, tmp_path):
    """transcribe_files.transcribe2midi: wav on disk -> HIP model -> note decoding -> MIDI file; the file is parsed back and
    compared with the notes decoded from the oracle's posteriorgram of the same audio."""
    import sys
    from scipy.io import wavfile
    from oracle import fixture as fx, model as om
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import transcribe_files as tf
    from reconvat_amd.decoding import extract_notes_wo_velocity
    from reconvat_amd.midi import parse_midi
    frames = 333
    audio = fx.fixture_audio(1, frames * 512, 'wav_case')[0]
    pcm = (audio * 32768.0).round().clamp(-32768, 32767).to(torch.int16)
    wav = tmp_path / 'in' / 'clip.wav'
    os.makedirs(wav.parent)
    wavfile.write(str(wav), 16000, pcm.numpy())
    m = build('frame', True, dev)
    tf.transcribe2midi([str(wav)], m, dev, str(tmp_path / 'out'))
    mid = tmp_path / 'out' / 'ReconVAT-clip.mid'
    assert mid.exists()
    params = fx.fixture_params('frame', True)
    with torch.no_grad():
        spec = om._spec(params, (pcm.float() / 32768.0).unsqueeze(0))
        _, roll, _, _ = om.forward_frame(params, False, spec, True)
    p_est, i_est = extract_notes_wo_velocity(roll[0].relu(), roll[0].relu(), 0.5, 0.5, rule='rule2')
    notes = parse_midi(str(mid)) if os.path.getsize(mid) > 26 else np.zeros((0, 4))
    # notes straddling the 0.5 threshold by rounding may differ: allow a 2 % mismatch in count, exact timing for the rest
    assert abs(len(notes) - len(p_est)) <= max(2, 0.02 * len(p_est)), (len(notes), len(p_est))
    if len(p_est) and len(notes) == len(p_est):
        want_on = np.sort(np.asarray(i_est)[:, 0] * 512 / 16000)
        assert np.allclose(np.sort(notes[:, 0]), want_on, atol=2e-3)
