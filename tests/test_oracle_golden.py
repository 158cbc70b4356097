"""The CPU oracle against the golden vectors produced by running the reference itself
(tests/golden/make_golden.py).  Runs anywhere; needs neither a GPU nor /root/reference."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import fixture as fx, frontend as ofe, model as om

G = os.path.join(os.path.dirname(__file__), 'golden')


def load(name):
    return np.load(os.path.join(G, name + '.npz'), allow_pickle=False)


def digest(t, n=96):
    f = t.detach().double().flatten()
    stride = max(1, f.numel() // n)
    return np.concatenate([[f.norm().item()], f[::stride][:n].numpy()])


def close_digest(t, gold, tol, n=96):
    d = digest(t, n)
    scale = max(np.abs(gold[1:]).max(), 1e-30)
    assert abs(d[0] - gold[0]) <= tol * max(gold[0], 1e-30)
    assert np.abs(d[1:] - gold[1:]).max() <= tol * scale


def test_frontend():
    g = load('frontend')
    bufs = ofe.frontend_buffers()
    assert int((bufs['spectrogram.mel_basis'] != 0).sum()) == int(g['mel_nnz']) == 2025
    audio = fx.fixture_audio(2, 65536)[:, :-1]
    mel = ofe.melspec_power(audio, bufs)
    assert rel_err(mel, torch.from_numpy(g['mel'])) < 1e-5
    ln = ofe.log_normalise(mel)
    assert (ln - torch.from_numpy(g['lognorm'])).abs().max().item() < 1e-5
    assert float(g['stft_rel']) < 1e-4        # independent torch.stft cross-check recorded at generation time


def test_unet_forward_backward():
    g = load('unet')
    params = fx.fixture_params('onset', False)
    x = fx.fixture_spec(2, 64).requires_grad_(True)
    for k in om.trainable_keys(params):
        params[k].requires_grad_(True)
    y = om.unet(om.Net(params, True), x, 'transcriber.Unet1_encoder', 'transcriber.Unet1_decoder')
    (y * fx.hashed('cot_unet', (2, 2, 64, 229))).sum().backward()
    assert rel_err(y, torch.from_numpy(g['y'])) < 1e-5
    assert rel_err(x.grad, torch.from_numpy(g['dx'])) < 1e-4
    for k in g.files:
        if k.startswith('g:'):
            close_digest(params[k[2:]].grad, g[k], 5e-4)
        elif k.startswith('s:'):
            assert rel_err(params[k[2:]], torch.from_numpy(g[k])) < 1e-5


def test_networks():
    g = load('networks')
    x = fx.fixture_spec(2, 128, 'spec_net')
    with torch.no_grad():
        p = fx.fixture_params('onset', True)
        rec, roll, onset, roll2, onset2, a = om.forward_onset(p, True, x, True)
        for n, t in (('rec', rec), ('roll', roll), ('onset', onset), ('roll2', roll2), ('onset2', onset2)):
            assert rel_err(t, torch.from_numpy(g['onset_' + n])) < 5e-5, n
        rec_e, roll_e = om.forward_onset(p, False, x, True)[:2]
        assert rel_err(roll_e, torch.from_numpy(g['onset_eval_roll'])) < 5e-5
        p = fx.fixture_params('frame', True)
        rec, roll, roll2, a = om.forward_frame(p, True, x, True)
        for n, t in (('rec', rec), ('roll', roll), ('roll2', roll2)):
            assert rel_err(t, torch.from_numpy(g['frame_' + n])) < 5e-5, n


def test_vat_well_conditioned():
    g = load('vat')
    x = fx.fixture_spec(2, 64, 'spec_vat')
    p = fx.fixture_params('onset', False)
    lds, r_adv, dn, grad = om.vat_onset(p, True, x, 1e-1, 2.0, fx.fixture_noise(x.shape, 'd0_onset'))
    assert rel_err(grad, torch.from_numpy(g['onset_wc_g'])) < 2e-3
    assert rel_err(r_adv, torch.from_numpy(g['onset_wc_radv'])) < 2e-3
    assert abs(lds['frame'].item() - g['onset_wc_lds'][0]) < 1e-3 * g['onset_wc_lds'][0]
    # every frame of r_adv has L2 norm eps
    assert torch.allclose(r_adv.norm(dim=-1), torch.full_like(r_adv.norm(dim=-1), 2.0), rtol=1e-5)


@pytest.mark.parametrize('kind', ['onset', 'frame'])
def test_run_on_batch_losses(kind):
    g = load('run_on_batch')
    fn = om.run_on_batch_onset if kind == 'onset' else om.run_on_batch_frame
    onset, frame = fx.fixture_labels(2, 64, 'L')
    bl = {'audio': fx.fixture_audio(2, 64 * 512, 'L'), 'onset': onset, 'frame': frame}
    o2, f2 = fx.fixture_labels(2, 64, 'UL')
    bul = {'audio': fx.fixture_audio(2, 64 * 512, 'UL'), 'onset': o2, 'frame': f2}
    n_ul, n_l = fx.fixture_noise((2, 1, 64, 229), 'd0_ul'), fx.fixture_noise((2, 1, 64, 229), 'd0_l')
    for recon, vat, training in ((True, True, True), (False, False, True), (True, True, False)):
        key = f'{kind}_r{int(recon)}_v{int(vat)}_t{int(training)}'
        p = fx.fixture_params(kind, recon)
        use_ul = vat and training
        _, losses, _ = fn(p, training, bl, bul if use_ul else None, vat, recon, d0_l=n_l, d0_ul=n_ul)
        assert list(losses.keys()) == list(g[key + '_keys'])
        for (k, v), ref in zip(losses.items(), g[key + '_losses']):
            tol = 2e-3 if 'LDS' in k else 1e-4
            assert abs(v.item() - ref) <= tol * max(abs(ref), 1e-6), (key, k, v.item(), ref)


def test_oracle_application_path_matches_reference_golden():
    """oracle.model.run_on_batch_application against the reference's own outputs (application.npz)."""
    import torch
    from oracle import fixture as fx, model as om
    g = np.load(os.path.join(G, 'application.npz'))

    def mk(tag):
        onset, frame = fx.fixture_labels(2, 64, tag)
        return {'audio': fx.fixture_audio(2, 64 * 512, tag), 'onset': onset, 'frame': frame}
    bl, bul = mk('L'), mk('UL')
    n_ul, n_l = fx.fixture_noise((2, 1, 64, 229), 'd0_ul'), fx.fixture_noise((2, 1, 64, 229), 'd0_l')
    for training in (True, False):
        pred, losses, _ = om.run_on_batch_application(fx.fixture_params('frame', True), training, bl, bul, True, d0_l=n_l, d0_ul=n_ul)
        key = f't{int(training)}'
        assert list(losses) == list(g[key + '_keys']) and list(pred) == list(g[key + '_pred_keys'])
        for (k, v), ref in zip(losses.items(), g[key + '_losses']):
            assert abs(float(v) - ref) <= (5e-3 if 'LDS' in k or 'r_norm' in k else 1e-4) * max(abs(ref), 1e-6), (k, float(v), ref)


def test_lds_spread_fixture_is_consistent_with_the_other_goldens():
    """lds_spread.npz re-runs the reference on the run_on_batch / train_step fixtures: its 8-thread values must be the ones
    those goldens hold, the non-VAT terms must not move between thread counts / precisions, the VAT terms do (that is the
    point of the fixture)."""
    sp = np.load(os.path.join(G, 'lds_spread.npz'))
    rb = np.load(os.path.join(G, 'run_on_batch.npz'))
    ts = np.load(os.path.join(G, 'train_step.npz'))
    for kind in ('onset', 'frame'):
        assert np.allclose(sp[f'{kind}_T64_f32_8t'], rb[f'{kind}_r1_v1_t1_losses'], rtol=1e-6)
        assert np.allclose(sp[f'{kind}_T64_step_f32_8t'], ts[f'{kind}_losses'], rtol=1e-6)
        for case in (f'{kind}_T64', f'{kind}_T640', f'{kind}_T32_smoke'):
            for k, s in zip(sp[case + '_keys'], sp[case + '_spread']):
                vat = 'LDS' in str(k) or 'r_norm' in str(k)
                assert (1e-5 < s < 1e-2) if vat else (s < 1e-5), (case, k, s)


def test_oracle_fullsize_gradients_match_the_reference_golden():
    """The oracle's full-size backward (B = 2 x 640 frames, reconstruction on, no VAT) against the REFERENCE's own loss.backward()
    stored in tests/golden/anchor_grads.npz: the oracle is the same sequence of torch ops, so losses agree to round-off and the
    gradients far inside the reference's own fp32-vs-fp64 error (what the GPU test's bar is made of)."""
    g = load('anchor_grads')
    tag = 'onset_novat'
    params = fx.fixture_params('onset', True)
    for k in om.trainable_keys(params):
        params[k].requires_grad_(True)
    onset, frame = fx.fixture_labels(2, 640, 'L')
    bl = {'audio': fx.fixture_audio(2, 640 * 512, 'L'), 'onset': onset, 'frame': frame}
    _, lo, _ = om.run_on_batch_onset(params, True, bl, None, False, True)
    assert list(lo.keys()) == [str(k) for k in g[tag + '_keys']]
    for (k, v), ref in zip(lo.items(), g[tag + '_f32_losses']):
        assert abs(float(v.detach()) - float(ref)) <= 2e-5 * max(abs(float(ref)), 1e-6), (k, float(v.detach()), float(ref))
    sum(lo.values()).backward()
    gmax = float(g[tag + '_gmax'])
    nograd = set(str(k) for k in g[tag + '_nograd'])
    checked = 0
    for k in om.trainable_keys(params):
        if k in nograd:
            assert params[k].grad is None, k
            continue
        d32, d64 = g[f'{tag}_f32_g:' + k].astype(np.float64), g[f'{tag}_f64_g:' + k]
        d = digest(params[k].grad, 512)
        den = max(np.linalg.norm(d64[1:]), 1e-4 * gmax * (len(d64) - 1) ** 0.5)
        e_oracle = np.linalg.norm(d[1:] - d32[1:]) / den          # oracle vs the reference's fp32 run (same arithmetic)
        e_ref = np.linalg.norm(d32[1:] - d64[1:]) / den           # the reference's fp32 run vs its own fp64 run
        assert e_oracle <= 0.5 * e_ref + 1e-4, (k, e_oracle, e_ref)
        checked += 1
    assert checked > 90


@pytest.mark.parametrize('tag', ['onset_novat', 'onset_radv', 'frame_novat', 'frame_radv'])
def test_oracle_six_step_trajectory_vs_reference(tag):
    """K = 6 iterations of the oracle's train_step (Adam + StepLR(step_size = 2) + post-step clip, cycled batches, BatchNorm running
    statistics) against the REFERENCE's own train_VAT_model run (tests/golden/trajectory.npz; model/helper_functions.py:570-615) in the
    two deterministic modes -- the CPU side of the GPU test test_six_step_trajectory_vs_reference; bars in tests/trajectory_check.py."""
    import trajectory_check as tc
    kind, mode = tag.split('_')
    c = fx.TRAJ
    lbs, ubs, noises = fx.trajectory_inputs()
    fn = om.run_on_batch_onset if kind == 'onset' else om.run_on_batch_frame
    params, state = fx.clone_params(fx.fixture_params(kind, True)), {}
    losses, lrs = [], []
    for i in range(c['K']):
        kw = dict(VAT=True, d0_ul=noises[i][0], d0_l=noises[i][1], n_power=0) if mode == 'radv' else dict(VAT=False)
        lrs.append(c['lr'] * c['gamma'] ** (i // c['step_size']))
        _, lo, _ = om.train_step(params, state, i, lbs[i % c['n_l']], ubs[i % c['n_ul']] if mode == 'radv' else None, fn, alpha=1.0,
                                 lr0=c['lr'], decay_steps=c['step_size'], decay_rate=c['gamma'], clip=c['clip'], reconstruction=True, **kw)
        assert list(lo.keys()) == [str(k) for k in tc.gold()[tag + '_keys']]
        losses.append([float(v.detach()) for v in lo.values()])
    lrs.append(c['lr'] * c['gamma'] ** (c['K'] // c['step_size']))
    keys = om.trainable_keys(params)
    p = {k: params[k] for k in keys}
    m = {k: state[k][0] for k in keys if k in state}
    v = {k: state[k][1] for k in keys if k in state}
    bufs = {k: t for k, t in params.items() if k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))}
    rows = tc.check(tag, losses, lrs, p, m, v, bufs, c['N'], 'oracle (CPU)')
    s = tc.summary(rows)
    # the oracle is the reference's own op sequence in fp32: it sits where the reference's fp32 run sits (half of the 2 x bar)
    assert s['p']['worst_share_of_bar'] < 0.8 and s['loss']['worst_share_of_bar'] < 0.8, s
