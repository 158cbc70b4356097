"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

For every case the reference implementation (imported unmodified through
``_refload``) is run on closed-form fixture weights / inputs
(``oracle/fixture.py``), the oracle restatement is run on the same inputs and
must agree (this is what pins the oracle), and the reference's outputs are
written as small ``.npz`` fixtures.  Large tensors are stored as
(l2-norm, strided sample) pairs -- see ``digest``.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _refload  # noqa: E402
from oracle import fixture as fx  # noqa: E402
from oracle import frontend as ofe  # noqa: E402
from oracle import model as om  # noqa: E402
from oracle import onset_frames as oo  # noqa: E402

torch.set_num_threads(8)
ref = _refload.load_reference()
DS = ((2, 2), (2, 2))


def digest(t, n=96):
    """(norm, strided sample) summary of a tensor -- keeps fixtures small."""
    f = t.detach().double().flatten()
    stride = max(1, f.numel() // n)
    return np.concatenate([[f.norm().item()], f[::stride][:n].numpy()]).astype(np.float64)


def close(a, b, tol, what):
    a, b = a.detach().double(), b.detach().double()
    err = (a - b).abs().max().item()
    scale = max(b.abs().max().item(), 1e-30)
    assert err <= tol * scale, f'oracle != reference for {what}: max err {err:.3e} (scale {scale:.3e})'
    return err / scale


def build_ref(kind, reconstruction, training=True, xi=1e-6, eps=2.0):
    cls = ref.UNet_Onset if kind == 'onset' else ref.UNet
    net = cls(*DS, log=True, reconstruction=reconstruction, mode='imagewise', spec='Mel', XI=xi, eps=eps)
    params = fx.fixture_params(kind, reconstruction)
    missing = net.load_state_dict(params, strict=True)
    net.train(training)
    return net, fx.clone_params(params)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB')


# ---------------------------------------------------------------------------------------------
def g_frontend():
    net, params = build_ref('onset', False)
    audio = fx.fixture_audio(2, 65536)[:, :-1]
    mel_ref = net.spectrogram(audio)
    mel_orc = ofe.melspec_power(audio, params)
    close(mel_orc, mel_ref, 1e-6, 'mel power')
    ln_ref = net.normalize.transform(torch.log(mel_ref + 1e-5)).transpose(-1, -2).unsqueeze(1)
    close(ofe.log_normalise(mel_orc), ln_ref, 1e-6, 'log-normalised mel')
    # independent check of the un-pinned nnAudio restatement: torch.stft power
    st = torch.stft(audio, 2048, 512, window=torch.from_numpy(ofe.hann_periodic().astype(np.float32)),
                    center=True, pad_mode='reflect', return_complex=True).abs() ** 2
    rel = ((params['spectrogram.mel_basis'] @ st - mel_ref).abs().max() / mel_ref.abs().max()).item()
    assert rel < 1e-4, rel
    # one full-length clip, digest only
    a2 = fx.fixture_audio(1, 327680, 'audio_full')[:, :-1]
    ln2 = ofe.frontend(a2, params)
    close(ln2, net.normalize.transform(torch.log(net.spectrogram(a2) + 1e-5)).transpose(-1, -2).unsqueeze(1),
          1e-6, 'full clip')
    save('frontend', mel=mel_ref, lognorm=ln_ref, full_digest=digest(ln2, 512),
         mel_nnz=(params['spectrogram.mel_basis'] != 0).sum().item(), stft_rel=rel)


def g_unet():
    """Encoder+Decoder of the onset transcriber: fwd, input grad, weight grads, BN running stats."""
    net, params = build_ref('onset', False)
    x = fx.fixture_spec(2, 64).requires_grad_(True)
    cot = fx.hashed('cot_unet', (2, 2, 64, 229))
    t = net.transcriber
    xe, s, c = t.Unet1_encoder(x)
    y = t.Unet1_decoder(xe, s, c)
    (y * cot).sum().backward()

    xo = x.detach().clone().requires_grad_(True)
    for k in om.trainable_keys(params):
        params[k].requires_grad_(True)
    yo = om.unet(om.Net(params, True), xo, 'transcriber.Unet1_encoder', 'transcriber.Unet1_decoder')
    (yo * cot).sum().backward()
    close(yo, y, 1e-5, 'unet fwd')
    close(xo.grad, x.grad, 1e-4, 'unet dx')
    out = dict(y=y, dx=x.grad, x4=digest(xe))
    sd = dict(net.named_parameters())
    for k, p in sd.items():
        if p.grad is None or 'Unet1' not in k:
            continue
        close(params[k].grad, p.grad, 2e-4, 'grad ' + k)
        out['g:' + k] = digest(p.grad)
    for k, b in net.state_dict().items():
        if k.endswith(('running_mean', 'running_var')) and 'Unet1' in k:
            close(params[k], b, 1e-5, k)
            out['s:' + k] = b.clone()
    save('unet', **out)


def g_attention():
    out = {}
    for tag, fin, fout, groups in (('t176', 176, 768, 6), ('r88', 88, 916, 4), ('f229', 229, 916, 4)):
        att = ref.UNet_onset.MutliHeadAttention1D(fin, fout, 31, position=True, groups=groups)
        p = {'a.rel': fx.hashed_normalish(tag + 'rel', (1, fout, 31), 0.5)}
        for w in ('W_q', 'W_k', 'W_v'):
            p[f'a.{w}.weight'] = fx.hashed(tag + w, (fout, fin), float(np.sqrt(3.0 / fin)))
        att.load_state_dict({k[2:]: v for k, v in p.items()})
        x = fx.hashed(tag + 'x', (2, 64, fin), 1.0).requires_grad_(True)
        cot = fx.hashed(tag + 'cot', (2, 64, fout), 1.0)
        o, a = att(x)
        (o * cot).sum().backward()
        po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        xo = x.detach().clone().requires_grad_(True)
        oo, ao = om.local_attention(om.Net(po), xo, 'a', groups)
        (oo * cot).sum().backward()
        close(oo, o, 1e-5, tag + ' out'); close(ao, a, 1e-5, tag + ' att'); close(xo.grad, x.grad, 1e-4, tag + ' dx')
        out.update({f'{tag}_out': o, f'{tag}_att': a, f'{tag}_dx': x.grad,
                    f'{tag}_drel': att.rel.grad,
                    f'{tag}_dWq': digest(att.W_q.weight.grad, 256),
                    f'{tag}_dWk': digest(att.W_k.weight.grad, 256),
                    f'{tag}_dWv': digest(att.W_v.weight.grad, 256)})
        for w in ('W_q', 'W_k', 'W_v'):
            close(po[f'a.{w}.weight'].grad, getattr(att, w).weight.grad, 1e-4, tag + w)
        close(po['a.rel'].grad, att.rel.grad, 1e-4, tag + 'rel')
    save('attention', **out)


def g_networks():
    out = {}
    for kind in ('onset', 'frame'):
        net, params = build_ref(kind, True)
        x = fx.fixture_spec(2, 128, 'spec_net')
        with torch.no_grad():
            r = net(x)
            if kind == 'onset':
                o = om.forward_onset(params, True, x, True)
                names = ('rec', 'roll', 'onset', 'roll2', 'onset2', 'att')
            else:
                o = om.forward_frame(params, True, x, True)
                names = ('rec', 'roll', 'roll2', 'att')
        for n, a, b in zip(names, r, o):
            close(b, a, 2e-5, f'{kind} forward {n}')
            out[f'{kind}_{n}'] = a if n != 'att' else digest(a, 512)
        # eval mode (running statistics) after that one training forward
        net.eval()
        with torch.no_grad():
            r = net(x)
            o = om.forward_onset(params, False, x, True) if kind == 'onset' else om.forward_frame(params, False, x, True)
        for n, a, b in zip(names, r, o):
            close(b, a, 2e-5, f'{kind} eval forward {n}')
            if n in ('roll', 'rec'):
                out[f'{kind}_eval_{n}'] = a
    # one full-size clip through the onset transcriber
    net, params = build_ref('onset', False)
    x = fx.fixture_spec(1, 640, 'spec_full')
    with torch.no_grad():
        roll, onset, a = net.transcriber(x)
        ro, oo, ao = om.spec2roll_onset(om.Net(params, True), x)
    close(ro, roll, 2e-5, 'full roll'); close(oo, onset, 2e-5, 'full onset')
    out['full_roll'] = roll; out['full_onset'] = onset
    save('networks', **out)


def g_vat():
    out = {}
    real_randn_like = torch.randn_like
    for kind in ('onset', 'frame'):
        for tag, xi, eps in (('wc', 1e-1, 2.0), ('real', 1e-6, 2.0)):
            net, params = build_ref(kind, False, xi=xi, eps=eps)
            x = fx.fixture_spec(2, 64, 'spec_vat')
            d0 = fx.fixture_noise(x.shape, 'd0_' + kind)
            grabbed = {}

            def fake(t, **kw):
                d = d0.clone()
                if kw.get('requires_grad'):
                    d.requires_grad_(True)
                grabbed['d'] = d
                return d
            torch.randn_like = fake
            try:
                lds, r_adv, dn = net.vat_loss(net, x)
            finally:
                torch.randn_like = real_randn_like
            g = grabbed['d'].grad
            if kind == 'onset':
                lo, ro, dno, go = om.vat_onset(params, True, x, xi, eps, d0)
                close(lo['frame'], lds['frame'], 1e-3, 'lds frame'); close(lo['onset'], lds['onset'], 1e-3, 'lds onset')
                out[f'{kind}_{tag}_lds'] = np.array([lds['frame'].item(), lds['onset'].item()])
            else:
                lo, ro, dno, go = om.vat_frame(params, True, x, xi, eps, d0)
                close(lo, lds, 1e-3, 'lds')
                out[f'{kind}_{tag}_lds'] = np.array([lds.item()])
            out[f'{kind}_{tag}_rnorm'] = dn.abs().mean().item()
            out[f'{kind}_{tag}_radv_rownorm'] = r_adv.norm(dim=-1).flatten()[:8]
            if tag == 'wc':
                close(go, g, 1e-3, 'd.grad'); close(ro, r_adv, 1e-3, 'r_adv')
                out[f'{kind}_{tag}_g'] = g
                out[f'{kind}_{tag}_radv'] = r_adv
            print(kind, tag, 'cos(r_adv ref, oracle) =',
                  F.cosine_similarity(ro.flatten(), r_adv.flatten(), dim=0).item())
    save('vat', **out)


def _batch(b, t, tag):
    onset, frame = fx.fixture_labels(b, t, tag)
    return {'audio': fx.fixture_audio(b, t * 512, tag), 'onset': onset, 'frame': frame}


def g_run_on_batch():
    out = {}
    real_randn_like = torch.randn_like
    for kind in ('onset', 'frame'):
        for recon in (False, True):
            for vat in (False, True):
                for training in (True, False):
                    net, params = build_ref(kind, recon, training)
                    bl, bul = _batch(2, 64, 'L'), _batch(2, 64, 'UL')
                    noises = [fx.fixture_noise((2, 1, 64, 229), 'd0_ul'), fx.fixture_noise((2, 1, 64, 229), 'd0_l')]
                    use_ul = vat and training
                    seq = list(noises if use_ul else noises[1:])

                    def fake(t, **kw):
                        d = seq.pop(0).clone()
                        return d.requires_grad_(True) if kw.get('requires_grad') else d
                    torch.randn_like = fake
                    try:
                        pr, lr, sr = net.run_on_batch(bl, bul if use_ul else None, vat)
                    finally:
                        torch.randn_like = real_randn_like
                    fn = om.run_on_batch_onset if kind == 'onset' else om.run_on_batch_frame
                    po, lo, so = fn(params, training, bl, bul if use_ul else None, vat, recon,
                                    d0_l=noises[1], d0_ul=noises[0])
                    assert list(lo.keys()) == list(lr.keys()), (list(lo.keys()), list(lr.keys()))
                    key = f'{kind}_r{int(recon)}_v{int(vat)}_t{int(training)}'
                    for k in lr:
                        close(lo[k], lr[k], 1e-3 if 'LDS' in k else 2e-5, key + k)
                    close(so, sr, 1e-6, 'spec')
                    close(po['frame'], pr['frame'], 2e-5, 'frame')
                    out[key + '_losses'] = np.array([v.item() for v in lr.values()])
                    out[key + '_keys'] = np.array(list(lr.keys()))
                    out[key + '_frame'] = digest(pr['frame'], 256)
                    if 'reconstruction' in pr:
                        out[key + '_rec'] = digest(pr['reconstruction'], 256)
    save('run_on_batch', **out)


def _ref_losses(kind, recon, training, bl, bul, noises, vat=True, dtype=torch.float32, threads=8, application=False):
    """The REFERENCE's run_on_batch (or run_on_batch_application) with injected VAT noise, at a thread count / dtype."""
    real_randn_like = torch.randn_like
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        net, _ = build_ref(kind, recon, training)
        if dtype == torch.float64:
            net = net.double()
        cast = lambda d: {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
        seq = [n.to(dtype) for n in noises]

        def fake(t, **kw):
            d = seq.pop(0).clone()
            return d.requires_grad_(True) if kw.get('requires_grad') else d
        torch.randn_like = fake
        fn = net.run_on_batch_application if application else net.run_on_batch
        pr, lr, sr = fn(cast(bl), cast(bul) if bul is not None else None, vat)
    finally:
        torch.randn_like = real_randn_like
        torch.set_num_threads(prev)
    return pr, lr, sr


def g_lds_spread():
    """How far the REFERENCE moves its own loss terms when nothing but the arithmetic changes: every VAT-carrying fixture
    the GPU tests compare against, re-run at 1 thread (fp32) and in fp64, next to the 8-thread fp32 values the other goldens
    hold.  The GPU tests accept |HIP - reference| <= max(1e-3, 3 x spread) per loss key.  Also the
    full-size anchor of the bench workload: B = 2, T = 640, VAT + reconstruction, both models, injected noise."""
    out = {}
    cases = []
    for kind in ('onset', 'frame'):
        cases.append((f'{kind}_T64', kind, 64, ('d0_ul', 'd0_l')))        # run_on_batch.npz  <kind>_r1_v1_t1
        cases.append((f'{kind}_T64_step', kind, 64, ('d0_0', 'd0_1')))    # train_step.npz
        cases.append((f'{kind}_T640', kind, 640, ('d0_ul', 'd0_l')))      # full-size anchor
        cases.append((f'{kind}_T32_smoke', kind, 32, ('smoke_ul', 'smoke_l')))
    for tag, kind, T, ntags in cases:
        if 'smoke' in tag:
            bl = _batch(2, T, 'smoke'); bul = bl
        else:
            bl, bul = _batch(2, T, 'L'), _batch(2, T, 'UL')
        noises = [fx.fixture_noise((2, 1, T, 229), n) for n in ntags]
        runs = {}
        # round 6: the 2- and 4-thread fp32 runs as well -- a FOUR-sample estimate of the reference's own noise
        # (1, 2, 4 threads and fp64 against the 8-thread golden) instead of a two-sample one; `_spread2` keeps the two-sample figure
        for name, dtype, threads in (('f32_8t', torch.float32, 8), ('f32_1t', torch.float32, 1), ('f64', torch.float64, 8),
                                     ('f32_2t', torch.float32, 2), ('f32_4t', torch.float32, 4)):
            pr, lr, sr = _ref_losses(kind, True, True, bl, bul, noises, dtype=dtype, threads=threads)
            runs[name] = (pr, lr)
            out[f'{tag}_{name}'] = np.array([float(v) for v in lr.values()], dtype=np.float64)
        out[f'{tag}_keys'] = np.array(list(runs['f32_8t'][1].keys()))
        base = out[f'{tag}_f32_8t']
        den = np.maximum(np.abs(base), 1e-12)
        out[f'{tag}_spread2'] = np.maximum(np.abs(out[f'{tag}_f32_1t'] - base), np.abs(out[f'{tag}_f64'] - base)) / den
        spread = np.max([np.abs(out[f'{tag}_{n}'] - base) for n in ('f32_1t', 'f32_2t', 'f32_4t', 'f64')], axis=0) / den
        out[f'{tag}_spread'] = spread
        print(tag, {k: f'{s_:.1e}' for k, s_ in zip(out[f'{tag}_keys'], spread)})
        if T == 640:
            pr = runs['f32_8t'][0]
            for k in ('frame', 'onset', 'frame2', 'reconstruction'):
                out[f'{tag}_{k}'] = digest(pr[k], 512)
            p64 = runs['f64'][0]
            out[f'{tag}_frame_f64_err'] = float((p64['frame'].float() - pr['frame']).abs().max())
        if T == 64 and 'step' not in tag:
            # the oracle on the same case (what bench.py's cpu leg and smoke() use as the checker)
            fn = om.run_on_batch_onset if kind == 'onset' else om.run_on_batch_frame
            _, lo, _ = fn(fx.fixture_params(kind, True), True, bl, bul, True, True, d0_l=noises[1], d0_ul=noises[0])
            out[f'{tag}_oracle'] = np.array([float(v) for v in lo.values()])
    # VAT alone at the real XI on the `vat.npz` inputs (recon=False models)
    for kind in ('onset', 'frame'):
        x = fx.fixture_spec(2, 64, 'spec_vat')
        d0 = fx.fixture_noise(x.shape, 'd0_' + kind)
        vals = {}
        for name, dtype, threads in (('f32_8t', torch.float32, 8), ('f32_1t', torch.float32, 1), ('f64', torch.float64, 8),
                                     ('f32_2t', torch.float32, 2), ('f32_4t', torch.float32, 4)):
            torch.set_num_threads(threads)
            net, _ = build_ref(kind, False, xi=1e-6, eps=2.0)
            if dtype == torch.float64:
                net = net.double()
            real = torch.randn_like
            torch.randn_like = lambda t, **kw: (d0.to(dtype).clone().requires_grad_(True) if kw.get('requires_grad') else d0.to(dtype).clone())
            try:
                lds, _, _ = net.vat_loss(net, x.to(dtype))
            finally:
                torch.randn_like = real
                torch.set_num_threads(8)
            vals[name] = np.array([float(lds['frame']), float(lds['onset'])] if kind == 'onset' else [float(lds)])
        base = vals['f32_8t']
        out[f'vat_{kind}_f32_8t'] = base
        out[f'vat_{kind}_spread2'] = np.maximum(np.abs(vals['f32_1t'] - base), np.abs(vals['f64'] - base)) / np.abs(base)
        out[f'vat_{kind}_spread'] = np.max([np.abs(vals[n] - base) for n in ('f32_1t', 'f32_2t', 'f32_4t', 'f64')], axis=0) / np.abs(base)
        print('vat', kind, out[f'vat_{kind}_spread'])
    save('lds_spread', **out)


def g_anchor_b8():
    """The full-size anchor at the BENCH's own batch: B_l = B_ul = 8 segments of 327 680 samples, UNet_Onset, VAT + reconstruction,
    injected noise -- the reference's eleven loss values at 8 threads and at 1 thread (fp32; an fp64 run of this size does not fit this
    container's memory, so the spread of this case is the 1-thread movement only) and digests of its posteriorgrams.  The GPU test and
    bench.py's parity leg run exactly this shape with exactly the shipped tiles."""
    out = {}
    tag, kind, T, B = 'onset_T640_B8', 'onset', 640, 8
    bl, bul = _batch(B, T, 'L'), _batch(B, T, 'UL')
    noises = [fx.fixture_noise((B, 1, T, 229), n) for n in ('d0_ul', 'd0_l')]
    runs = {}
    for name, threads in (('f32_8t', 8), ('f32_1t', 1)):
        pr, lr, sr = _ref_losses(kind, True, True, bl, bul, noises, dtype=torch.float32, threads=threads)
        runs[name] = {k: v.detach() for k, v in pr.items() if torch.is_tensor(v)}
        out[f'{tag}_{name}'] = np.array([float(v) for v in lr.values()], dtype=np.float64)
        if name == 'f32_8t':
            out[f'{tag}_keys'] = np.array(list(lr.keys()))
        del pr, lr, sr
    base = out[f'{tag}_f32_8t']
    out[f'{tag}_spread'] = np.abs(out[f'{tag}_f32_1t'] - base) / np.maximum(np.abs(base), 1e-12)
    print(tag, {k: f'{s_:.1e}' for k, s_ in zip(out[f'{tag}_keys'], out[f'{tag}_spread'])})
    for k in ('frame', 'onset', 'frame2', 'reconstruction'):
        out[f'{tag}_{k}'] = digest(runs['f32_8t'][k], 512)
    # BASELINE config 2 at the script's own batch sizes (train_UNet_VAT.py:54,56: train_batch_size = 1, batch_size = 8): the no-onset
    # UNet, VAT + reconstruction, ONE labelled and EIGHT unlabelled full segments
    tag = 'frame_T640_B1_8'
    bl, bul = _batch(1, T, 'L'), _batch(8, T, 'UL')
    noises = [fx.fixture_noise((8, 1, T, 229), 'd0_ul'), fx.fixture_noise((1, 1, T, 229), 'd0_l')]
    runs = {}
    for name, threads in (('f32_8t', 8), ('f32_1t', 1)):
        pr, lr, sr = _ref_losses('frame', True, True, bl, bul, noises, dtype=torch.float32, threads=threads)
        runs[name] = {k: v.detach() for k, v in pr.items() if torch.is_tensor(v)}
        out[f'{tag}_{name}'] = np.array([float(v) for v in lr.values()], dtype=np.float64)
        if name == 'f32_8t':
            out[f'{tag}_keys'] = np.array(list(lr.keys()))
        del pr, lr, sr
    base = out[f'{tag}_f32_8t']
    out[f'{tag}_spread'] = np.abs(out[f'{tag}_f32_1t'] - base) / np.maximum(np.abs(base), 1e-12)
    print(tag, {k: f'{s_:.1e}' for k, s_ in zip(out[f'{tag}_keys'], out[f'{tag}_spread'])})
    for k in ('frame', 'frame2', 'reconstruction'):
        out[f'{tag}_{k}'] = digest(runs['f32_8t'][k], 512)
    save('anchor_b8', **out)


def g_anchor_grads():
    """Full-size PARAMETER GRADIENTS against the reference: B = 2 segments of 327 680 samples (640 frames),
    reconstruction on, both models, the reference's own loop rule (model/helper_functions.py:589-600: every LDS key x alpha/2,
    everything else x 1, `loss.backward()`), in two deterministic modes:
      * `novat`: run_on_batch(batch, None, False) -- no VAT term at all (model/UNet_onset.py:380-405,460-483);
      * `radv` : VAT on with n_power = 0 -- the reference's power-iteration loop never runs and the patched `randn_like` goes straight
        into r_adv = eps * d / ||d|| (:129-151), so the WHOLE eleven-term step (both VAT branches, reconstruction branch) is
        deterministic and its gradient is comparable between implementations.
    Each at 8 threads fp32 (the reference as shipped) and in fp64 (the yardstick of the `e_gpu <= 2 x e_cpu32 + 2e-3` bar).
    Stored per parameter: (norm, strided sample of <= 512 values) of both runs, and the loss values."""
    out = {}
    real_randn_like = torch.randn_like
    T, B, N = 640, 2, 512
    for kind in ('onset', 'frame'):
        for mode in ('novat', 'radv'):
            bl, bul = _batch(B, T, 'L'), _batch(B, T, 'UL')
            noises = [fx.fixture_noise((B, 1, T, 229), 'radv_ul'), fx.fixture_noise((B, 1, T, 229), 'radv_l')]
            tag = f'{kind}_{mode}'
            for name, dtype in (('f32', torch.float32), ('f64', torch.float64)):
                torch.set_num_threads(8)
                net, _ = build_ref(kind, True)
                if dtype == torch.float64:
                    net = net.double()
                cast = lambda d: {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
                seq = [n.to(dtype) for n in noises]

                def fake(t, **kw):
                    d = seq.pop(0).clone()
                    return d.requires_grad_(True) if kw.get('requires_grad') else d
                torch.randn_like = fake
                try:
                    if mode == 'radv':
                        net.vat_loss.n_power = 0
                        pr, lr, _ = net.run_on_batch(cast(bl), cast(bul), True)
                    else:
                        pr, lr, _ = net.run_on_batch(cast(bl), None, False)
                finally:
                    torch.randn_like = real_randn_like
                loss = 0                                   # model/helper_functions.py:589-595 with alpha = 1
                for key in lr:
                    loss = loss + (0.5 * lr[key] if key.startswith('loss/train_LDS') else lr[key])
                loss.backward()
                named = dict(net.named_parameters())
                out[f'{tag}_{name}_losses'] = np.array([float(v) for v in lr.values()], dtype=np.float64)
                if name == 'f32':
                    out[f'{tag}_keys'] = np.array(list(lr.keys()))
                    out[f'{tag}_nograd'] = np.array([k for k, p in named.items() if p.grad is None])
                    out[f'{tag}_gmax'] = max(p.grad.abs().max().item() for p in named.values() if p.grad is not None)
                    for k in ('frame', 'frame2', 'reconstruction'):
                        out[f'{tag}_{k}'] = digest(pr[k], 512)
                for k, p in named.items():
                    if p.grad is not None:
                        d = digest(p.grad, N)
                        out[f'{tag}_{name}_g:' + k] = d.astype(np.float32) if name == 'f32' else d
                del net, pr, lr, loss, named
            # what the bar will be made of: the reference's own fp32 error against its fp64 run, on the stored samples
            errs = []
            for k in [k[len(f'{tag}_f64_g:'):] for k in out if k.startswith(f'{tag}_f64_g:')]:
                a, b = out[f'{tag}_f32_g:' + k].astype(np.float64)[1:], out[f'{tag}_f64_g:' + k][1:]
                errs.append(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
            print(tag, 'losses', dict(zip(out[f'{tag}_keys'], np.round(out[f'{tag}_f32_losses'], 5))),
                  f'| reference fp32 vs fp64 gradient error per tensor: median {np.median(errs):.2e}, max {np.max(errs):.2e}')
    save('anchor_grads', **out)


def g_application():
    """UNet.run_on_batch_application (model/self_attention_VAT.py:1205-1291) and UNet.transcribe (:1293-1314), reference run."""
    out = {}
    bl, bul = _batch(2, 64, 'L'), _batch(2, 64, 'UL')
    noises = [fx.fixture_noise((2, 1, 64, 229), 'd0_ul'), fx.fixture_noise((2, 1, 64, 229), 'd0_l')]
    for training in (True, False):
        pr, lr, sr = _ref_losses('frame', True, training, bl, bul, noises, application=True)
        po, lo, so = om.run_on_batch_application(fx.fixture_params('frame', True), training, bl, bul, True, d0_l=noises[1], d0_ul=noises[0])
        assert list(lo.keys()) == list(lr.keys()) and list(po.keys()) == list(pr.keys())
        key = f't{int(training)}'
        for k in lr:
            close(lo[k], lr[k], 1e-3 if 'LDS' in k else 2e-5, key + k)
        close(so, sr, 1e-6, 'spec')
        out[key + '_keys'] = np.array(list(lr.keys()))
        out[key + '_pred_keys'] = np.array(list(pr.keys()))
        out[key + '_losses'] = np.array([float(v) for v in lr.values()])
        out[key + '_frame'] = digest(pr['frame'], 256)
        out[key + '_frame_shape'] = np.array(pr['frame'].shape)
        if training:
            close(po['ul_frame2'], pr['ul_frame2'], 2e-5, 'ul_frame2')
            out[key + '_ul_frame'] = digest(pr['ul_frame'], 256)
            out[key + '_ul_frame2'] = digest(pr['ul_frame2'], 256)
    net, params = build_ref('frame', True, False)
    with torch.no_grad():
        tr = net.transcribe(bl)
    out['transcribe_frame'] = digest(tr['frame'], 256)
    out['transcribe_keys'] = np.array(list(tr.keys()))
    save('application', **out)


def g_train_step():
    """One iteration of the reference's own train_VAT_model (torch Adam + StepLR, decay every step so
    the LR path is exercised).  Adam's first update is lr*sign(g): for weights whose gradient is
    rounding noise the sign is arbitrary, so parameters are pinned by the FRACTION that agree and the
    gradients (post-step clip_grad_norm_, helper_functions.py:606-607) by value."""
    out = {}
    real_randn_like = torch.randn_like
    for kind in ('onset', 'frame'):
        net, params = build_ref(kind, True)
        opt = torch.optim.Adam(net.parameters(), 1e-3)
        sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.98)

        class Loader(list):
            batch_size = 2
        bl, bul = _batch(2, 64, 'L'), _batch(2, 64, 'UL')
        noises = [fx.fixture_noise((2, 1, 64, 229), f'd0_{i}') for i in range(2)]
        seq = [n.clone() for n in noises]

        def fake(t, **kw):
            d = seq.pop(0)
            return d.requires_grad_(True) if kw.get('requires_grad') else d
        torch.randn_like = fake
        try:
            pr, lr, _ = ref.train_VAT_model(net, 1, 1, Loader([bl]), Loader([bul]), opt, sched, 3, 1, True, 0)
        finally:
            torch.randn_like = real_randn_like
        fn = om.run_on_batch_onset if kind == 'onset' else om.run_on_batch_frame
        state = {}
        po, lo, tot = om.train_step(params, state, 0, bl, bul, fn, alpha=1.0, lr0=1e-3, decay_steps=1,
                                    decay_rate=0.98, clip=3.0, VAT=True, reconstruction=True,
                                    d0_ul=noises[0], d0_l=noises[1])
        for k in lr:
            e = close(lo[k], lr[k], 1e-3 if 'LDS' in k else 2e-5, 'step ' + k)
        sd = net.state_dict()
        named = dict(net.named_parameters())
        agree = total = 0
        gmax = max(p.grad.abs().max().item() for p in named.values() if p.grad is not None)
        out[f'{kind}_gmax'] = gmax
        for k in om.trainable_keys(params):
            diff = (params[k].detach() - sd[k]).abs()
            agree += (diff < 1e-5).sum().item(); total += diff.numel()
            out[f'{kind}_p:' + k] = digest(sd[k], 32)
            if named[k].grad is not None:
                # conv biases feeding a train-mode BN have an analytically zero gradient (pure
                # rounding noise) -> tolerance is relative to the tensor AND to the global scale
                err = (params[k].grad - named[k].grad).abs().max().item()
                assert err <= 5e-3 * named[k].grad.abs().max().item() + 1e-5 * gmax, ('grad ' + k, err)
                out[f'{kind}_g:' + k] = digest(named[k].grad, 32)
        print(kind, f'params agreeing to 1e-5 after the Adam step: {agree}/{total}')
        assert agree / total > 0.97
        out[f'{kind}_agree'] = agree / total
        out[f'{kind}_losses'] = np.array([v.item() for v in lr.values()])
        out[f'{kind}_keys'] = np.array(list(lr.keys()))
        out[f'{kind}_total'] = tot.item()
        out[f'{kind}_lr'] = opt.param_groups[0]['lr']
        out[f'{kind}_nograd'] = np.array([k for k, p in named.items() if p.grad is None])
    save('train_step', **out)


TRAJ = fx.TRAJ
trajectory_inputs = fx.trajectory_inputs


def g_trajectory():
    """K = 6 iterations of the REFERENCE's own train_VAT_model (model/helper_functions.py:570-615) -- torch Adam, StepLR(step_size = 2:
    two decay boundaries are crossed), post-step clip_grad_norm_, `cycle`d loaders -- at B = 2, T = 64, reconstruction on, both models,
    in the two modes in which the reference's step is deterministic: `novat` (VAT=False: run_on_batch(batch, None, False)) and `radv`
    (VAT on with n_power = 0: the injected noise goes straight into r_adv, all eleven loss terms incl. both LDS branches).  Each at 8
    threads fp32 (the reference as shipped: the golden), at 1, 2 and 4 threads fp32 (three more draws of its rounding noise) and in fp64
    (the yardstick).  The trajectory amplifies rounding noise -- Adam's first updates are lr * sign(g) -- so a product trajectory is held
    to the reference's OWN drift: e <= 2 x (the worst of the four fp32 runs against the fp64 run) + 1e-3 (tests/trajectory_check.py).
    Stored: the loss terms and the learning rate of every iteration (all runs); after the sixth step (norm, strided sample) of every
    parameter and of Adam's exp_avg / exp_avg_sq and every BatchNorm running_mean / running_var in full (8-thread fp32 and fp64 runs),
    num_batches_tracked, and per tensor the worst error of the four fp32 runs against the fp64 run (`eref`).  The oracle is run through
    the same six steps and must meet the product's bar."""
    sys.path.insert(0, os.path.dirname(HERE))
    import trajectory_check as tc
    c = TRAJ
    out = {}
    real_randn_like = torch.randn_like
    K, N = c['K'], c['N']

    class Loader(list):
        batch_size = c['B']
    for kind in ('onset', 'frame'):
        for mode in ('novat', 'radv'):
            lbs, ubs, noises = trajectory_inputs(mode)
            tag = f'{kind}_{mode}'
            final = {}
            for name, dtype, threads in (('f64', torch.float64, 8), ('f32', torch.float32, 8), ('f32_1t', torch.float32, 1),
                                         ('f32_2t', torch.float32, 2), ('f32_4t', torch.float32, 4)):
                torch.set_num_threads(threads)
                net, _ = build_ref(kind, True)
                if dtype == torch.float64:
                    net = net.double()
                cast = lambda d: {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
                opt = torch.optim.Adam(net.parameters(), c['lr'])
                sched = torch.optim.lr_scheduler.StepLR(opt, step_size=c['step_size'], gamma=c['gamma'])
                rec = {'losses': [], 'lr': []}
                orig = net.run_on_batch

                def wrapped(b, bu, vat, _orig=orig, _rec=rec, _opt=opt):
                    _rec['lr'].append(_opt.param_groups[0]['lr'])            # the rate this iteration's optimizer.step() will use
                    pr, lr_, sp = _orig(b, bu, vat)
                    _rec['losses'].append([float(v.detach()) for v in lr_.values()])
                    _rec['keys'] = list(lr_.keys())
                    return pr, lr_, sp
                net.run_on_batch = wrapped
                seq = [n.to(dtype) for pair in noises for n in pair]           # per iteration: unlabelled first, labelled second

                def fake(t, **kw):
                    d = seq.pop(0).clone()
                    return d.requires_grad_(True) if kw.get('requires_grad') else d
                torch.randn_like = fake
                try:
                    if mode == 'radv':
                        net.vat_loss.n_power = 0
                    ref.train_VAT_model(net, K, 1, Loader([cast(b) for b in lbs]), Loader([cast(b) for b in ubs]), opt, sched,
                                        c['clip'], 1, mode == 'radv', 0)
                finally:
                    torch.randn_like = real_randn_like
                    torch.set_num_threads(8)
                assert len(rec['losses']) == K and (mode == 'novat' or not seq)
                out[f'{tag}_{name}_losses'] = np.array(rec['losses'], dtype=np.float64)
                named = dict(net.named_parameters())
                if name == 'f32':
                    out[f'{tag}_keys'] = np.array(rec['keys'])
                    out[f'{tag}_lr'] = np.array(rec['lr'] + [opt.param_groups[0]['lr']], dtype=np.float64)      # K rates used + the rate after step K
                    out[f'{tag}_nograd'] = np.array([k for k, p in named.items() if p.grad is None])
                dig = {}
                for k, p in named.items():
                    dig['p:' + k] = digest(p, N)
                    st = opt.state.get(p)
                    if st:
                        dig['m:' + k] = digest(st['exp_avg'], N // 2)
                        dig['v:' + k] = digest(st['exp_avg_sq'], N // 4)
                for k, b in net.state_dict().items():
                    if k.endswith(('running_mean', 'running_var')):
                        dig['s:' + k] = b.detach().double().numpy().copy()
                    elif k.endswith('num_batches_tracked'):
                        dig['s:' + k] = np.array(int(b))
                final[name] = dig
                if name in ('f32', 'f64'):
                    small = (lambda d: d.astype(np.float32) if d.ndim else d) if name == 'f32' else (lambda d: d)   # (the fp32 run's digests as float32: exact)
                    for k, d in dig.items():
                        out[f'{tag}_{name}_' + k] = small(d) if not k.startswith('s:') else d
                del net, opt, sched
            # ---- the reference's own noise: per tensor the WORST of its four fp32 runs against its fp64 run (the product's yardstick) ----
            shapes = {k: tuple(v.shape) for k, v in fx.fixture_params(kind, True).items()}
            for what in ('p', 'm', 'v', 's'):
                names = [k[2:] for k in final['f64'] if k.startswith(what + ':') and not k.endswith('num_batches_tracked')]
                floor = tc.floor_rms({k: final['f64'][what + ':' + k] for k in names}, shapes) if what != 's' else 0.0
                eref = [max(tc.tensor_err(final[r][what + ':' + k], final['f64'][what + ':' + k], floor, digest=what != 's')
                            for r in ('f32', 'f32_1t', 'f32_2t', 'f32_4t')) for k in names]
                out[f'{tag}_eref_{what}_names'] = np.array(names)
                out[f'{tag}_eref_{what}'] = np.array(eref, dtype=np.float64)
            for r in ('f32', 'f32_1t', 'f32_2t', 'f32_4t'):
                for k in final['f64']:
                    if k.endswith('num_batches_tracked'):
                        assert int(final[r][k]) == int(final['f64'][k])
            # ---- the oracle through the same six steps (fp32), through the product's checker ----
            fn = om.run_on_batch_onset if kind == 'onset' else om.run_on_batch_frame
            params, state = fx.clone_params(fx.fixture_params(kind, True)), {}
            losses, lrs = [], []
            for i in range(K):
                kw = dict(VAT=True, d0_ul=noises[i][0], d0_l=noises[i][1], n_power=0) if mode == 'radv' else dict(VAT=False)
                lrs.append(c['lr'] * c['gamma'] ** (i // c['step_size']))
                _, lo, _ = om.train_step(params, state, i, lbs[i % c['n_l']], ubs[i % c['n_ul']] if mode == 'radv' else None, fn, alpha=1.0,
                                         lr0=c['lr'], decay_steps=c['step_size'], decay_rate=c['gamma'], clip=c['clip'], reconstruction=True, **kw)
                losses.append([float(v.detach()) for v in lo.values()])
            lrs.append(c['lr'] * c['gamma'] ** (K // c['step_size']))
            keys_ = om.trainable_keys(params)
            rows = tc.check(tag, losses, lrs, {k: params[k] for k in keys_}, {k: state[k][0] for k in keys_ if k in state},
                            {k: state[k][1] for k in keys_ if k in state},
                            {k: t for k, t in params.items() if k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))}, N,
                            'oracle (generator)', gold=out)
            print(tag, f'final lr {out[tag + "_lr"][-1]:.3e}', tc.summary(rows))
    save('trajectory', **out)


def g_lds_backward():
    """The backward of the LDS terms ALONE with an injected perturbation.  At XI = 1e-6 the power
    iteration's direction is rounding noise, so gradients through it are not comparable between implementations -- but the
    backward GIVEN a perturbation is deterministic: with `n_power = 0` the reference's loop (model/UNet_onset.py:129-142)
    never runs, `d = torch.randn_like(x)` (patched: a closed-form tensor) goes straight into
    r_adv = eps * d / ||d|| (:145), and alpha/2 * sum(LDS terms) back-propagates through
    transcriber(clamp(x + r_adv)) against the no_grad targets -- soft-target BCE, the clamp, every transcriber layer.
    Stored: the reference's LDS values and its parameter gradients of 0.5 * sum(LDS) (digests); the oracle must agree."""
    out = {}
    real_randn_like = torch.randn_like
    for kind in ('onset', 'frame'):
        net, params = build_ref(kind, True)
        net.vat_loss.n_power = 0
        bl, bul = _batch(2, 64, 'L'), _batch(2, 64, 'UL')
        noises = [fx.fixture_noise((2, 1, 64, 229), 'radv_ul'), fx.fixture_noise((2, 1, 64, 229), 'radv_l')]
        seq = [n.clone() for n in noises]

        def fake(t, **kw):
            d = seq.pop(0)
            return d.requires_grad_(True) if kw.get('requires_grad') else d
        torch.randn_like = fake
        try:
            pr, lr, _ = net.run_on_batch(bl, bul, True)
        finally:
            torch.randn_like = real_randn_like
        lds_keys = [k for k in lr if 'LDS' in k]
        (0.5 * sum(lr[k] for k in lds_keys)).backward()
        for k in om.trainable_keys(params):
            params[k].requires_grad_(True)
        fn = om.run_on_batch_onset if kind == 'onset' else om.run_on_batch_frame
        po, lo, _ = fn(params, True, bl, bul, True, True, d0_ul=noises[0], d0_l=noises[1], n_power=0)
        (0.5 * sum(lo[k] for k in lds_keys)).backward()
        for k in lr:
            close(lo[k], lr[k], 2e-5, 'lds_backward ' + k)              # deterministic now: tight also on the LDS terms
        close(po['r_adv'], pr['r_adv'], 1e-6, 'r_adv')
        named = dict(net.named_parameters())
        gmax = max(p.grad.abs().max().item() for p in named.values() if p.grad is not None)
        worst = 0.0
        for k, p in named.items():
            if p.grad is None:
                assert params[k].grad is None, k
                continue
            err = (params[k].grad - p.grad).abs().max().item()
            assert err <= 5e-3 * p.grad.abs().max().item() + 1e-5 * gmax, ('grad ' + k, err)
            worst = max(worst, err / gmax)
            out[f'{kind}_g:' + k] = digest(p.grad, 32)
        print(kind, 'LDS-only gradients: oracle vs reference worst abs err / gmax =', worst)
        out[f'{kind}_gmax'] = gmax
        out[f'{kind}_keys'] = np.array(list(lr.keys()))
        out[f'{kind}_losses'] = np.array([v.item() for v in lr.values()])
        out[f'{kind}_radv'] = digest(pr['r_adv'], 256)
        out[f'{kind}_nograd'] = np.array([k for k, p in named.items() if p.grad is None])
    save('lds_backward', **out)


def g_dataset():
    """The reference's PianoRollAudioDataset.__getitem__ (model/dataset.py:35-69) on in-memory tracks: crop positions
    of a RandomState(42) stream over 12 consecutive items, and the decoded crops themselves (exact)."""
    from oracle import dataset as od
    tracks = od.synthetic_tracks()

    class InMemory(ref.dataset.PianoRollAudioDataset):
        @classmethod
        def available_groups(cls):
            return ['g']

        def files(self, group):
            return [(i, None) for i in range(len(tracks))]

        def load(self, i, _tsv):
            t = tracks[i]
            return dict(path=t['path'], audio=torch.from_numpy(t['audio']), label=torch.from_numpy(t['label']),
                        velocity=torch.from_numpy(t['velocity']))

    seq = 16384
    ds = InMemory('.', sequence_length=seq, seed=42)
    rs = np.random.RandomState(42)
    out = {'order': [], 'start_idx': []}
    order = [0, 1, 2, 2, 1, 0, 0, 0, 1, 2, 1, 2]
    for n, idx in enumerate(order):
        item = ds[idx]
        step_begin, begin = od.draw_begin(rs, len(tracks[idx]['audio']), seq)
        mine = od.crop_item(tracks[idx], step_begin, seq)
        assert item['start_idx'] == begin == mine['start_idx']
        for k in ('audio', 'onset', 'offset', 'frame', 'velocity'):
            assert np.array_equal(item[k].numpy(), mine[k]), (n, k)          # bit-exact
            if n < 3:
                out[f'{n}_{k}'] = item[k].numpy()
        out['order'].append(idx)
        out['start_idx'].append(begin)
        out.setdefault('audio_sum', []).append(float(item['audio'].double().sum()))
        out.setdefault('frame_sum', []).append(float(item['frame'].sum()))
        out.setdefault('onset_sum', []).append(float(item['onset'].sum()))
        out.setdefault('velocity_sum', []).append(float(item['velocity'].double().sum()))
    # whole-track item (sequence_length=None): no crop, velocity float
    full = InMemory('.', sequence_length=None)[1]
    out['full_len'] = len(full['audio'])
    out['full_audio_sum'] = float(full['audio'].double().sum())
    out['full_frame_sum'] = float(full['frame'].sum())
    save('dataset', **{k: np.asarray(v) for k, v in out.items()})


def g_ingest():
    """model/dataset.py:85-142 (`load`: audio + tsv -> int16 audio / uint8 label / velocity rolls) and the group -> file rules
    of MAPS (:182-214, incl. overlapping.pkl and supersmall) and MusicNet (:238-342), run by the REFERENCE classes on the
    synthetic corpus above (soundfile -- absent here -- stubbed with a wav reader; the corpus is wav with the .flac name the
    reference globs for)."""
    import shutil
    import tempfile
    from scipy.io import wavfile
    from oracle import dataset as od
    root = od.ingest_corpus(tempfile.mkdtemp())
    # the reference globs '*.flac' only: give every wav a .flac twin (same bytes; the stubbed reader sniffs nothing)
    for dirpath, _dirs, files in os.walk(root):
        for f in files:
            if f.endswith('.wav'):
                shutil.copy(os.path.join(dirpath, f), os.path.join(dirpath, f[:-4] + '.flac'))

    def sf_read(path, dtype='int16'):
        sr, pcm = wavfile.read(path)
        return pcm, sr
    sys.modules['soundfile'].read = sf_read
    ref.dataset.soundfile.read = sf_read
    cwd = os.getcwd()
    os.chdir(root)                                    # the reference opens 'overlapping.pkl' relative to the cwd
    out = {}
    try:
        def summarise(tag, ds):
            out[tag + '_paths'] = np.array([os.path.relpath(d['path'], root)[:-5] for d in ds.data])
            out[tag + '_label_sum'] = np.array([int(d['label'].long().sum()) for d in ds.data])
            out[tag + '_label_w'] = np.array([int((d['label'].long() * torch.arange(1, 89)).sum()) for d in ds.data])
            out[tag + '_vel_sum'] = np.array([int(d['velocity'].long().sum()) for d in ds.data])
            out[tag + '_steps'] = np.array([d['label'].shape[0] for d in ds.data])
            out[tag + '_audio_sum'] = np.array([int(d['audio'].long().sum()) for d in ds.data])
        summarise('maps_small', ref.dataset.MAPS(path='MAPS', groups=['AkPnBcht'], overlap=False, refresh=True))
        summarise('maps_supersmall', ref.dataset.MAPS(path='MAPS', groups=['AkPnBcht'], overlap=False, supersmall=True, refresh=True))
        summarise('maps_test', ref.dataset.MAPS(path='MAPS', groups=['ENSTDkAm'], overlap=True, refresh=True))
        d0 = ref.dataset.MAPS(path='MAPS', groups=['ENSTDkAm'], overlap=True, refresh=True).data[0]
        out['maps_test_label0'] = d0['label'].numpy()
        out['maps_test_velocity0'] = d0['velocity'].numpy()
        for g in ('train_string_l', 'train_string_ul', 'train_violin_l', 'train_violin_ul', 'test_violin', 'train_wind_l',
                  'train_wind_ul', 'test_wind', 'train_flute_l', 'train_flute_ul', 'test_flute'):
            summarise('mn_' + g, ref.dataset.MusicNet(path='MusicNet', groups=[g], refresh=True))
    finally:
        os.chdir(cwd)
        shutil.rmtree(root)
    save('ingest', **out)


def decoding_rolls(seed, T=300, P=88):
    """Smooth random posteriorgrams with note-like runs (shared with tests/test_decoding.py through this recipe)."""
    rng = np.random.RandomState(seed)
    frames = np.zeros((T, P), np.float32)
    onsets = np.zeros((T, P), np.float32)
    for _ in range(120):
        t0, p, ln = rng.randint(0, T), rng.randint(0, P), rng.randint(1, 40)
        frames[t0:t0 + ln, p] = rng.uniform(0.3, 1.0)
        if rng.rand() < 0.8:
            onsets[t0:min(T, t0 + rng.randint(1, 4)), p] = rng.uniform(0.3, 1.0)
    onsets += rng.uniform(0, 0.2, size=onsets.shape).astype(np.float32)
    frames += rng.uniform(0, 0.2, size=frames.shape).astype(np.float32)
    velocity = rng.uniform(0, 1, size=frames.shape).astype(np.float32)
    return onsets, frames, velocity


def g_decoding():
    """model/decoding.py (extract_notes_wo_velocity rule1 / rule2, extract_notes, notes_to_frames) on seeded rolls."""
    from reconvat_amd import decoding as md
    out = {}
    for seed in (0, 1, 2):
        on, fr, vel = (torch.from_numpy(a) for a in decoding_rolls(seed))
        for rule in ('rule1', 'rule2'):
            p, i = ref.decoding.extract_notes_wo_velocity(on, fr, 0.5, 0.5, rule=rule)
            p2, i2 = md.extract_notes_wo_velocity(on, fr, 0.5, 0.5, rule=rule)
            assert np.array_equal(p, p2) and np.array_equal(i, i2), (seed, rule)
            out[f'{seed}_{rule}_p'], out[f'{seed}_{rule}_i'] = p, i
        p, i, v = ref.decoding.extract_notes(on, fr, vel, 0.4, 0.6)
        p2, i2, v2 = md.extract_notes(on, fr, vel, 0.4, 0.6)
        assert np.array_equal(p, p2) and np.array_equal(i, i2) and np.allclose(v, v2, rtol=1e-6), seed
        out[f'{seed}_v_p'], out[f'{seed}_v_i'], out[f'{seed}_v_v'] = p, i, v
        t, f = ref.decoding.notes_to_frames(out[f'{seed}_rule1_p'], out[f'{seed}_rule1_i'], fr.shape)
        out[f'{seed}_nf_count'] = np.array([len(x) for x in f])
    save('decoding', **out)


def build_onf(training=True, xi=1e-6, eps=1e-1):
    """The reference's Onsets&Frames baseline (model/onset_frame_VAT.py:603) on the oracle's fixture weights, with the
    drop probability of its nn.Dropout instances set to 0 on the INSTANCES (random masks have no golden value)."""
    net = ref.OnsetsAndFrames_VAT_full(229, 88, model_complexity=48, log=True, mode='imagewise', spec='Mel', XI=xi, eps=eps)
    params = oo.fixture_params()
    net.load_state_dict(params, strict=True)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    net.train(training)
    return net, fx.clone_params(params)


def g_onset_frames():
    out = {}
    real_randn_like = torch.randn_like
    x = fx.fixture_spec(2, 64, 'onf_spec').squeeze(1)
    # forward / backward of the network proper, train and eval mode
    for training in (True, False):
        net, params = build_onf(training)
        o_r, a_r, f_r = net(x)
        o_o, a_o, f_o = oo.forward(params, training, x)
        for nm, a, b in (('onset', o_o, o_r), ('act', a_o, a_r), ('frame', f_o, f_r)):
            close(a, b, 2e-5, f'onf {nm} t{int(training)}')
            out[f'fwd_t{int(training)}_{nm}'] = b
        if training:
            gy = [fx.hashed(f'onf_gy{i}', tuple(o_r.shape), 1.0) for i in range(3)]
            (o_r * gy[0] + a_r * gy[1] + f_r * gy[2]).sum().backward()
            sd = net.state_dict()
            for k in sd:
                if 'running' in k:
                    close(params[k], sd[k], 1e-5, k)
                    out['bn:' + k] = sd[k].clone()
            for k in params:
                if params[k].dtype == torch.float32 and not k.startswith('spectrogram') and 'running' not in k:
                    params[k].requires_grad_(True)
            o_o, a_o, f_o = oo.forward(params, True, x)
            (o_o * gy[0] + a_o * gy[1] + f_o * gy[2]).sum().backward()
            named = dict(net.named_parameters())
            gmax = max(p.grad.abs().max().item() for p in named.values())
            for k, p in named.items():
                err = (params[k].grad - p.grad).abs().max().item()
                assert err <= 2e-3 * p.grad.abs().max().item() + 1e-5 * gmax, ('onf grad ' + k, err)
                out['grad:' + k] = digest(p.grad, 48)
            out['gmax'] = gmax
    # stepwise VAT with injected noise: well-conditioned (XI=1e-1) and the script's own (XI=1e-6, eps=1e-1)
    for tag, xi, eps in (('wc', 1e-1, 2.0), ('real', 1e-6, 1e-1)):
        net, params = build_onf(True, xi, eps)
        d0 = fx.fixture_noise(x.shape, 'onf_d0')
        grabbed = {}

        def fake(t, **kw):
            d = d0.clone()
            if kw.get('requires_grad'):
                d.requires_grad_(True)
            grabbed['d'] = d
            return d
        torch.randn_like = fake
        try:
            lds, r_adv, dn = net.vat_loss(net, x)
        finally:
            torch.randn_like = real_randn_like
        lo, ro, dno, go = oo.vat(params, True, x, xi, eps, d0)
        close(lo, lds, 1e-3, 'onf lds ' + tag)
        out[f'vat_{tag}_lds'] = lds.item()
        out[f'vat_{tag}_rnorm'] = dn.abs().mean().item()
        out[f'vat_{tag}_radv_rownorm'] = r_adv.norm(dim=-1).flatten()[:8]
        if tag == 'wc':
            close(go, grabbed['d'].grad, 1e-3, 'onf d.grad'); close(ro, r_adv, 1e-3, 'onf r_adv')
            out['vat_wc_g'] = grabbed['d'].grad
            out['vat_wc_radv'] = r_adv
        print('onf vat', tag, 'cos(r_adv ref, oracle) =', F.cosine_similarity(ro.flatten(), r_adv.flatten(), dim=0).item())
    # run_on_batch: every (VAT, mode) combination the scripts reach
    for vat in (False, True):
        for training in (True, False):
            net, params = build_onf(training, 1e-6, 1e-1)
            bl, bul = _batch(2, 64, 'L'), _batch(2, 64, 'UL')
            noises = [fx.fixture_noise((2, 64, 229), 'onf_d0_ul'), fx.fixture_noise((2, 64, 229), 'onf_d0_l')]
            use_ul = vat and training
            seq = list(noises if use_ul else noises[1:])

            def fake(t, **kw):
                d = seq.pop(0).clone()
                return d.requires_grad_(True) if kw.get('requires_grad') else d
            torch.randn_like = fake
            try:
                pr, lr, sr = net.run_on_batch(bl, bul if use_ul else None, vat)
            finally:
                torch.randn_like = real_randn_like
            po, lo, so = oo.run_on_batch(params, training, bl, bul if use_ul else None, vat, 1e-6, 1e-1,
                                         d0_l=noises[1], d0_ul=noises[0])
            assert list(lo.keys()) == list(lr.keys()), (list(lo.keys()), list(lr.keys()))
            key = f'rob_v{int(vat)}_t{int(training)}'
            for k in lr:
                close(lo[k], lr[k], 5e-3 if 'r_norm' in k else (1e-3 if 'LDS' in k else 2e-5), key + k)   # XI=1e-6: d.grad is near rounding noise, its direction (r_norm) is ill-conditioned
            close(so, sr, 1e-6, 'onf spec')
            close(po['frame'], pr['frame'], 2e-5, 'onf frame')
            out[key + '_losses'] = np.array([v.item() for v in lr.values()])
            out[key + '_keys'] = np.array(list(lr.keys()))
            out[key + '_frame'] = digest(pr['frame'], 256)
            out[key + '_onset'] = digest(pr['onset'], 256)
    # the single-stack variants of the baseline script (model_name = 'frame' / 'onset')
    def zero_dropout(net):
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    for vat in (False, True):
        for training in (True, False):
            net = ref.Frame_stack_VAT(229, 88, model_complexity=48, log=True, mode='imagewise', spec='Mel', XI=1e-1, eps=2.0, VAT_mode='all')
            params = oo.fixture_params(kind='frame')
            net.load_state_dict(params, strict=True); zero_dropout(net); net.train(training)
            bl = _batch(2, 64, 'L')
            d0 = fx.fixture_noise((2, 64, 229), 'onf_fs_d0')
            torch.randn_like = lambda t, **kw: (d0.clone().requires_grad_(True) if kw.get('requires_grad') else d0.clone())
            try:
                pr, lr, sr = net.run_on_batch(bl, None, vat)
            finally:
                torch.randn_like = real_randn_like
            po, lo, so = oo.run_on_batch_frame_stack(fx.clone_params(params), training, bl, vat, 1e-1, 2.0, d0_l=d0)
            assert list(lo.keys()) == list(lr.keys())
            key = f'fs_v{int(vat)}_t{int(training)}'
            for k in lr:
                close(lo[k], lr[k], 1e-3 if 'LDS' in k else 2e-5, key + k)
            close(po['frame'], pr['frame'], 2e-5, 'fs frame')
            if vat:
                close(po['r_adv'], pr['r_adv'], 1e-3, 'fs r_adv')
                out[key + '_radv'] = digest(pr['r_adv'], 256)
            out[key + '_losses'] = np.array([v.item() for v in lr.values()])
            out[key + '_keys'] = np.array(list(lr.keys()))
            out[key + '_frame'] = digest(pr['frame'], 256)
    for training in (True, False):
        net = ref.Onset_stack_VAT(229, 88, model_complexity=48, log=True, mode='imagewise', spec='Mel', XI=1e-5, eps=10, VAT_mode='all')
        params = oo.fixture_params(kind='onset')
        net.load_state_dict(params, strict=True); zero_dropout(net); net.train(training)
        bl = _batch(2, 64, 'L')
        pr, lr, sr = net.run_on_batch(bl, None, False)
        po, lo, so = oo.run_on_batch_onset_stack(fx.clone_params(params), training, bl)
        assert list(lo.keys()) == list(lr.keys())
        key = f'os_t{int(training)}'
        for k in lr:
            close(lo[k], lr[k], 2e-5, key + k)
        close(po['onset'], pr['onset'], 2e-5, 'os onset')
        out[key + '_losses'] = np.array([v.item() for v in lr.values()])
        out[key + '_keys'] = np.array(list(lr.keys()))
        out[key + '_onset'] = digest(pr['onset'], 256)
    save('onset_frames', **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['frontend', 'unet', 'attention', 'networks', 'vat', 'run_on_batch', 'train_step', 'dataset',
                             'decoding', 'onset_frames', 'ingest', 'lds_spread', 'application', 'lds_backward', 'anchor_b8', 'anchor_grads', 'trajectory']
    for w in which:
        print('==', w)
        globals()['g_' + w]()
