"""BASELINE config 3 as a NUMERICS experiment: what bf16-input MFMA (fp32 accumulate, fp32 BatchNorm
statistics / losses / master weights; the XI = 1e-6 power iteration stays fp32) does to the loss terms, the posteriorgrams and the
gradients of the step -- measured with the fp32 kernels by rounding the OPERANDS of the selected convolutions to bf16
(round-to-nearest-even).  Products of two bf16 numbers are exact in fp32, so this is what a bf16-input MFMA with fp32 accumulation
computes, up to the summation order.

The "final graphs" are: unlabelled final pass, labelled final pass, main forward, reconstructor, transcriber on the reconstruction
-- every conv whose weights are live (the no_grad target pass and the detached XI*d pass of the power iteration stay fp32).
    fp32          nothing rounded (the baseline of this tool)
    bwd           bf16 operands in the BACKWARD convs (input and weight gradients) of the final graphs; every forward untouched
    fwd+bwd       ... and in their forward convs, except the main forward (it doubles as the labelled VAT target)
    all fwd+bwd   main forward included

Fixture: tests/golden/lds_spread.npz case onset_T640 (B = 2 x 327 680 samples, closed-form weights / inputs / injected noise); the
reference's own loss values are the yardstick for the loss terms, the fp32 variant for posteriorgrams and gradients.

    python tools/bf16_emulation.py            (one JSON line per variant)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def rb(t):
    return None if t is None else t.to(torch.bfloat16).to(torch.float32)


def main():
    import reconvat_amd as ra
    from reconvat_amd import ops
    from oracle import fixture as fx          # closed-form fixture tensors (a tool: checker input only)
    dev = torch.device('cuda:0')
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'lds_spread.npz'))
    case = 'onset_T640'

    def mk(tag):
        onset, frame = fx.fixture_labels(2, 640, tag)
        return {'audio': fx.fixture_audio(2, 640 * 512, tag).to(dev), 'onset': onset.to(dev), 'frame': frame.to(dev)}
    bl, bul = mk('L'), mk('UL')
    noise = [fx.fixture_noise((2, 1, 640, 229), 'd0_ul').to(dev), fx.fixture_noise((2, 1, 640, 229), 'd0_l').to(dev)]

    mode = {'fwd': False, 'bwd': False, 'main_fwd': False}
    st = {'live': False, 'main': False, 'calls': 0}
    real = {'fwd': ops.conv_forward_into, 'dgrad': ops.conv_dgrad_into, 'wgrad': ops.conv_wgrad,
            'cf': ops.ConvFn.forward, 'cb': ops.ConvFn.backward, 'uf': ops.UpCatFn.forward, 'ub': ops.UpCatFn.backward}

    def fwd_into(kind, x, w, b, out, stats=None):
        if mode['fwd'] and st['live'] and (mode['main_fwd'] or not st['main']):
            return real['fwd'](kind, rb(x), rb(w), b, out, stats)
        return real['fwd'](kind, x, w, b, out, stats)

    def dgrad_into(kind, dy, w, dx, bn_link=None, accumulate=False):
        if mode['bwd'] and st['live']:
            return real['dgrad'](kind, rb(dy), rb(w), dx, bn_link, accumulate)
        return real['dgrad'](kind, dy, w, dx, bn_link, accumulate)

    def wgrad(kind, x, dy, w, want_bias=True, dw_acc=None, db_acc=None):
        if mode['bwd']:                       # weight gradients exist in the final graphs only
            return real['wgrad'](kind, rb(x), rb(dy), w, want_bias, dw_acc, db_acc)
        return real['wgrad'](kind, x, dy, w, want_bias, dw_acc, db_acc)

    def wrap(fwd_key, bwd_key, weight_slots):
        def forward(ctx, *a):
            ctx.live = any(ctx.needs_input_grad[i] for i in weight_slots)        # live weights <=> a final graph
            st['live'] = ctx.live
            try:
                return real[fwd_key](ctx, *a)
            finally:
                st['live'] = False

        def backward(ctx, *a):
            st['live'] = ctx.live
            try:
                return real[bwd_key](ctx, *a)
            finally:
                st['live'] = False
        return staticmethod(forward), staticmethod(backward)
    ops.ConvFn.forward, ops.ConvFn.backward = wrap('cf', 'cb', (1,))
    ops.UpCatFn.forward, ops.UpCatFn.backward = wrap('uf', 'ub', (1, 4))
    ops.conv_forward_into, ops.conv_dgrad_into, ops.conv_wgrad = fwd_into, dgrad_into, wgrad

    def run(name, fwd, bwd, main_fwd):
        mode.update(fwd=fwd, bwd=bwd, main_fwd=main_fwd)
        cls = ra.UNet_Onset
        m = cls((2, 2), (2, 2), log=True, reconstruction=True, mode='imagewise', spec='Mel', XI=1e-6, eps=2)
        m.load_state_dict(fx.fixture_params('onset', True))
        m.to(dev).train()
        opt = ra.FlatAdam(m.parameters(), lr=0.0)
        seq = {'i': 0}

        def draw(t):
            seq['i'] += 1
            return noise[(seq['i'] - 1) % 2].clone()
        m.vat_loss.noise = draw
        # grad-enabled, live-weight transcriber passes of a single-stream step, in order: UL final, MAIN forward, L final, T(recon)
        real_t = m.transcriber.forward
        st['calls'] = 0

        def t_forward(x, detach=False):
            live = torch.is_grad_enabled() and not detach
            if live:
                st['calls'] += 1
            st['main'] = live and st['calls'] == 2
            try:
                return real_t(x, detach)
            finally:
                st['main'] = False
        m.transcriber.forward = t_forward
        opt.zero_grad()
        pred, losses, _ = m.run_on_batch(bl, bul, True)
        ra.weighted_loss(losses, 1.0).backward()
        torch.cuda.synchronize()
        return ({k: float(v.detach()) for k, v in losses.items()},
                {k: pred[k].detach().clone() for k in ('frame', 'onset', 'frame2', 'onset2', 'reconstruction')}, opt.flat_grad.clone(),
                {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})

    keys = [str(k) for k in g[case + '_keys']]
    ref = dict(zip(keys, (float(v) for v in g[case + '_f32_8t'])))
    base = None
    for name, fwd, bwd, main_fwd in (('fp32', False, False, False), ('bwd', False, True, False), ('fwd+bwd', True, True, False),
                                     ('all fwd+bwd', True, True, True)):
        losses, pred, grad, named = run(name, fwd, bwd, main_fwd)
        if base is None:
            base = (losses, pred, grad, named)
        err = {k.split('/')[-1]: abs(losses[k] - ref[k]) / max(abs(ref[k]), 1e-6) for k in keys}
        non_vat = max(v for k, v in err.items() if 'LDS' not in k and 'r_norm' not in k)
        vat = max(v for k, v in err.items() if 'LDS' in k or 'r_norm' in k)
        pdelta = {k: float((pred[k] - base[1][k]).abs().max() / base[1][k].abs().max()) for k in pred}
        gl2 = float((grad - base[2]).norm() / base[2].norm())
        per = sorted(((float((named[n] - base[3][n]).norm() / max(float(base[3][n].norm()), 1e-12)), n) for n in named), reverse=True)
        print(json.dumps({'variant': name, 'loss_rel_err_vs_reference_non_vat_max': float(f'{non_vat:.3e}'),
                          'loss_rel_err_vs_reference_vat_max': float(f'{vat:.3e}'),
                          'posteriorgram_max_rel_delta_vs_fp32': {k: float(f'{v:.3e}') for k, v in pdelta.items()},
                          'gradient_rel_l2_delta_vs_fp32': float(f'{gl2:.3e}'),
                          'gradient_per_tensor_rel_l2_median': float(f'{per[len(per) // 2][0]:.3e}'),
                          'gradient_worst_tensors': [(n, float(f'{e:.3e}')) for e, n in per[:3]]}))


if __name__ == '__main__':
    main()
