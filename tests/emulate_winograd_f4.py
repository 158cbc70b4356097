"""Winograd F(4x4, 3x3) priced on NUMERICS before any kernel (second half).

F(4x4, 3x3) needs 36 products per 16 outputs and (cin, cout) pair where F(2x2, 3x3) needs 64 -- 1.78x fewer MFMAs -- but its transforms
have entries up to 8 and 1/24, and in fp32 its rounding error is roughly an order of magnitude above F(2x2)'s.  The question that decides
whether a kernel is worth building: what does that do to the loss terms at the real XI = 1e-6, where the VAT direction is driven by
rounding noise?  Here: the ORACLE (CPU restatement pinned to the reference) with the forward value of every Winograd-eligible 3x3
convolution (stride 1, pad 1, Cin % 16 == 0: what conv3x3_wino_k serves) replaced by an fp32 F(4x4, 3x3) evaluation (explicit B^T d B,
G g G^T, A^T M A in fp32, Lavin & Gray's matrices); the backward stays the exact fp32 backward.  `--f2` does the same with F(2x2, 3x3)
(what ships) as the yardstick.

    python tests/emulate_winograd_f4.py [--case b2|b8] [--min-width 0] > profiles/r05_winograd_f4_emulation.txt

(a script, not a pytest module; it lives under tests/ because it executes the oracle)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]],
                   dtype=torch.float32)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                  dtype=torch.float32)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float32)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd_conv(x, w, m):
    """fp32 F(m x m, 3x3) of a stride-1 pad-1 convolution: x [B, C, H, W], w [Cout, C, 3, 3] -> [B, Cout, H, W] (no bias)."""
    BT, G, AT = (BT4, G4, AT4) if m == 4 else (BT2, G2, AT2)
    t = m + 2
    b, c, h, wd = x.shape
    nty, ntx = -(-h // m), -(-wd // m)
    xp = F.pad(x, (1, ntx * m - wd + 1, 1, nty * m - h + 1))
    d = xp.unfold(2, t, m).unfold(3, t, m)                                  # [B, C, nty, ntx, t, t]
    v = torch.einsum('ik,bcyxkl,jl->bcyxij', BT, d, BT)                      # B^T d B
    u = torch.einsum('ik,ockl,jl->ocij', G, w, G)                           # G g G^T
    mm = torch.einsum('ocij,bcyxij->boyxij', u, v)                          # sum over input channels, per xi
    y = torch.einsum('ik,boyxkl,jl->boyxij', AT, mm, AT)                    # A^T M A: [B, Cout, nty, ntx, m, m]
    y = y.permute(0, 1, 2, 4, 3, 5).reshape(b, w.shape[0], nty * m, ntx * m)
    return y[:, :, :h, :wd]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--case', default='b2')
    ap.add_argument('--min-width', type=int, default=0, help='only layers whose input is at least this wide (114: the two top resolutions)')
    ap.add_argument('--threads', type=int, default=8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    from oracle import fixture as fx, model as om
    if args.case == 'b8':
        g, case, nb = np.load(os.path.join(ROOT, 'tests', 'golden', 'anchor_b8.npz')), 'onset_T640_B8', 8
    else:
        g, case, nb = np.load(os.path.join(ROOT, 'tests', 'golden', 'lds_spread.npz')), 'onset_T640', 2
    keys = [str(k) for k in g[case + '_keys']]
    ref = dict(zip(keys, (float(v) for v in g[case + '_f32_8t'])))
    spread = dict(zip(keys, (float(v) for v in g[case + '_spread'])))

    def mk(tag):
        onset, frame = fx.fixture_labels(nb, 640, tag)
        return {'audio': fx.fixture_audio(nb, 640 * 512, tag), 'onset': onset, 'frame': frame}
    bl, bul = mk('L'), mk('UL')
    noise = [fx.fixture_noise((nb, 1, 640, 229), 'd0_ul'), fx.fixture_noise((nb, 1, 640, 229), 'd0_l')]
    state = {'m': 0, 'n': 0, 'err': []}
    real_conv, real_convT = om.Net.conv, om.Net.convT

    def patched(real, transposed):
        def f(self, x, name, **kw):
            y = real(self, x, name, **kw)
            w = self.p(name + '.weight')
            if state['m'] == 0 or w.shape[-1] != 3 or kw.get('padding') != 1 or kw.get('stride', 1) != 1 or x.shape[1] % 16 or x.shape[3] < args.min_width:
                return y
            state['n'] += 1
            with torch.no_grad():
                wc = w.detach().flip(2, 3).transpose(0, 1) if transposed else w.detach()       # ConvTranspose2d(k=3, p=1) == conv with flipped / transposed weights
                ye = winograd_conv(x.detach(), wc, state['m']) + self.p(name + '.bias').detach().view(1, -1, 1, 1)
                state['err'].append(float((ye - y).abs().max() / y.abs().max()))
            return y + (ye - y).detach()
        return f
    om.Net.conv, om.Net.convT = patched(real_conv, False), patched(real_convT, True)

    print(f'# Winograd forms in fp32 on the oracle, case {case} (B_l = B_ul = {nb} x 327 680 samples, XI = 1e-6), layers: 3x3, Cin % 16 == 0'
          f'{", input width >= " + str(args.min_width) if args.min_width else ""}; errors relative to the REFERENCE (8 threads fp32); the reference\'s own movement on this '
          f'fixture: VAT terms {max(v for k, v in spread.items() if "LDS" in k or "r_norm" in k):.1e}', flush=True)
    for name, m in (('direct (oracle)', 0), ('F(2x2,3x3)', 2), ('F(4x4,3x3)', 4)):
        state.update(m=m, n=0, err=[])
        t0 = time.time()
        pred, losses, _ = om.run_on_batch_onset(fx.clone_params(fx.fixture_params('onset', True)), True, bl, bul, True, True,
                                                d0_ul=noise[0].clone(), d0_l=noise[1].clone())
        losses = {k: float(v) for k, v in losses.items()}
        err = {k.split('/')[-1]: abs(losses[k] - ref[k]) / max(abs(ref[k]), 1e-6) for k in keys}
        vat = max(v for k, v in err.items() if 'LDS' in k or 'r_norm' in k)
        non = max(v for k, v in err.items() if not ('LDS' in k or 'r_norm' in k))
        print(json.dumps({'form': name, 'convs_replaced_per_step': state['n'],
                          'conv_output_max_rel_err_vs_direct': float(f'{max(state["err"]):.2e}') if state['err'] else 0.0,
                          'loss_rel_err_vs_reference_non_vat_max': float(f'{non:.3e}'), 'loss_rel_err_vs_reference_vat_max': float(f'{vat:.3e}'),
                          'loss_rel_err': {k: float(f'{v:.2e}') for k, v in err.items()}, 'seconds': round(time.time() - t0, 1)}), flush=True)


if __name__ == '__main__':
    main()
