"""Per-parameter gradient error of the HIP Onsets&Frames model against the CPU oracle (debug helper, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import fixture as fx, onset_frames as oo
from tests.test_onset_frames import build

dev = torch.device('cuda:0')
m = build(dev, True)
x = fx.fixture_spec(2, 64, 'onf_spec').squeeze(1)
params = oo.fixture_params()
for k in params:
    if params[k].dtype == torch.float32 and not k.startswith('spectrogram') and 'running' not in k:
        params[k].requires_grad_(True)
gy = [fx.hashed(f'onf_gy{i}', (2, 64, 88), 1.0) for i in range(3)]
xo = x.clone().requires_grad_(True)
o, a, f = oo.forward(params, True, xo)
(o * gy[0] + a * gy[1] + f * gy[2]).sum().backward()
xg = x.to(dev).requires_grad_(True)
o2, a2, f2 = m(xg)
(o2 * gy[0].to(dev) + a2 * gy[1].to(dev) + f2 * gy[2].to(dev)).sum().backward()
print('fwd', (o2.cpu() - o).abs().max().item(), (a2.cpu() - a).abs().max().item(), (f2.cpu() - f).abs().max().item())
print('dx', ((xg.grad.cpu() - xo.grad).abs().max() / xo.grad.abs().max()).item())
for k, p in m.named_parameters():
    e = (p.grad.cpu() - params[k].grad).abs().max().item()
    s = params[k].grad.abs().max().item()
    print(f'{k:55s} err {e:.3e} scale {s:.3e} rel {e / s:.2e}')

# which side is closer to a float64 evaluation of the same graph?
p64 = {k: (v.detach().double() if v.dtype == torch.float32 else v.clone()) for k, v in oo.fixture_params().items()}
for k in p64:
    if p64[k].dtype == torch.float64 and not k.startswith('spectrogram') and 'running' not in k:
        p64[k].requires_grad_(True)
o, a, f = oo.forward(p64, True, x.double())
(o * gy[0].double() + a * gy[1].double() + f * gy[2].double()).sum().backward()
print('--- vs float64')
for k, p in m.named_parameters():
    if 'cnn' in k and 'onset' in k:
        s = p64[k].grad.abs().max().item()
        print(f'{k:45s} hip-f64 {(p.grad.cpu().double() - p64[k].grad).abs().max().item() / s:.2e}   '
              f'oracle32-f64 {(params[k].grad.double() - p64[k].grad).abs().max().item() / s:.2e}')
