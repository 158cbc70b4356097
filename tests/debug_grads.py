"""Debug: per-parameter gradient error of run_on_batch (VAT off) vs the oracle in fp32 and fp64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reconvat_amd as ra
from oracle import fixture as fx, model as om

dev = torch.device('cuda:0')
kind = sys.argv[1] if len(sys.argv) > 1 else 'onset'
recon = True


def mk(tag):
    onset, frame = fx.fixture_labels(2, 64, tag)
    return {'audio': fx.fixture_audio(2, 64 * 512, tag), 'onset': onset, 'frame': frame}


bl = mk('L')
cls = ra.UNet_Onset if kind == 'onset' else ra.UNet
m = cls((2, 2), (2, 2), log=True, reconstruction=recon, mode='imagewise', spec='Mel', XI=1e-6, eps=2)
m.load_state_dict(fx.fixture_params(kind, recon))
m.to(dev).train()
_, losses, _ = m.run_on_batch({k: v.to(dev) for k, v in bl.items()}, None, False)
sum(losses.values()).backward()
fn = om.run_on_batch_onset if kind == 'onset' else om.run_on_batch_frame
res = {}
for dt in (torch.float32, torch.float64):
    params = fx.fixture_params(kind, recon)
    params = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in params.items()}
    for k in om.trainable_keys(params):
        params[k].requires_grad_(True)
    b = {k: v.to(dt) for k, v in bl.items()}
    _, lo, _ = fn(params, True, b, None, False, recon)
    sum(lo.values()).backward()
    res[dt] = (params, lo)
p32, p64 = res[torch.float32][0], res[torch.float64][0]
print('losses gpu / cpu32 / cpu64')
for k in losses:
    print(f'  {k:32s} {float(losses[k]):.7f} {float(res[torch.float32][1][k]):.7f} {float(res[torch.float64][1][k]):.7f}')
print(f'{"param":60s} {"scale":>10s} {"gpu-64":>10s} {"cpu32-64":>10s}')
for k, p in m.named_parameters():
    g64 = p64[k].grad
    if g64 is None:
        continue
    sc = g64.abs().max().item()
    e_gpu = (p.grad.cpu().double() - g64).abs().max().item()
    e_cpu = (p32[k].grad.double() - g64).abs().max().item()
    flag = ' <<<' if e_gpu > 5 * e_cpu + 1e-6 * sc else ''
    print(f'{k:60s} {sc:10.3e} {e_gpu / sc:10.2e} {e_cpu / sc:10.2e}{flag}')
