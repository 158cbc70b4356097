import sys, time, torch
sys.path.insert(0, '.')
import reconvat_amd as ra
dev = torch.device('cuda:0')
torch.manual_seed(0)
for cls in (ra.UNet_Onset, ra.UNet):
    m = cls((2, 2), (2, 2), log=True, reconstruction=True, mode='imagewise', spec='Mel', XI=1e-6, eps=2).to(dev).eval()
    for frames in (9999, 18750):          # 5 min 20 s and 10 min of audio
        audio = (torch.rand(1, frames * 512) * 0.2 - 0.1).to(dev)
        lab = (torch.rand(1, frames, 88) > 0.95).float().to(dev)
        batch = {'audio': audio, 'onset': lab, 'frame': lab}
        with torch.no_grad():
            t0 = time.time(); pred, losses, spec = m.run_on_batch(batch); torch.cuda.synchronize(); dt = time.time() - t0
        ok = all(torch.isfinite(v).all().item() for v in pred.values() if torch.is_tensor(v))
        print(cls.__name__, frames, 'frames:', tuple(pred['frame'].shape), 'finite', ok, f'{dt*1e3:.0f} ms', {k: round(float(v), 4) for k, v in list(losses.items())[:3]})
