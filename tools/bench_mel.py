"""Time the fused log-Mel front-end on one batch of 8 full segments (327 679 samples -> 640 frames x 229 mel).
Algorithmic HBM bytes per call: 8 x (1.31 MB audio in + 0.59 MB out, written, then read + written by the normalisation)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconvat_amd.frontend import MelSpectrogram
dev = torch.device('cuda:0')
m = MelSpectrogram().to(dev)
x = (torch.rand(8, 327680, device=dev) * 0.2 - 0.1)[:, :-1]
for _ in range(3):
    y = m.lognorm(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    y = m.lognorm(x)
e1.record()
e1.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
mb = 8 * (327679 * 4 + 3 * 640 * 229 * 4) / 1e6
print(f'melspec+log+normalise, 8 segments: {us:.1f} us per call ({mb:.1f} MB algorithmic -> {mb / us * 1e6 / 1e6:.2f} TB/s); '
      f'output range [{float(y.min())}, {float(y.max())}]')
