"""fp32 vs bf16-operand 3x3 weight-gradient kernel (rv_conv_wgrad mode bit 8) at the BASELINE layer shapes (us per launch)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconvat_amd import ops

dev = torch.device('cuda:0')
B = 8
SHAPES = [(16, 16, 640, 229), (32, 32, 320, 114), (64, 64, 160, 57), (128, 128, 80, 28), (192, 96, 80, 28), (96, 48, 160, 57),
          (48, 24, 320, 114), (16, 32, 320, 114), (32, 64, 160, 57), (64, 128, 80, 28), (16, 8, 640, 229)]
tot = [0.0, 0.0]
for cin, cout, h, w in SHAPES:
    x, dy = torch.randn(B, h, w, cin, device=dev), torch.randn(B, h, w, cout, device=dev)
    wt = torch.zeros(cout, cin, 3, 3, device=dev)
    us = []
    for bf in (False, True):
        for _ in range(3):
            ops.conv_wgrad('c3', x, dy, wt, True, bf16=bf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv_wgrad('c3', x, dy, wt, True, bf16=bf)
        e1.record(); e1.synchronize()
        us.append(e0.elapsed_time(e1) * 100)
    fl = 2.0 * B * h * w * cin * cout * 9
    tot[0] += us[0]; tot[1] += us[1]
    print(f'{cin:4d}->{cout:4d} {h}x{w}: fp32 {us[0]:7.1f} us ({fl / us[0] / 1e6:6.1f} TF/s)   bf16 {us[1]:7.1f} us ({fl / us[1] / 1e6:6.1f} TF/s)   x{us[0] / us[1]:.2f}   (incl. the reduction launch)')
print(f'sum: fp32 {tot[0]:.0f} us, bf16 {tot[1]:.0f} us, x{tot[0] / tot[1]:.2f}')
