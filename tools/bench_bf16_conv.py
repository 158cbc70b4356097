"""fp32 vs bf16-operand persistent 3x3 conv kernel (rv_conv_fwd algo bit 20) at the BASELINE layer shapes: the on-line tuner times
every legal tile of both and the best of each is printed (us per launch, TFLOP/s, speed-up)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconvat_amd import ops

dev = torch.device('cuda:0')
ops.AUTOTUNE = True
B = 8
SHAPES = [(16, 16, 640, 229), (32, 32, 320, 114), (64, 64, 160, 57), (128, 128, 80, 28), (192, 96, 80, 28), (96, 48, 160, 57),
          (48, 96, 160, 57), (96, 192, 80, 28), (48, 24, 320, 114), (16, 32, 320, 114), (32, 64, 160, 57), (64, 128, 80, 28)]
tot = [0.0, 0.0]
for cin, cout, h, w in SHAPES:
    x = torch.randn(B, h, w, cin, device=dev)
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    bias = torch.zeros(cout, device=dev)
    out = torch.empty(B, h, w, cout, device=dev)
    pack = ops._pack('c3', wt, 'fwd')
    us = []
    for bf in (False, True):
        ops._conv_call(0, x, cin, B, h, w, cin, out, cout, h, w, cout, pack, bias, bf16=bf)
        key = (0, B, h, w, cin, cout, cin, cout, False, False) + (('bf16',) if bf else ())
        us.append((ops._tune_us[('conv', key)], ops._algo_cache[key]))
    fl = 2.0 * B * h * w * cin * cout * 9
    tot[0] += us[0][0]; tot[1] += us[1][0]
    print(f'{cin:4d}->{cout:4d} {h}x{w}: fp32 {us[0][0]:7.1f} us ({fl / us[0][0] / 1e6:6.1f} TF/s, algo {us[0][1]:#x})   '
          f'bf16 {us[1][0]:7.1f} us ({fl / us[1][0] / 1e6:6.1f} TF/s, algo {us[1][1]:#x})   x{us[0][0] / us[1][0]:.2f}')
print(f'sum: fp32 {tot[0]:.0f} us, bf16 {tot[1]:.0f} us, x{tot[0] / tot[1]:.2f}')
