"""Micro-benchmark one conv launch configuration (forward kernel or wgrad) -- used under rocprofv3 --pmc.

    python tools/bench_conv.py fwd c3 16 16 640 229 [reps]
    python tools/bench_conv.py wgrad c3 32 32 320 114 [reps]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconvat_amd import ops

what, kind, cin, cout, h, w = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 20
B = int(os.environ.get('RV_BENCH_B', '8'))
dev = torch.device('cuda:0')
x = torch.rand(B, h, w, cin, device=dev) - 0.5
wshape = {'c3': (cout, cin, 3, 3), 't3': (cin, cout, 3, 3), 'c1': (cout, cin, 1, 1), 'down': (cout, cin, 2, 2), 'up': (cin, cout, 2, 2)}[kind]
wt = (torch.rand(*wshape, device=dev) - 0.5) * 0.1
bias = torch.zeros(cout, device=dev)
ho, wo = ops._out_hw(kind, h, w, None)
y = torch.empty(B, ho, wo, cout, device=dev)
dy = torch.rand(B, ho, wo, cout, device=dev) - 0.5
taps = {'c3': 9, 't3': 9, 'c1': 1, 'down': 4, 'up': 4}[kind]
flops = 2.0 * B * (h * w if kind == 'up' else ho * wo) * cin * cout * taps


BF = os.environ.get('RV_BENCH_BF16') == '1'        # the opt-in bf16-operand variants (3x3 kernels)


def run():
    if what == 'fwd':
        ops.conv_forward_into(kind, x, wt, bias, y, bf16=BF)
    else:
        ops.conv_wgrad(kind, x, dy, wt, True, bf16=BF)


if os.environ.get('BURST'):
    # short bursts after a pause: the clock state of a conv launch inside a real training step (mixed with
    # bandwidth-bound kernels) rather than that of a sustained MFMA loop
    import time
    samples = []
    for _ in range(8):
        torch.cuda.synchronize()
        time.sleep(0.02)
        run(); run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        e1.synchronize()
        samples.append(e0.elapsed_time(e1) / 5)
    ms = sorted(samples)[len(samples) // 2]
    print(f'{what} {kind} {cin}->{cout} {h}x{w}: {ms * 1e3:.1f} us  {flops / ms / 1e9:.1f} TFLOP/s (bursts)')
    sys.exit(0)

for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
e1.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f'{what} {kind} {cin}->{cout} {h}x{w}: {ms * 1e3:.1f} us  {flops / ms / 1e9:.1f} TFLOP/s')
