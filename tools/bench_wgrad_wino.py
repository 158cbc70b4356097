"""Winograd F(3x3, 2x2) weight-gradient kernel (plan code nw = 24) against the direct form (the plan-table partition of each shape) on the
3x3 weight-gradient shapes of the BASELINE step (B = 8; partial-sum kernel + its reduction launch, HIP events, best of 3 bursts of 5).

    python tools/bench_wgrad_wino.py
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconvat_amd import ops, _lib, plans

SHAPES = [(16, 16, 640, 229), (16, 8, 640, 229), (16, 32, 320, 114), (32, 32, 320, 114), (48, 24, 320, 114), (24, 16, 320, 114),
          (32, 64, 160, 57), (64, 64, 160, 57), (96, 48, 160, 57), (48, 32, 160, 57), (32, 32, 160, 57),
          (64, 128, 80, 28), (128, 128, 80, 28), (192, 96, 80, 28), (96, 64, 80, 28), (64, 64, 80, 28)]
B = 8
dev = torch.device('cuda:0')
lib = _lib.load()
st = torch.cuda.current_stream()
ops.AUTOTUNE = False


def timed(x, dy, w):
    for _ in range(2):
        ops.conv_wgrad('c3', x, dy, w, True)
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            ops.conv_wgrad('c3', x, dy, w, True)
        e1.record(st)
        e1.synchronize()
        t = e0.elapsed_time(e1) / 5 * 1e3
        best = t if best is None else min(best, t)
    return best


for cin, cout, h, w_ in SHAPES:
    x = torch.rand(B, h, w_, cin, device=dev) - 0.5
    dy = torch.rand(B, h, w_, cout, device=dev) - 0.5
    wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
    plan = plans.lookup_wgrad((9, B, h, w_, cin, cout)) or (8, 256)
    if plan[0] == 24:
        plan = (8, 256)
    assert lib.rv_conv_wgrad_set_plan(9, B, h, cin, cout, *plan) == 0
    t0 = timed(x, dy, wt)
    res = []
    for wgs in (128, 256, 512):
        if lib.rv_conv_wgrad_set_plan(9, B, h, cin, cout, 24, wgs) != 0:
            continue
        try:
            res.append((timed(x, dy, wt), wgs))
        except RuntimeError as e:
            res.append((float('inf'), wgs))
    res.sort()
    flops = 2.0 * B * h * w_ * cin * cout * 9
    print(f'{cin:>3}->{cout:<3} {h}x{w_}: direct {plan} {t0:.1f} us ({flops / t0 / 1e6:.0f} TF)   winograd best wgs={res[0][1]} {res[0][0]:.1f} us '
          f'({flops / res[0][0] / 1e6:.0f} TF-equivalent)  x{t0 / res[0][0]:.2f}   ' + ' '.join(f'{w}:{t:.1f}' for t, w in res[1:]), flush=True)
    lib.rv_conv_wgrad_set_plan(9, B, h, cin, cout, 0, 0)
