"""Time every tile of every 3x3 kernel family on the step's main shapes; print the best of each family per shape.

    python tools/sweep_conv_tiles.py            # families: 1 direct, 2/3/4/7 LDS kernel with 4/8/16/12 waves
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconvat_amd import ops, _lib

SHAPES = [(16, 16, 640, 229), (48, 24, 320, 114), (32, 32, 320, 114), (64, 64, 160, 57), (96, 48, 160, 57), (128, 128, 80, 28),
          (192, 96, 80, 28), (256, 256, 40, 14), (128, 64, 80, 28), (64, 32, 160, 57)]
B = 8
TH = range(0, 41) if '--th' in sys.argv else (0,)       # --th: also sweep the rows per band
dev = torch.device('cuda:0')
lib = _lib.load()
st = torch.cuda.current_stream()
for cin, cout, h, w in SHAPES:
    x = torch.rand(B, h, w, cin, device=dev) - 0.5
    wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
    bias = torch.zeros(cout, device=dev)
    y = torch.empty(B, h, w, cout, device=dev)
    stats = torch.zeros(ops.bn_ws_doubles(cout), dtype=torch.float64, device=dev)
    wp = ops._pack('c3', wt, 'fwd')
    args = (0, ops.ptr(x), cin, B, h, w, cin, ops.ptr(y), cout, h, w, cout, ops.ptr(wp), ops.ptr(bias), 0)
    tail = (None, 0, None, 0.0)
    flops = 2.0 * B * h * w * cin * cout * 9
    best = {}
    ntile_n = (cout + 15) // 16
    for fam in (1, 2, 3, 4, 7):
        for nt in (1, 2, 3, 4):
            if ntile_n % nt:
                continue
            for mt in (1, 2, 3, 4, 5, 6, 8):
              for th in (TH if fam > 1 else (0,)):
                cand = th << 12 | fam << 8 | nt << 4 | mt
                if lib.rv_conv_fwd(*args, cand, ops.ptr(stats), *tail, st.cuda_stream) != 0:
                    continue
                t = None
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    for _ in range(5):
                        lib.rv_conv_fwd(*args, cand, ops.ptr(stats), *tail, st.cuda_stream)
                    e1.record(st)
                    e1.synchronize()
                    dt = e0.elapsed_time(e1) / 5
                    t = dt if t is None else min(t, dt)
                if fam not in best or t < best[fam][0]:
                    best[fam] = (t, cand)
    print(f'{cin:4d}->{cout:<4d} {h}x{w}: ' + '  '.join(f'{hex(c)} {t * 1e3:6.1f}us {flops / t / 1e9:5.1f}TF' for f, (t, c) in sorted(best.items())))
