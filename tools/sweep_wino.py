"""Time the Winograd F(2x2,3x3) tiles (families 0x6NM / 0xANM / 0xCNM, every legal rows-per-band) against the plan-table tile of the direct
persistent kernel on the 3x3 launch shapes of the BASELINE step (B = 8, fused statistics).

    python tools/sweep_wino.py            (one line per shape)
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconvat_amd import ops, _lib, plans

SHAPES = [(16, 16, 640, 229), (16, 32, 320, 114), (32, 32, 320, 114), (32, 64, 160, 57), (64, 64, 160, 57), (64, 128, 80, 28),
          (128, 128, 80, 28), (192, 96, 80, 28), (96, 64, 80, 28), (96, 48, 160, 57), (48, 32, 160, 57), (48, 24, 320, 114),
          (16, 8, 640, 229), (64, 64, 80, 28), (32, 32, 160, 57), (16, 16, 320, 114), (128, 64, 80, 28), (64, 32, 160, 57), (32, 16, 320, 114),
          (48, 96, 160, 57), (96, 192, 80, 28)]
B = 8
dev = torch.device('cuda:0')
lib = _lib.load()
st = torch.cuda.current_stream()


def timed(args, algo, stats):
    if lib.rv_conv_fwd(*args, algo, ops.ptr(stats), None, 0, None, 0.0, st.cuda_stream) != 0:
        return None
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            lib.rv_conv_fwd(*args, algo, ops.ptr(stats), None, 0, None, 0.0, st.cuda_stream)
        e1.record(st)
        e1.synchronize()
        t = e0.elapsed_time(e1) / 5 * 1e3
        best = t if best is None else min(best, t)
    return best


for cin, cout, h, w in SHAPES:
    x = torch.rand(B, h, w, cin, device=dev) - 0.5
    wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
    bias = torch.zeros(cout, device=dev)
    y = torch.empty(B, h, w, cout, device=dev)
    stats = torch.zeros(ops.bn_ws_doubles(cout), dtype=torch.float64, device=dev)
    wp = ops._pack('c3', wt, 'fwd')
    args = (0, ops.ptr(x), cin, B, h, w, cin, ops.ptr(y), cout, h, w, cout, ops.ptr(wp), ops.ptr(bias), 0)
    key = (0, B, h, w, cin, cout, cin, cout, True, False)
    hit = plans.lookup_conv(key)
    base_algo = hit[0] if hit else 0
    t0 = timed(args, base_algo, stats)
    res = []
    ntile_n = (cout + 15) // 16
    only_new = os.environ.get('SWEEP_ONLY_NEW') == '1'
    for fam, nw in ((6, 8), (10, 8), (12, 12), (8, 8), (9, 8), (11, 4), (13, 12)):
        if only_new and fam in (6, 10, 12):
            continue
        for nt, mt in ((1, 1), (2, 1), (1, 2), (2, 2), (1, 4)):
            legal = {6: ((1, 1),), 10: ((1, 1), (2, 1)), 12: ((1, 1),), 8: ((1, 1),), 9: ((1, 1), (2, 1), (1, 2)),
                     11: ((1, 2), (2, 1)), 13: ((1, 1),)}[fam]
            if ntile_n % nt or (nt, mt) not in legal:
                continue
            wt_ = (w + 1) // 2
            th_max = min(h, 2 * ((nw * mt * 16) // wt_))
            ths = [0] + [t for t in range(2, th_max, 2) if -(-h // t) != -(-h // (t + 2))]
            if fam in (8, 9, 11, 13) and os.environ.get('SWEEP_ALL_TH') != '1':
                ths = [0] + [t for t in ths[1:] if t >= th_max // 2]
            for th in ths:
                algo = th << 12 | fam << 8 | nt << 4 | mt
                t = timed(args, algo, stats)
                if t is not None:
                    res.append((t, algo))
    res.sort()
    flops = 2.0 * B * h * w * cin * cout * 9
    best = res[0] if res else (None, 0)
    print(f'{cin:>3}->{cout:<3} {h}x{w}: table {base_algo:#x} {t0:.1f} us ({flops / t0 / 1e6:.0f} TF)   winograd best {best[1]:#x} '
          f'{best[0]:.1f} us ({flops / best[0] / 1e6:.0f} TF-equivalent)  x{t0 / best[0]:.2f}   next: '
          + ' '.join(f'{a:#x}:{t:.1f}' for t, a in res[1:4]), flush=True)
