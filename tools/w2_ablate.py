"""Timing ablations of the software-pipelined Winograd kernel (conv_wino2.hip, -DRV_W2_DEV build loaded through RECONVAT_HIP_LIB):
every mask of RV_W2_ABL on a few BASELINE shapes (B = 8, fused statistics).  Results are WRONG by design for mask != 0.

    RECONVAT_HIP_LIB=reconvat_amd/libreconvat_hip_dev.so python tools/w2_ablate.py
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconvat_amd import ops, _lib

SHAPES = [(64, 64, 160, 57), (96, 48, 160, 57), (16, 16, 640, 229), (32, 32, 320, 114), (128, 128, 80, 28)]
MASKS = [(0, 'full'), (64, '- statistics tail'), (96, '- band epilogue'), (120, '- barrier, staging'), (122, '- patch reads / transform'),
         (126, '- weight reads (MFMAs only)'), (124, 'MFMAs + patch side work only'), (1, 'everything but the MFMAs'), (2, 'no patch side work'),
         (24, 'no barrier, no staging'), (26, 'no barrier / staging / patch side work'), (30, 'no barrier / staging / patch / weight reads'),
         (32, 'no band epilogue')]
if os.environ.get('W2_MASKS'):
    MASKS = [(int(m), '') for m in os.environ['W2_MASKS'].split(',')]
ALGO = int(os.environ.get('W2_ALGO', '0x811'), 0)
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
    flops = 2.0 * B * h * w * cin * cout * 9
    os.environ.pop('RV_W2_ABL', None)
    told = timed(args, 0x611, stats)
    print(f'{cin}->{cout} {h}x{w}: round-4 kernel 0x611 {told:.1f} us; MFMA-only bound at 157.3 TF: {flops / 2.25 / 157.3e6:.1f} us')
    for mask, name in MASKS:
        os.environ['RV_W2_ABL'] = str(mask)
        t = timed(args, ALGO, stats)
        print(f'    mask {mask:>3} {name:<34} {t:7.1f} us', flush=True)
    os.environ.pop('RV_W2_ABL', None)
