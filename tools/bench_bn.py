"""Streaming-kernel bandwidth check: BatchNorm apply / reduce vs plain device copies (B=8, 640x229)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconvat_amd import ops

dev = torch.device('cuda:0')


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for c, h, w in ((16, 640, 229), (32, 320, 114), (64, 160, 57), (128, 80, 28)):
    z = torch.rand(8, h, w, c, device=dev)
    y = torch.empty_like(z)
    mb = z.numel() * 4 / 1e6
    t = timeit(lambda: y.copy_(z))
    print(f'C={c:3d} {h}x{w}  {mb:6.1f} MB  copy {t:6.1f} us {2 * mb / t:5.2f} TB/s', end='')
    g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    rm, rv, nbt = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.zeros((), dtype=torch.long, device=dev)
    zz = z.clone().requires_grad_(True)
    t = timeit(lambda: ops.BnActFn.apply(z, g, b, rm, rv, nbt, None, True, 0.01))
    print(f' | bn fwd (reduce+apply, 2 reads 1 write) {t:6.1f} us {3 * mb / t:5.2f} TB/s', end='')
    yy = ops.BnActFn.apply(zz, g, b, rm, rv, nbt, None, True, 0.01)
    dy = torch.rand_like(yy)
    t = timeit(lambda: torch.autograd.grad(yy, zz, dy, retain_graph=True))
    print(f' | bn bwd (4 reads 1 write) {t:6.1f} us {5 * mb / t:5.2f} TB/s')
