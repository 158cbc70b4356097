"""Micro-benchmark the local-attention kernels alone (B=8, L=640): python tools/bench_attn.py [F G Fin]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconvat_amd import ops

f, g, fin = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (768, 6, 176)
dev = torch.device('cuda:0')
torch.manual_seed(0)
x = (torch.rand(8, 640, fin, device=dev) - 0.5).requires_grad_(True)
w = [((torch.rand(f, fin, device=dev) - 0.5) * 0.1).requires_grad_(True) for _ in range(3)]
rel = torch.randn(1, f, 31, device=dev, requires_grad=True)
for _ in range(3):
    out, att = ops.LocalAttnFn.apply(x, w[0], w[1], w[2], rel, g)
    out.sum().backward()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    out, att = ops.LocalAttnFn.apply(x, w[0], w[1], w[2], rel, g)
    out.sum().backward()
e1.record()
e1.synchronize()
print(f'attention F={f} G={g}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per fwd+bwd (incl. projections)')
