"""Split-K sweep for the weight-gradient GEMMs of the linear heads (A = dY^T, K = B*T rows): fp32-atomic accumulation against the
deterministic ticketed in-order fold (rv_gemm with / without the partial-tile workspace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconvat_amd import ops
dev = torch.device('cuda:0')
for (m, n, k) in ((88, 229, 5120), (88, 768, 5120), (2304, 176, 5120), (768, 88, 5120), (229, 916, 5120)):
    dz = torch.randn(k, m, device=dev)
    x = torch.randn(k, n, device=dev)
    g = torch.zeros(m, n, device=dev)
    for det in (False, True):
        line = f'M,N,K=({m},{n},{k}) {"ticketed in-order fold" if det else "fp32 atomics          "}:'
        ops.ARENA.begin_step(dev, 64 << 20)
        for sk in (1, 2, 4, 8, 16, 32):
            for _ in range(3):
                ops.gemm(dz.t(), x, g, accumulate=True, splitk=sk, deterministic=det)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm(dz.t(), x, g, accumulate=True, splitk=sk, deterministic=det)
            e1.record(); e1.synchronize()
            line += f'  sk{sk}={e0.elapsed_time(e1) / 20 * 1e3:.1f}us'
        ops.ARENA.end_step()
        print(line)
