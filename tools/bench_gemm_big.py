"""Block-tile policy of rv_gemm on the k-contiguous, unsplit GEMM shapes of the step: 64 x 64 (gemm_mfma_k) vs 128 x 64 / 128 x 128 (gemm_big_k) --
microseconds per launch (hipGraph of 20 launches on rotating operands) and BIT-IDENTITY of the results (same MFMA sequence per output element).
    python tools/bench_gemm_big.py > profiles/r06_gemm_big_tiles.txt"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconvat_amd import ops, _lib

dev = torch.device('cuda:0')
lib = ctypes.CDLL(_lib.LIB_PATH)
setmode = lib.rv_debug_set_gemm_big
SHAPES = [(5120, 2304, 176), (5120, 2748, 229), (5120, 2748, 88), (5120, 916, 229), (5120, 916, 88), (5120, 768, 88), (5120, 229, 88), (5120, 1536, 768), (5120, 1536, 176),
          (5120, 768, 1536), (5120, 229, 916), (5120, 88, 768), (5120, 176, 2304), (640, 2304, 176), (640, 916, 229)]
NSET, REPS = 4, 20


def timeit(launch):
    for i in range(3):
        launch(i % NSET)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(REPS):
            launch(i % NSET)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / REPS)
    return sorted(ts)[2]


print('# M x N x K (unsplit, both operands k-contiguous, bias + sigmoid epilogue): us per launch by block tile; "auto" = the size-based policy (RV_GEMM_BIG=1; the shipped default is 64 x 64 everywhere); results compared bit for bit')
for m, n, k in SHAPES:
    torch.manual_seed(m + n + k)
    A = [torch.randn(m, k, device=dev) for _ in range(NSET)]
    W = [torch.randn(n, k, device=dev) * (1.0 / k ** 0.5) for _ in range(NSET)]
    bias = torch.randn(n, device=dev)
    out, res = {}, {}
    for name, mode in (('64x64', 0), ('128x64', 3), ('128x128', 2), ('auto', 1)):
        setmode(mode)
        C = [torch.empty(m, n, device=dev) for _ in range(NSET)]

        def run(i):
            ops.gemm(A[i], W[i].t(), C[i], bias=bias, act=1, splitk=1)
        out[name] = timeit(run)
        res[name] = C[0].clone()
    setmode(0)
    same = all(torch.equal(res['64x64'], res[nm]) for nm in ('128x64', '128x128', 'auto'))
    gf = 2.0 * m * n * k / 1e9
    best = min(out, key=out.get)
    print(f'{m:5d} x {n:5d} x {k:5d} {gf:6.2f} GF  ' + '  '.join(f'{nm} {out[nm]:6.1f}' for nm in ('64x64', '128x64', '128x128', 'auto')) +
          f'   best {best:8s} auto/64x64 x{out["64x64"] / out["auto"]:.2f}   bit-identical: {same}')
