"""bf16x6 GEMM prototype (reconvat_amd/csrc/gemm_bf16x6.hip, rv_debug_gemm_bf16x6) against the shipped exact-f32 kernel (rv_gemm) on the
k-contiguous GEMM shapes of the training step: microseconds per launch (hipGraph of 20 launches, rotating operand sets) and the error of
both against an fp64 product.   python tools/bench_bf16x6.py > profiles/r06_bf16x6_gemm.txt"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconvat_amd import ops, _lib

dev = torch.device('cuda:0')
lib = ctypes.CDLL(_lib.LIB_PATH)
fn = lib.rv_debug_gemm_bf16x6
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
# (M, N, K) of the forward / input-gradient linear GEMMs of the step (reconvat_amd/tuned_plans.json, both operands k-contiguous), B_l = B_ul = 8
SHAPES = [(5120, 1536, 768), (5120, 768, 1536), (5120, 2304, 176), (5120, 176, 2304), (5120, 1536, 176), (5120, 2748, 229), (5120, 229, 2748), (5120, 916, 229),
          (5120, 229, 916), (5120, 2748, 88), (5120, 88, 2748), (5120, 916, 88), (5120, 88, 916), (5120, 768, 88), (5120, 88, 768), (5120, 229, 88), (5120, 88, 229)]
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


print('# M x N x K: exact f32 kernel (rv_gemm, tuned split-K) vs bf16x6 prototype; errors = relative L2 against an fp64 product')
tot32 = tot6 = 0.0
for m, n, k in SHAPES:
    torch.manual_seed(m + n + k)
    A = [torch.randn(m, k, device=dev) for _ in range(NSET)]
    W = [torch.randn(n, k, device=dev) * (1.0 / k ** 0.5) for _ in range(NSET)]
    bias = torch.randn(n, device=dev)
    C32 = [torch.empty(m, n, device=dev) for _ in range(NSET)]
    C6 = [torch.empty(m, n, device=dev) for _ in range(NSET)]

    def f32(i):
        ops.gemm(A[i], W[i].t(), C32[i], bias=bias)

    def b6(i):
        rc = fn(A[i].data_ptr(), k, W[i].data_ptr(), k, C6[i].data_ptr(), n, bias.data_ptr(), m, n, k, 0, 0, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    def ladder(planes):
        def run(i):
            rc = fn(A[i].data_ptr(), k, W[i].data_ptr(), k, C6[i].data_ptr(), n, bias.data_ptr(), m, n, k, planes << 8, 0, torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        return run
    t1, t3 = timeit(ladder(1)), timeit(ladder(2))
    t32, t6 = timeit(f32), timeit(b6)
    ref = (A[0].double().cpu() @ W[0].double().cpu().t()) + bias.double().cpu()
    e32 = float((C32[0].double().cpu() - ref).norm() / ref.norm())
    e6 = float((C6[0].double().cpu() - ref).norm() / ref.norm())
    gf = 2.0 * m * n * k / 1e9
    tot32 += t32; tot6 += t6
    print(f'{m:5d} x {n:5d} x {k:5d}  {gf:6.2f} GFLOP   f32 {t32:7.1f} us {gf / t32 * 1e3:6.1f} TF/s  err {e32:.2e}   |  bf16x6 {t6:7.1f} us {gf / t6 * 1e3:6.1f} TF/s  err {e6:.2e}   '
          f'x{t32 / t6:.2f}   | same kernel with 3 products {t3:6.1f} us, 1 product (plain bf16) {t1:6.1f} us')
print(f'# sum over the {len(SHAPES)} shapes: f32 {tot32:.0f} us, bf16x6 {tot6:.0f} us (x{tot32 / tot6:.2f})')
