"""Raw f32 MFMA issue rate of the chip (rv_debug_mfma_peak: NACC independent accumulators per wave, no memory traffic): the 146-156 TFLOP/s the
conv kernels are graded against."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reconvat_amd import _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
fn = lib.rv_debug_mfma_peak
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.empty(4096 * 256, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for blocks in (256, 512, 1024, 2048):
    for nacc in (2, 4, 8):
        iters = 2000
        fn(out.data_ptr(), blocks, 10, nacc, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(out.data_ptr(), blocks, iters, nacc, st)
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1)
        fl = blocks * 4 * iters * 4 * nacc * 2048.0
        print(f'blocks={blocks} (waves/SIMD={blocks*4/1024:.1f}) nacc={nacc}: {ms*1e3:.0f} us  {fl/ms/1e9:.1f} TFLOP/s')
