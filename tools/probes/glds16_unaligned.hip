// Probe: does global_load_lds_dwordx4 accept a global source address that is only 4-byte aligned?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/glds16_unaligned.hip -o tools/probes/glds16_unaligned && tools/probes/glds16_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* in, float* out, int shift) {
    __shared__ __attribute__((aligned(16))) float smem[256];
    const float* src = in + shift + threadIdx.x * 4;          // 4-byte aligned only when shift % 4 != 0
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)smem, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int q = 0; q < 4; ++q) out[threadIdx.x * 4 + q] = smem[threadIdx.x * 4 + q];
}
int main() {
    const int n = 1024;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1.f + i;
    float *d, *o;
    (void)hipMalloc(&d, n * 4); (void)hipMalloc(&o, 256 * 4);
    (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    int rc = 0;
    for (int shift = 0; shift < 4; ++shift) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, shift);
        std::vector<float> r(256);
        (void)hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) if (r[i] != 1.f + shift + i) { if (bad < 4) printf("shift %d elem %d: got %g want %g\n", shift, i, r[i], 1.f + shift + i); ++bad; }
        printf("global_load_lds_dwordx4, source shifted by %d floats: %s\n", shift, bad ? "WRONG" : "ok");
        rc |= bad != 0;
    }
    return rc;
}
