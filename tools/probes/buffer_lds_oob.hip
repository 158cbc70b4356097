// Probe: does `buffer_load_dwordx4 ... offen lds` write ZEROS for lanes whose offset is outside the buffer resource?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/buffer_lds_oob.hip -o gpurun_out/buffer_lds_oob && gpurun_out/buffer_lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* in, float* out, unsigned bytes) {
    __shared__ __attribute__((aligned(16))) float smem[256];
    smem[threadIdx.x * 4 + 0] = -7.f; smem[threadIdx.x * 4 + 1] = -7.f; smem[threadIdx.x * 4 + 2] = -7.f; smem[threadIdx.x * 4 + 3] = -7.f;
    __syncthreads();
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, bytes, 0x00020000);
    const unsigned lane = threadIdx.x;
    // lanes 0..15 in range; 16..31 offset >= bytes by a little; 32..47 offset 0x40000000 + x; 48..63 offset 0x80000000 + x
    unsigned voff = lane * 16u;
    if (lane >= 16 && lane < 32) voff = bytes + (lane - 16) * 16u;
    if (lane >= 32 && lane < 48) voff = 0x40000000u + lane * 16u;
    if (lane >= 48) voff = 0x80000000u + lane * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int q = 0; q < 4; ++q) out[threadIdx.x * 4 + q] = smem[threadIdx.x * 4 + q];
}
int main() {
    const int n = 1024;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1.f + i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 256 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    const unsigned bytes = 16 * 16;            // only the first 16 quads are inside the resource
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, bytes);
    std::vector<float> r(256);
    hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int q = 0; q < 4; ++q) {
            const float want = l < 16 ? 1.f + l * 4 + q : 0.f;
            if (r[l * 4 + q] != want) { if (bad < 8) printf("lane %d elem %d: got %g want %g\n", l, q, r[l * 4 + q], want); ++bad; }
        }
    printf("buffer_load ... lds out-of-range lanes: %s (%d mismatches)\n", bad ? "NOT zero-filled" : "zero-filled", bad);
    return bad != 0;
}
