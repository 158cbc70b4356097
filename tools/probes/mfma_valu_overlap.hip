// Probe: do vector-ALU instructions execute in the shadow of matrix instructions of the SAME SIMD?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_valu_overlap.hip -o gpurun_out/mfma_valu_overlap && gpurun_out/mfma_valu_overlap
// One kernel per (matrix instruction, NV): a loop of [1 MFMA, NV independent VALU instructions] groups (order pinned with
// sched_barrier), 4 independent accumulators, W waves per SIMD.  If the VALU work hides behind the matrix pipe, the time per group
// stays at the MFMA's own issue interval until the VALU work exceeds it; if both run on the same lanes, it grows from NV = 1 on.
// Question behind it (round 5): the Winograd conv kernels issue 3.6 VALU instructions per v_mfma_f32_16x16x4_f32 -- are those free?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NV, int PK>
__global__ __launch_bounds__(256) void probe_k(float* out, int iters, float seed) {
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float a0 = seed + threadIdx.x, b0 = seed * 0.5f + threadIdx.x;
    bf16x8 ah, bh;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(a0 + i); bh[i] = (__bf16)(b0 - i); }
    f32x2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (f32x2){a0 + i, b0 - i};
    const f32x2 c = (f32x2){seed, -seed};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[u & 3], 0, 0, 0);
            else acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[u & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int j = (u * NV + k) & 7;
                if (PK) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c));
                else asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j][0]) : "v"(seed));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    float t = s[0] + s[1] + s[2] + s[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) t += v[i][0] + v[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int KIND, int NV, int PK>
static double run(float* out, int waves_per_simd, int iters) {
    const int blocks = 256 * waves_per_simd;              // 4 waves per block = one per SIMD and block
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe_k<KIND, NV, PK>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best * 1e6 / ((double)iters * 8 * waves_per_simd);        // ns per [MFMA + NV VALU] group and SIMD
}

#define ROW(KIND, PK, name)                                                                                         \
    for (int w = 1; w <= 2; ++w) {                                                                                   \
        const double t[] = {run<KIND, 0, PK>(out, w, iters), run<KIND, 1, PK>(out, w, iters), run<KIND, 2, PK>(out, w, iters), \
                            run<KIND, 4, PK>(out, w, iters), run<KIND, 8, PK>(out, w, iters)};                      \
        printf("%-34s %d wave(s)/SIMD: ns per group at NV = 0/1/2/4/8: %6.2f %6.2f %6.2f %6.2f %6.2f   slope %.2f ns per VALU\n", name, w, \
               t[0], t[1], t[2], t[3], t[4], (t[4] - t[0]) / 8);                                                    \
    }

int main() {
    float* out;
    hipMalloc(&out, 256 * 2 * 256 * sizeof(float));
    const int iters = 4000;
    ROW(0, 0, "v_mfma_f32_16x16x4_f32 + v_add_f32")
    ROW(0, 1, "v_mfma_f32_16x16x4_f32 + v_pk_add_f32")
    ROW(1, 0, "v_mfma_f32_16x16x32_bf16 + v_add_f32")
    ROW(1, 1, "v_mfma_f32_16x16x32_bf16 + v_pk_add_f32")
    // verdict line for the test: the f32 case
    const double f0 = run<0, 0, 0>(out, 1, iters), f4 = run<0, 4, 0>(out, 1, iters);
    const double h0 = run<1, 0, 0>(out, 1, iters), h4 = run<1, 4, 0>(out, 1, iters);
    printf("f32 MFMA: 4 VALU per MFMA cost %+.0f %% ; bf16 MFMA: %+.0f %%\n", (f4 / f0 - 1) * 100, (h4 / h0 - 1) * 100);
    return 0;
}
