// Probe: HOW does v_mfma_f32_16x16x4_f32 round?  D[i][j] = C[i][j] + sum_{k<4} A[i][k] * B[k][j] -- as a chain of fused multiply-adds in k order
// (d = fma(a0, b0, c); d = fma(a1, b1, d); ...), in another order, as a pairwise tree, or with unfused products?  Decides whether a vector-ALU fmaf chain
// can reproduce a 1x1 convolution of the MFMA kernels BIT FOR BIT (round 6: dense skip conv evaluated inside the BatchNorm apply kernel).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f32_order.hip -o gpurun_out/mfma_f32_order && gpurun_out/mfma_f32_order
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, const float* C, float* D) {     // A[16][4], B[4][16], C/D[16][16]; one wave
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    // A operand: lane (i, g) holds A[i][k = g]; B operand: lane (j = i, g) holds B[k = g][j]; D: lane holds D[4g + r][i]
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * g + r) * 16 + i];
    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 4 + g], B[g * 16 + i], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = d[r];
}
static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
int main() {
    const int trials = 4000;
    std::vector<float> A(64), B(64), C(256), D(256);
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    auto val = [&]() { const double m = rnd() * 2 - 1; const int e = (int)(rnd() * 24) - 12; return (float)ldexp(m, e); };      // wide exponent range: cancellation happens
    long mism[6] = {0, 0, 0, 0, 0, 0};
    const char* names[6] = {"fma chain k = 0,1,2,3 from C", "fma chain k = 3,2,1,0 from C", "products summed first (fma chain from 0), then + C", "pairwise tree of fused products", "unfused: round each product, add in k order", "exact sum rounded once (fp64)"};
    for (int t = 0; t < trials; ++t) {
        for (auto& x : A) x = val();
        for (auto& x : B) x = val();
        for (auto& x : C) x = (t & 1) ? val() : 0.f;
        hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                const float c = C[i * 16 + j], a0 = A[i * 4], a1 = A[i * 4 + 1], a2 = A[i * 4 + 2], a3 = A[i * 4 + 3];
                const float b0 = B[j], b1 = B[16 + j], b2 = B[32 + j], b3 = B[48 + j];
                float h[6];
                h[0] = fmaf(a3, b3, fmaf(a2, b2, fmaf(a1, b1, fmaf(a0, b0, c))));
                h[1] = fmaf(a0, b0, fmaf(a1, b1, fmaf(a2, b2, fmaf(a3, b3, c))));
                h[2] = fmaf(a3, b3, fmaf(a2, b2, fmaf(a1, b1, a0 * b0))) + c;
                h[3] = (fmaf(a1, b1, a0 * b0) + fmaf(a3, b3, a2 * b2)) + c;
                h[4] = (((c + a0 * b0) + a1 * b1) + a2 * b2) + a3 * b3;
                h[5] = (float)((double)c + (double)a0 * b0 + (double)a1 * b1 + (double)a2 * b2 + (double)a3 * b3);
                for (int q = 0; q < 6; ++q) mism[q] += bits(h[q]) != bits(D[i * 16 + j]);
            }
    }
    const long total = (long)trials * 256;
    for (int q = 0; q < 6; ++q) printf("%-55s: %ld of %ld results differ\n", names[q], mism[q], total);
    printf("v_mfma_f32_16x16x4_f32 %s a chain of fused multiply-adds in k order starting from C\n", mism[0] == 0 ? "IS" : "is NOT");
    return 0;
}
