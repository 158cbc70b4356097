// Shared device/host helpers for libreconvat_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define RV_OK 0
#define RV_EINVAL (-1)
#define RV_ELAUNCH (-2)
#define RV_EUNSUPPORTED (-3)

extern "C" void rv_set_error(const char* fmt, ...);

#define RV_CHECK_ARG(cond, ...)                        \
    do {                                               \
        if (!(cond)) {                                 \
            rv_set_error(__VA_ARGS__);                 \
            return RV_EINVAL;                          \
        }                                              \
    } while (0)

#define RV_LAUNCH_CHECK(name)                                                      \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            rv_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
            return RV_ELAUNCH;                                                     \
        }                                                                          \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- wave64 reductions -------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware block remap (8 XCDs, dispatcher places block b on XCD b%8): give every XCD a
// contiguous chunk of the logical grid so neighbouring tiles share an L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// Division by a launch-constant via multiply-high (n < 2^31, d >= 1): q = (umulhi(n, mul) + n) >> shift.
struct FastDiv {
    unsigned mul, shift, d;
};
static inline FastDiv fastdiv_make(unsigned d) {
    FastDiv f;
    f.d = d;
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    f.shift = s;
    f.mul = (unsigned)((((1ull << s) - d) << 32) / d + 1);
    return f;
}
__device__ __forceinline__ unsigned fastdiv(unsigned n, const FastDiv& f) {
    return (unsigned)(((unsigned long long)__umulhi(n, f.mul) + n) >> f.shift);
}

// LDS-DMA (global -> LDS without a VGPR round trip, tracked by vmcnt): every active lane moves 16 (or 4) bytes to
// LDS address = wave-uniform base + lane * 16 (or 4).  Wait with s_waitcnt vmcnt before a barrier publishes the data.
__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

// BatchNorm reduction workspace: RV_BN_NREP replicas of [2C] fp64 sums.  A producer workgroup adds into replica
// (blockIdx.x % RV_BN_NREP); consumers add the replicas up.  Same-address fp64 atomics serialise at ~23 ns each at the
// memory side, so spreading the producers over 8 copies cuts the atomic tail of every producer kernel by 8.
#define RV_BN_NREP 8
__device__ __forceinline__ double bn_sum_replicas(const double* sums, int C, int idx) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < RV_BN_NREP; ++r) s += sums[r * 2 * C + idx];
    return s;
}

// bn.hip: (replicated, see above) sums[0..C) += sum_p z[p][c], sums[C..2C) += sum_p z[p][c]^2 (fp64) -- the BatchNorm2d batch statistics pass,
// also used by rv_conv_fwd behind the conv kernels that do not produce the statistics in their epilogue.
int rv_internal_bn_stats(const float* z, int z_ld, long P, int C, double* sums, hipStream_t st);
// ... and the backward reduction: sums += (sum dd, sum dd * xhat), dd = dy * lrelu'(z*scale+shift); coef = [mean|invstd|scale|shift|..]
int rv_internal_bn_bwd_stats(const float* dy, int dy_ld, const float* z, int z_ld, long P, int C, const float* coef, float slope,
                             double* sums, hipStream_t st);
