// fp32 GEMM on the bf16 matrix pipe with THREE-WAY SPLIT operands ("bf16x6"): an OPT-IN experiment (never the
// headline -- the reference computes in fp32, model/UNet_onset.py:50-52,275,292-293,324 are plain nn.Linear calls).
//
//   a = a0 + a1 + a2,  b = b0 + b1 + b2   (each part a bf16: 3 x 8 = 24 mantissa bits, the split of an fp32 number is EXACT)
//   a * b ~= a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)        six products, fp32 accumulate; dropped: 2^-24 relative and below
//
// i.e. numerically "another fp32 summation order" (profiles/r05_bf16_split_emulation.txt: non-VAT terms bit-identical, VAT terms <= 8.6e-4 at
// B = 8 + 8 in emulation), but the products run on v_mfma_f32_16x16x32_bf16 -- 16 cycles for 16 k of a 16 x 16 tile where the exact
// v_mfma_f32_16x16x4_f32 takes 32 cycles for 4 k -- and vector-ALU work (the split) issues BESIDE bf16 MFMAs (tools/probes/mfma_valu_overlap.hip),
// where next to f32 MFMAs it adds.  Per 32 k of a 64 x 64 tile a wave issues 24 bf16 MFMAs (384 cycles) instead of 32 f32 MFMAs (1 024).
//
//   C[m][n] (+)= act( sum_k A[m][k] * B[n][k] + bias[n] )      A: [M][K] row stride lda, B: [N][K] row stride ldb (both k-contiguous: the
//                                                              forward X @ W^T and, through ops._lin_t, the input gradient dY @ W)
//
// 64 x 64 block tile, BK = 32, four waves of 32 x 32; operand tiles travel global -> registers (prefetched one tile ahead) -> split into three
// bf16 planes -> LDS (double-buffered: ONE barrier per k-tile) -> 16-byte fragment reads (row stride 80 B: the sixteen rows of a lane group
// cover all 64 banks once).  Same accumulator layout and epilogue conventions as gemm_mfma_k (gemm.hip).
#include "common.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef f32x4 f32x4u __attribute__((aligned(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define XBM 64
#define XBN 64
#define XBK 32
#define XROW 80                       // bytes per staged row of one plane: 32 bf16 + 16 B pad
#define XPLANE (64 * XROW)            // one plane of one operand tile
#define XOPER (3 * XPLANE)            // three planes
#define XBUF (2 * XOPER)              // A and B of one k-tile

struct Gemm6Args {
    const float* A; long lda;
    const float* B; long ldb;
    float* C; long ldc;
    const float* bias;
    int M, N, K;
    int act;          // 0 none, 1 sigmoid
    int accumulate;   // C += result
    int a_vec, b_vec; // rows may be read 16 bytes at a time (always true for fp32 rows on this target; kept for A/B runs)
};

// (hi, mid, lo) bf16 parts of two floats, packed two to a dword: x = hi + mid + lo exactly (round-to-nearest-even conversions)
__device__ __forceinline__ void split2(const float x0, const float x1, unsigned& h, unsigned& m, unsigned& l) {
    const bf16x2 hh = __builtin_convertvector((f32x2){x0, x1}, bf16x2);
    h = __builtin_bit_cast(unsigned, hh);
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    const bf16x2 mm = __builtin_convertvector((f32x2){r0, r1}, bf16x2);
    m = __builtin_bit_cast(unsigned, mm);
    const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    const bf16x2 ll = __builtin_convertvector((f32x2){s0, s1}, bf16x2);
    l = __builtin_bit_cast(unsigned, ll);
}

// one operand tile in flight: thread t holds elements (row = idx >> 3, k = (idx & 7) * 4 .. + 3), idx = t + 256 e
struct Tile6 {
    f32x4 r[2];
    __device__ __forceinline__ void load(const float* base, long ld, int row0, int nrows, int k0, int K, int tid) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = tid + e * 256, row = idx >> 3, k = (idx & 7) * 4;
            const int gr = row0 + row, gk = k0 + k;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (gr < nrows && gk < K) {
                const float* p = base + (long)gr * ld + gk;
                if (gk + 3 < K) v = *reinterpret_cast<const f32x4u*>(p);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (gk + j < K) v[j] = p[j];
                }
            }
            r[e] = v;
        }
    }
    template <int NP>
    __device__ __forceinline__ void split_store(char* oper, int tid) const {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = tid + e * 256, row = idx >> 3, k = (idx & 7) * 4;
            unsigned h0, m0, l0, h1, m1, l1;
            split2(r[e][0], r[e][1], h0, m0, l0);
            split2(r[e][2], r[e][3], h1, m1, l1);
            char* p = oper + row * XROW + k * 2;
            *reinterpret_cast<u32x2*>(p) = (u32x2){h0, h1};
            if constexpr (NP >= 2) *reinterpret_cast<u32x2*>(p + XPLANE) = (u32x2){m0, m1};
            if constexpr (NP >= 3) *reinterpret_cast<u32x2*>(p + 2 * XPLANE) = (u32x2){l0, l1};
        }
    }
};

// NP: bf16 planes kept per operand -- 3 (six products: the fp32-faithful form), 2 (three products, "bf16x3") or 1 (plain bf16); 2 and 1 exist for the
// cost ladder of tools/bench_bf16x6.py (what the kernel would cost if the extra products, their LDS traffic and their split were free)
template <int NP>
__global__ __launch_bounds__(256) void gemm_bf16x6_k(Gemm6Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem6[];          // [2 buffers][A | B][3 planes][64 rows][80 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * XBM, n0 = blockIdx.x * XBN;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x4 acc[2][2];   // [n-tile][m-tile]: a lane holds four consecutive n (4g .. 4g+3) of m = li
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // operand tiles in flight: tile t is multiplied out of LDS buffer t & 1 while tile t+1 (registers, set (t+1) & 1) is split and staged into the
    // other buffer and tiles t+2 / t+3 travel from global memory -- a prefetch distance of TWO k-tiles (with one, an iteration lasted exactly
    // one memory latency: 1.3 us per 32 k, four times the MFMA time)
    Tile6 ta[2], tb[2];
    const int ntile = (a.K + XBK - 1) / XBK;
    ta[0].load(a.A, a.lda, m0, a.M, 0, a.K, tid);
    tb[0].load(a.B, a.ldb, n0, a.N, 0, a.K, tid);
    if (ntile > 1) {
        ta[1].load(a.A, a.lda, m0, a.M, XBK, a.K, tid);
        tb[1].load(a.B, a.ldb, n0, a.N, XBK, a.K, tid);
    }
    ta[0].template split_store<NP>(smem6, tid);
    tb[0].template split_store<NP>(smem6 + XOPER, tid);
    if (ntile > 2) {
        ta[0].load(a.A, a.lda, m0, a.M, 2 * XBK, a.K, tid);
        tb[0].load(a.B, a.ldb, n0, a.N, 2 * XBK, a.K, tid);
    }
    __syncthreads();
    auto step = [&](auto par, const int t) {
        constexpr int P = decltype(par)::value;            // t & 1
        const char* As = smem6 + P * XBUF;
        const char* Bs = As + XOPER;
        // fragments of this k-tile: lane (li, g) holds k = 8g .. 8g+7 of row li -- 16 bytes per plane and tile
        bf16x8 af[2][NP], bf[2][NP];
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int p = 0; p < NP; ++p) af[y][p] = *reinterpret_cast<const bf16x8*>(As + p * XPLANE + (wm + y * 16 + li) * XROW + g * 16);
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int p = 0; p < NP; ++p) bf[x][p] = *reinterpret_cast<const bf16x8*>(Bs + p * XPLANE + (wn + x * 16 + li) * XROW + g * 16);
        if (t + 1 < ntile) {
            char* nx = smem6 + (P ^ 1) * XBUF;
            ta[P ^ 1].template split_store<NP>(nx, tid);
            tb[P ^ 1].template split_store<NP>(nx + XOPER, tid);
            if (t + 3 < ntile) {
                ta[P ^ 1].load(a.A, a.lda, m0, a.M, (t + 3) * XBK, a.K, tid);
                tb[P ^ 1].load(a.B, a.ldb, n0, a.N, (t + 3) * XBK, a.K, tid);
            }
        }
        // six products per tile pair, the small ones first (D^T = B^T A^T: A-operand = B tile, see gemm.hip)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                f32x4 c = acc[x][y];
                if constexpr (NP >= 3) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[x][2], af[y][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[x][1], af[y][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[x][0], af[y][2], c, 0, 0, 0);
                }
                if constexpr (NP >= 2) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[x][1], af[y][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[x][0], af[y][1], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[x][0], af[y][0], c, 0, 0, 0);
                acc[x][y] = c;
            }
        __syncthreads();                // the other buffer is complete, this one is free for tile t+2
    };
    for (int t = 0; t < ntile; t += 2) {
        step(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntile) step(std::integral_constant<int, 1>{}, t + 1);
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int m = m0 + wm + y * 16 + li;
            const int nb = n0 + wn + x * 16 + 4 * g;
            if (m >= a.M || nb >= a.N) continue;
            f32x4 v = acc[x][y];
            if (a.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nb + r < a.N) v[r] += a.bias[nb + r];
            }
            if (a.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = 1.f / (1.f + __expf(-v[r]));
            }
            float* c = a.C + (long)m * a.ldc + nb;
            if (nb + 3 < a.N) {
                if (a.accumulate) { const f32x4 o = *reinterpret_cast<f32x4u*>(c); v += o; }
                *reinterpret_cast<f32x4u*>(c) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nb + r < a.N) c[r] = a.accumulate ? c[r] + v[r] : v[r];
            }
        }
}

extern "C" {

// Experiment entry point (NOT part of include/reconvat_hip.h): C[M][N] (+)= act(A[M][K] . B[N][K]^T + bias) with three-way split bf16 operands.
int rv_debug_gemm_bf16x6(const float* A, long lda, const float* B, long ldb, float* C, long ldc, const float* bias, int M, int N, int K, int act,
                         int accumulate, void* stream) {
    RV_CHECK_ARG(M > 0 && N > 0 && K > 0, "rv_debug_gemm_bf16x6: empty problem");
    Gemm6Args a;
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.bias = bias; a.M = M; a.N = N; a.K = K; a.act = act;
    a.accumulate = accumulate; a.a_vec = 1; a.b_vec = 1;
    const int planes = (act >> 8) & 3;            // (cost ladder: act | 1 << 8 = one plane, 2 << 8 = two; default three)
    a.act = act & 0xff;
    const dim3 grid(cdiv(N, XBN), cdiv(M, XBM));
    if (planes == 1) hipLaunchKernelGGL(gemm_bf16x6_k<1>, grid, dim3(256), 2 * XBUF, (hipStream_t)stream, a);
    else if (planes == 2) hipLaunchKernelGGL(gemm_bf16x6_k<2>, grid, dim3(256), 2 * XBUF, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(gemm_bf16x6_k<3>, grid, dim3(256), 2 * XBUF, (hipStream_t)stream, a);
    RV_LAUNCH_CHECK("rv_debug_gemm_bf16x6");
    return RV_OK;
}

}  // extern "C"
