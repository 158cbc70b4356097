// Generic-stride fp32 GEMM on v_mfma_f32_16x16x4_f32 for the linear heads and the attention
// projections (reference nn.Linear call sites: model/UNet_onset.py:50-52,62-64,275,292-293,324).
//
//   C[m][n] (+)= act( sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] + bias[n] )
//
// One kernel covers forward (X @ W^T), input-gradient (dY @ W) and weight-gradient (dY^T @ X, split
// along the reduction dimension) through the strides.  64x64 block tile, BK = 32, four waves each
// owning a 32x32 sub-tile.  Operand tiles are fetched with 16-byte loads along whichever dimension is
// contiguous in memory (scalar fallback for unaligned rows such as K = 229), kept in registers while
// the previous tile is multiplied (software prefetch) and written to LDS in the matching orientation.
// The MFMA is issued as D^T = B^T A^T so that an accumulator lane holds four consecutive n of one m
// (16-byte stores along the contiguous dimension of C).
#include "common.h"
#include <stdlib.h>
#include <mutex>
#include <atomic>

#define GBM 64
#define GBN 64
#define GBK 32
#define LDK (GBK + 4)    // row stride of a [rows][k] tile
#define LDM (GBM + 4)    // row stride of a [k][rows] tile

// 16-byte accesses to rows that are only 4-byte aligned (row strides 229, 317, ...): legal on this target (unaligned access mode is on for
// global memory -- hipcc itself emits global_load_dwordx4 for such a type) and a quarter of the instructions of the scalar path
typedef f32x4 f32x4u __attribute__((aligned(4)));

struct GemmArgs {
    const float* A; long sam, sak;
    const float* B; long sbk, sbn;
    float* C; long scm, scn;
    float* C2; long sc2m, sc2n;       // optional second destination (same values)
    const float* bias;
    int M, N, K;
    int act;            // 0 none, 1 sigmoid
    int accumulate;     // C += (plain read-modify-write; requires splitk == 1)
    int splitk;         // >1: K split over blockIdx.z; with `part` the partial tiles are parked there and gemm_fold_k adds them in k order and
                        // runs the epilogue: deterministic, no atomics on C
    float* part;        // [batch][splitk][tiles][4][256] f32x4 slots (accumulator layout), uninitialised
    int* tickets;       // (unused since round 5)
    float* rs_part;     // [splitk][M] partial row sums of A (a_rowsum with splitk > 1)
    int a_vec, b_vec;   // 16-byte loads along the contiguous dimension are legal
    int batch;          // independent problems over blockIdx.z / splitk: A += z*bsa, B += z*bsb, C += z*bsc
    long bsa, bsb, bsc;
    int atomic_out;     // C (and a_rowsum) are updated by atomics even without split-K: several problems of a grouped launch may
                        // accumulate into the same destination (the same layer's gradient from two passes)
    float* a_rowsum;    // optional: a_rowsum[m] += sum_k A[m][k] (the bias gradient riding on a weight-gradient GEMM; k slices folded in order)
};

// One operand tile: `rows` (m or n) x GBK.  KFAST: memory is contiguous along k -> LDS layout [row][k];
// otherwise contiguous along the row index -> LDS layout [k][row].  Each thread owns 8 elements = 2 float4.
template <bool KFAST>
struct TileIO {
    f32x4 r[2];
    __device__ __forceinline__ void load(const float* base, long s_row, long s_k, int row0, int nrows, int k0, int k_end,
                                         bool vec, int tid) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = tid + e * 256;
            int row, k;
            if (KFAST) { row = idx >> 3; k = (idx & 7) * 4; } else { k = idx >> 4; row = (idx & 15) * 4; }
            const int gr = row0 + row, gk = k0 + k;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (KFAST) {
                if (gr < nrows) {
                    const float* p = base + (long)gr * s_row + (long)gk * s_k;
                    if (vec && gk + 3 < k_end) v = *reinterpret_cast<const f32x4u*>(p);
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (gk + j < k_end) v[j] = p[(long)j * s_k];
                    }
                }
            } else {
                if (gk < k_end) {
                    const float* p = base + (long)gr * s_row + (long)gk * s_k;
                    if (vec && gr + 3 < nrows) v = *reinterpret_cast<const f32x4u*>(p);
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (gr + j < nrows) v[j] = p[(long)j * s_row];
                    }
                }
            }
            r[e] = v;
        }
    }
    __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = tid + e * 256;
            if (KFAST) { int row = idx >> 3, k = (idx & 7) * 4; *reinterpret_cast<f32x4*>(lds + row * LDK + k) = r[e]; }
            else { int k = idx >> 4, row = (idx & 15) * 4; *reinterpret_cast<f32x4*>(lds + k * LDM + row) = r[e]; }
        }
    }
    // element (row, k) of the staged tile
    static __device__ __forceinline__ float at(const float* lds, int row, int k) {
        return KFAST ? lds[row * LDK + k] : lds[k * LDM + row];
    }
};

// Epilogue of one 64 x 64 tile: bias, activation, (accumulating / atomic / two-destination) stores.  acc[n-tile][m-tile] in the MFMA layout of
// gemm_tile (a lane holds four consecutive n of one m).
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x4 (&acc)[2][2], const int m0, const int n0, const int wm, const int wn,
                                              const int li, const int g, const int kz) {
    const bool atomic_k = (a.splitk > 1 && !a.part) || a.atomic_out;     // atomic split-K (parameter gradients: order-dependent rounding is acceptable)
    const bool vec_ok = a.scn == 1 && !a.C2 && !atomic_k;                  // (rows need not be 16-byte aligned: f32x4u)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int m = m0 + wm + y * 16 + li;
            const int nb = n0 + wn + x * 16 + 4 * g;
            if (m >= a.M || nb >= a.N) continue;
            f32x4 v = acc[x][y];
            if (a.bias && (!(a.splitk > 1 && !a.part) || kz == 0)) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nb + r < a.N) v[r] += a.bias[nb + r];
            }
            if (a.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = 1.f / (1.f + __expf(-v[r]));
            }
            float* c = a.C + (long)m * a.scm + (long)nb * a.scn;
            if (vec_ok && nb + 3 < a.N) {
                if (a.accumulate) { f32x4 o = *reinterpret_cast<f32x4u*>(c); v += o; }
                *reinterpret_cast<f32x4u*>(c) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (nb + r >= a.N) continue;
                    float* cc = c + (long)r * a.scn;
                    if (atomic_k) { atomicAdd(cc, v[r]); continue; }
                    *cc = a.accumulate ? *cc + v[r] : v[r];
                    if (a.C2) a.C2[(long)m * a.sc2m + (long)(nb + r) * a.sc2n] = v[r];
                }
            }
        }
}

// One 64 x 64 output tile of one problem: (bx, by) = tile, bzi = batch index * splitk + k slice, (gdx, gdy) = tiles of the problem.
template <bool AK, bool BK_>
__device__ __forceinline__ void gemm_tile(GemmArgs a, const int bx, const int by, const int bzi, const int gdx, const int gdy) {
    const int bz = bzi / a.splitk, kz = bzi - bz * a.splitk;
    a.A += (long)bz * a.bsa; a.B += (long)bz * a.bsb; a.C += (long)bz * a.bsc;
    __shared__ __attribute__((aligned(16))) float As[GBM * LDK > GBK * LDM ? GBM * LDK : GBK * LDM];
    __shared__ __attribute__((aligned(16))) float Bs[GBN * LDK > GBK * LDM ? GBN * LDK : GBK * LDM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int m0 = by * GBM, n0 = bx * GBN;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int kper = ((a.K + a.splitk - 1) / a.splitk + GBK - 1) / GBK * GBK;
    const int k_begin = kz * kper;
    const int k_end = min(a.K, k_begin + kper);

    f32x4 acc[2][2];   // [n-tile][m-tile]
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};

    TileIO<AK> ta;
    TileIO<BK_> tb;
    float rowsum = 0.f;
    if (k_begin < k_end) {
        ta.load(a.A, a.sam, a.sak, m0, a.M, k_begin, k_end, a.a_vec, tid);
        tb.load(a.B, a.sbn, a.sbk, n0, a.N, k_begin, k_end, a.b_vec, tid);
    }
    for (int k0 = k_begin; k0 < k_end; k0 += GBK) {
        __syncthreads();                    // previous tile fully consumed
        ta.store(As, tid);
        tb.store(Bs, tid);
        __syncthreads();
        if (k0 + GBK < k_end) {             // prefetch the next tile while this one is multiplied
            ta.load(a.A, a.sam, a.sak, m0, a.M, k0 + GBK, k_end, a.a_vec, tid);
            tb.load(a.B, a.sbn, a.sbk, n0, a.N, k0 + GBK, k_end, a.b_vec, tid);
        }
        if (a.a_rowsum && bx == 0 && tid < GBM) {       // the first column of workgroups also sums the rows of A
#pragma unroll
            for (int kk = 0; kk < GBK; ++kk) rowsum += TileIO<AK>::at(As, tid, kk);      // (out-of-range elements are staged as 0)
        }
#pragma unroll
        for (int ks = 0; ks < GBK; ks += 4) {
            float af[2], bf[2];
#pragma unroll
            for (int y = 0; y < 2; ++y) af[y] = TileIO<AK>::at(As, wm + y * 16 + li, ks + g);
#pragma unroll
            for (int x = 0; x < 2; ++x) bf[x] = TileIO<BK_>::at(Bs, wn + x * 16 + li, ks + g);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
                    // D[row = n_local][col = m_local] : A-operand = B tile (i = n), B-operand = A tile (j = m)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[x], af[y], acc[x][y], 0, 0, 0);
        }
    }

    if (a.splitk > 1 && a.part) {
        // deterministic split-K, first half: park this slice's accumulators (and its share of the row sums); gemm_fold_k adds the slices in
        // k order and runs the epilogue.  (Until round 5 the last slice of a tile to arrive -- a ticket -- folded them inside this kernel; the
        // device-scope fences that needs cost ~30 us per slice on this multi-XCD part: sk = 2 was 1.7x SLOWER than sk = 1.)
        const int ntiles = gdx * gdy, tile = by * gdx + bx;
        f32x4* mine = reinterpret_cast<f32x4*>(a.part) + (((long)bz * a.splitk + kz) * ntiles + tile) * (4 * 256);
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) mine[(x * 2 + y) * 256 + tid] = acc[x][y];
        if (a.a_rowsum && bx == 0 && tid < GBM && m0 + tid < a.M) a.rs_part[(long)kz * a.M + m0 + tid] = rowsum;
        return;
    }
    if (a.a_rowsum && bx == 0 && tid < GBM && m0 + tid < a.M) {
        if (a.splitk > 1 || a.atomic_out) atomicAdd(&a.a_rowsum[m0 + tid], rowsum);
        else a.a_rowsum[m0 + tid] += rowsum;                             // one workgroup per row block: plain read-modify-write
    }
    gemm_epilogue(a, acc, m0, n0, wm, wn, li, g, kz);
}

// Second half of the split-K: one workgroup per output tile adds the parked slices IN k ORDER (whichever order they were written in, the
// summation order is the same) and runs the ordinary epilogue (bias, act, accumulate, C2; the row sums of A likewise).
__device__ __forceinline__ void gemm_fold_tile(GemmArgs a, const int bx, const int by, const int bz, const int gdx, const int gdy) {
    a.C += (long)bz * a.bsc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int m0 = by * GBM, n0 = bx * GBN;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int ntiles = gdx * gdy, tile = by * gdx + bx;
    const f32x4* all = reinterpret_cast<const f32x4*>(a.part) + ((long)bz * a.splitk * ntiles + tile) * (4 * 256) + tid;
    const long zs = (long)ntiles * (4 * 256);
    f32x4 acc[2][2];
    // the loads of four slices x four fragments (16 independent 16-byte loads) are in flight together
    for (int z0 = 0; z0 < a.splitk; z0 += 4) {
        f32x4 v[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                v[j][q] = (z0 + j < a.splitk) ? __builtin_nontemporal_load(all + (long)(z0 + j) * zs + q * 256) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (z0 + j >= a.splitk) break;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (z0 + j == 0) acc[q >> 1][q & 1] = v[j][q];
                else acc[q >> 1][q & 1] += v[j][q];
            }
        }
    }
    if (a.a_rowsum && bx == 0 && tid < GBM && m0 + tid < a.M) {
        float r = a.rs_part[m0 + tid];
        for (int z = 1; z < a.splitk; ++z) r += a.rs_part[(long)z * a.M + m0 + tid];
        if (a.atomic_out) atomicAdd(&a.a_rowsum[m0 + tid], r);
        else a.a_rowsum[m0 + tid] += r;
    }
    gemm_epilogue(a, acc, m0, n0, wm, wn, li, g, 0);
}

__global__ __launch_bounds__(256) void gemm_fold_k(GemmArgs a) {
    gemm_fold_tile(a, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

template <bool AK, bool BK_>
__global__ __launch_bounds__(256) void gemm_mfma_k(GemmArgs a) {
    gemm_tile<AK, BK_>(a, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

// ---- round 6: larger block tiles for the k-contiguous GEMMs (forward X @ W^T and, through the transposed weight copies, dY @ W) -------------
// tools/bench_bf16x6.py's ladder showed the 64 x 64 kernel bound by operand movement, not by the matrix pipe: a 64 x 64 tile moves (64 + 64) x K x 4
// bytes through L2 -> registers -> LDS per 2 x 64 x 64 x K flop (16 flop / byte; the same kernel with its MFMAs all but removed still takes 53 % of its
// time).  gemm_big_k computes the SAME sums -- per output element the identical sequence of v_mfma_f32_16x16x4_f32 (k-tiles of 32 in order, k = ks + g
// inside a tile, C = 0 start, the shared epilogue) so results are BIT-IDENTICAL to gemm_mfma_k<true, true> at splitk = 1 -- with a BM x BN block tile of
// four waves owning (BM/2) x (BN/2) each: 32 flop / byte at 128 x 128, 21 at 128 x 64, 8 scalar fragment reads per 16 MFMAs instead of 4 per 4, and a
// double-buffered LDS tile (ONE barrier per 32 k).  No split-K / batch / row sums / second destination: rv_gemm routes those to gemm_mfma_k.
// MEASURED (tools/bench_gemm_big.py, profiles/r06_gemm_big_tiles.txt): bit-identical on every shape and SLOWER on 13 of 15 (x0.56 .. x1.00; 5120 x 1536 x 768:
// 127 vs 139 us is the one win) -- thousands of small workgroups at 6-8 per CU hide the load -> LDS -> MFMA chain better than 2-3 large ones; the operand-traffic
// reading of the bf16x6 ladder was wrong.  Kept as an opt-in policy (RV_GEMM_BIG=1 / rv_debug_set_gemm_big) so that the table can be re-measured; default off.
template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_big_k(GemmArgs a) {
    constexpr int TY = BM / 32, TX = BN / 32;            // 16 x 16 tiles per wave along m / n (wave grid 2 x 2)
    constexpr int LA = BM / 32, LB = BN / 32;            // float4 per thread and k-tile
    extern __shared__ __attribute__((aligned(16))) float smemg[];      // [2][BM][LDK] | [2][BN][LDK]
    float* As0 = smemg;
    float* Bs0 = smemg + 2 * BM * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    f32x4 acc[TX][TY];
#pragma unroll
    for (int x = 0; x < TX; ++x)
#pragma unroll
        for (int y = 0; y < TY; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ra[LA], rb[LB];
    auto load = [&](f32x4* r, const int n4, const float* base, const long s_row, const int row0, const int nrows, const int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e >= n4) break;
            const int idx = tid + e * 256, row = idx >> 3, k = (idx & 7) * 4;
            const int gr = row0 + row, gk = k0 + k;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (gr < nrows && gk < a.K) {
                const float* p = base + (long)gr * s_row + gk;
                if (gk + 3 < a.K) v = *reinterpret_cast<const f32x4u*>(p);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (gk + j < a.K) v[j] = p[j];
                }
            }
            r[e] = v;
        }
    };
    auto store = [&](const f32x4* r, const int n4, float* lds) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e >= n4) break;
            const int idx = tid + e * 256, row = idx >> 3, k = (idx & 7) * 4;
            *reinterpret_cast<f32x4*>(lds + row * LDK + k) = r[e];
        }
    };
    const int ntile = (a.K + GBK - 1) / GBK;
    load(ra, LA, a.A, a.sam, m0, a.M, 0);
    load(rb, LB, a.B, a.sbn, n0, a.N, 0);
    store(ra, LA, As0);
    store(rb, LB, Bs0);
    if (ntile > 1) {
        load(ra, LA, a.A, a.sam, m0, a.M, GBK);
        load(rb, LB, a.B, a.sbn, n0, a.N, GBK);
    }
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const float* As = As0 + (t & 1) * BM * LDK;
        const float* Bs = Bs0 + (t & 1) * BN * LDK;
        if (t + 1 < ntile) {                   // the next tile (registers) into the other buffer; the one after next leaves for the registers
            store(ra, LA, As0 + ((t + 1) & 1) * BM * LDK);
            store(rb, LB, Bs0 + ((t + 1) & 1) * BN * LDK);
            if (t + 2 < ntile) {
                load(ra, LA, a.A, a.sam, m0, a.M, (t + 2) * GBK);
                load(rb, LB, a.B, a.sbn, n0, a.N, (t + 2) * GBK);
            }
        }
#pragma unroll
        for (int ks = 0; ks < GBK; ks += 4) {
            float af[TY], bf[TX];
#pragma unroll
            for (int y = 0; y < TY; ++y) af[y] = As[(wm + y * 16 + li) * LDK + ks + g];
#pragma unroll
            for (int x = 0; x < TX; ++x) bf[x] = Bs[(wn + x * 16 + li) * LDK + ks + g];
#pragma unroll
            for (int x = 0; x < TX; ++x)
#pragma unroll
                for (int y = 0; y < TY; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[x], af[y], acc[x][y], 0, 0, 0);
        }
        __syncthreads();
    }
    // the epilogue of gemm_mfma_k, tile pair by tile pair (bias, activation, accumulate)
#pragma unroll
    for (int x0 = 0; x0 < TX; x0 += 2)
#pragma unroll
        for (int y0 = 0; y0 < TY; y0 += 2) {
            f32x4 sub[2][2];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) sub[x][y] = acc[x0 + x][y0 + y];
            gemm_epilogue(a, sub, m0, n0, wm + y0 * 16, wn + x0 * 16, li, g, 0);
        }
}

template <int BM, int BN>
static void launch_gemm_big(const GemmArgs& a, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(float);
    static std::once_flag once;
    std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)gemm_big_k<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    hipLaunchKernelGGL((gemm_big_k<BM, BN>), dim3(cdiv(a.N, BN), cdiv(a.M, BM)), dim3(256), lds, st, a);
}

// Grouped launch: a device table of independent problems (all of the same operand orientation), one 1-D grid over the tiles of
// all of them.  The parameter-gradient GEMMs of a backward pass (M or N of 31 .. 229, K = B*T = 5120: a few dozen workgroups
// each, latency-bound alone) then run side by side in ONE launch instead of one under-filled launch each.
struct GemmEntry { GemmArgs a; int block0, gx, gy, fold0; };     // fold0: prefix of the fold workgroups (tiles x batch of the entries that park)
template <bool AK, bool BK_>
__global__ __launch_bounds__(256) void gemm_table_k(const GemmEntry* tab, int count) {
    int e = 0;
    while (e + 1 < count && (int)blockIdx.x >= tab[e + 1].block0) ++e;            // wave-uniform scan (scalar loads), count <= 64
    const GemmEntry& t = tab[e];
    const int local = blockIdx.x - t.block0;
    const int bx = local % t.gx, rest = local / t.gx;
    gemm_tile<AK, BK_>(t.a, bx, rest % t.gy, rest / t.gy, t.gx, t.gy);
}

// The folds of a grouped launch's parked entries, as one more 1-D grid (entries that do not park own no workgroups here).
__global__ __launch_bounds__(256) void gemm_table_fold_k(const GemmEntry* tab, int count) {
    int e = 0;
    while (e + 1 < count && (int)blockIdx.x >= tab[e + 1].fold0) ++e;
    const GemmEntry& t = tab[e];
    const int local = blockIdx.x - t.fold0;
    const int bx = local % t.gx, rest = local / t.gx;
    gemm_fold_tile(t.a, bx, rest % t.gy, rest / t.gy, t.gx, t.gy);
}

__global__ void zero_strided_k(float* c, long scm, long scn, int M, int N) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M * N) return;
    long m = i / N; int n = (int)(i - m * N);
    c[m * scm + n * scn] = 0.f;
}

// block-tile policy of rv_gemm (see gemm_big_k): 0 = always 64 x 64 (DEFAULT: profiles/r06_gemm_big_tiles.txt -- the larger tiles are bit-identical and
// slower on 13 of 15 shapes), 1 = automatic (RV_GEMM_BIG=1), 2 / 3 = force 128 x 128 / 128 x 64
static std::atomic<int> g_gemm_big{-1};
static int gemm_big_mode() {
    int m = g_gemm_big.load(std::memory_order_relaxed);
    if (m < 0) {
        m = getenv("RV_GEMM_BIG") ? atoi(getenv("RV_GEMM_BIG")) : 0;
        g_gemm_big.store(m, std::memory_order_relaxed);
    }
    return m;
}

extern "C" {

// experiment / test hook (not part of include/reconvat_hip.h): set the block-tile policy, returns the previous one
int rv_debug_set_gemm_big(int mode) {
    const int prev = gemm_big_mode();
    g_gemm_big.store(mode, std::memory_order_relaxed);
    return prev;
}

// C[m*scm + n*scn] (+)= act(sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] + bias[n]);  act: 0 none, 1 sigmoid.
// splitk > 1 splits the reduction over extra workgroups.  With splitk_ws == NULL the slices accumulate by fp32 atomics (C zeroed
// first unless accumulate; act must be 0, C2 null; rounding depends on arrival order -- used for parameter gradients).  With
// splitk_ws the split is DETERMINISTIC: every k slice parks its partial tile in `splitk_ws` (rv_gemm_splitk_workspace_bytes,
// uninitialised) and a second launch (gemm_fold_k, same stream) adds the slices in k order and runs the ordinary epilogue (bias, act,
// accumulate, C2 all allowed): results do not depend on the arrival order, no atomics touch C.  `splitk_tickets` is unused since
// round 5 (may be NULL; rv_gemm_splitk_ticket_bytes returns 0): the in-kernel ticketed fold it served needed device-scope fences
// that cost more than the second launch.
// C2 (nullable) receives a second copy with its own strides.
// batch > 1: `batch` independent problems of the same shape, problem z at A + z*bsa, B + z*bsb, C + z*bsc (C2 must be null).
// a_rowsum (nullable, batch == 1): a_rowsum[m] += sum_k A[m][k] -- the bias gradient of a linear layer for free on its
// weight-gradient GEMM (A = dY^T); also folded in k order.
long rv_gemm_splitk_workspace_bytes(int M, int N, int splitk, int batch) {
    if (splitk <= 1) return 0;
    return ((long)batch * splitk * cdiv(M, GBM) * cdiv(N, GBN) * (4 * 256 * 16)) + (long)splitk * M * 4;
}
long rv_gemm_splitk_ticket_bytes(int M, int N, int splitk, int batch) {
    (void)M; (void)N; (void)splitk; (void)batch;
    return 0;
}

static int gemm_args_make(GemmArgs& a, const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long scm, long scn,
                          float* C2, long sc2m, long sc2n, const float* bias, int M, int N, int K, int act, int accumulate, int splitk,
                          int batch, long bsa, long bsb, long bsc, float* a_rowsum, void* splitk_ws, void* splitk_tickets) {
    RV_CHECK_ARG(M > 0 && N > 0 && K > 0, "rv_gemm: empty problem");
    RV_CHECK_ARG(!a_rowsum || batch == 1, "rv_gemm: a_rowsum excludes batch");
    RV_CHECK_ARG(splitk >= 1, "rv_gemm: splitk must be >= 1");
    RV_CHECK_ARG(batch >= 1 && (batch == 1 || !C2) && (long)batch * splitk < 65536, "rv_gemm: bad batch %d", batch);
    if (splitk > 1 && !splitk_ws) RV_CHECK_ARG(act == 0 && !C2, "rv_gemm: atomic split-K excludes act/C2");
    a.A = A; a.sam = sam; a.sak = sak; a.B = B; a.sbk = sbk; a.sbn = sbn; a.C = C; a.scm = scm; a.scn = scn;
    a.C2 = C2; a.sc2m = sc2m; a.sc2n = sc2n; a.bias = bias; a.M = M; a.N = N; a.K = K; a.act = act;
    a.accumulate = accumulate; a.splitk = splitk; a.batch = batch; a.bsa = bsa; a.bsb = bsb; a.bsc = bsc; a.a_rowsum = a_rowsum;
    a.part = (float*)splitk_ws; a.tickets = (int*)splitk_tickets; a.atomic_out = 0;
    a.rs_part = (splitk > 1 && splitk_ws) ? (float*)splitk_ws + (long)batch * splitk * cdiv(M, GBM) * cdiv(N, GBN) * (4 * 256 * 4) : nullptr;
    const bool a_kfast = (sak <= sam), b_kfast = (sbk <= sbn);
    // 16-byte loads need a unit stride along the fast dimension, a 16-byte multiple along the slow one and an
    // aligned base; K-split offsets are multiples of GBK so they preserve alignment
    static const int uvec_env = getenv("RV_GEMM_UNALIGNED_VEC") ? atoi(getenv("RV_GEMM_UNALIGNED_VEC")) : 1;
    a.a_vec = ((a_kfast ? sak : sam) == 1) && (uvec_env || ((((a_kfast ? sam : sak) & 3) == 0) && ((((uintptr_t)A) & 15) == 0) && (batch == 1 || (bsa & 3) == 0)));
    a.b_vec = ((b_kfast ? sbk : sbn) == 1) && (uvec_env || ((((b_kfast ? sbn : sbk) & 3) == 0) && ((((uintptr_t)B) & 15) == 0) && (batch == 1 || (bsb & 3) == 0)));
    return RV_OK;
}

int rv_gemm(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long scm, long scn, float* C2,
            long sc2m, long sc2n, const float* bias, int M, int N, int K, int act, int accumulate, int splitk, int batch,
            long bsa, long bsb, long bsc, float* a_rowsum, void* splitk_ws, void* splitk_tickets, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    GemmArgs a;
    const int rc = gemm_args_make(a, A, sam, sak, B, sbk, sbn, C, scm, scn, C2, sc2m, sc2n, bias, M, N, K, act, accumulate, splitk, batch,
                                  bsa, bsb, bsc, a_rowsum, splitk_ws, splitk_tickets);
    if (rc != RV_OK) return rc;
    const bool a_kfast = (sak <= sam), b_kfast = (sbk <= sbn);
    if (splitk > 1 && !splitk_ws && !accumulate) {
        RV_CHECK_ARG(batch == 1, "rv_gemm: batched atomic split-K needs accumulate (zero C yourself)");
        hipLaunchKernelGGL(zero_strided_k, dim3(cdiv((long)M * N, 256)), dim3(256), 0, st, C, scm, scn, M, N);
        RV_LAUNCH_CHECK("rv_gemm(zero)");
    }
    dim3 grid(cdiv(N, GBN), cdiv(M, GBM), splitk * batch);
    // larger block tiles for the plain k-contiguous problems (bit-identical sums, see gemm_big_k): the largest tile that still leaves >= 2 workgroups per CU
    // (>= 1.5 for 128 x 64) -- a short grid of big tiles loses more to the tail than it gains in operand traffic.  RV_GEMM_BIG=0: always 64 x 64.
    const int big_env = gemm_big_mode();
    if (big_env && a_kfast && b_kfast && splitk == 1 && batch == 1 && !a_rowsum && !C2 && sak == 1 && sbk == 1 && scn == 1 && M >= 128) {
        const long wg128 = (long)cdiv(M, 128) * cdiv(N, 128), wg64 = (long)cdiv(M, 128) * cdiv(N, 64);
        const int force = big_env > 1 ? big_env : 0;              // (2: force 128 x 128, 3: force 128 x 64 -- tests / A-B runs)
        if (force == 2 || (!force && wg128 >= 512 && (long)cdiv(N, 128) * 128 * 10 <= (long)cdiv(N, 64) * 64 * 11)) {
            launch_gemm_big<128, 128>(a, st);
            RV_LAUNCH_CHECK("rv_gemm(128x128)");
            return RV_OK;
        }
        if (force == 3 || (!force && wg64 >= 384)) {
            launch_gemm_big<128, 64>(a, st);
            RV_LAUNCH_CHECK("rv_gemm(128x64)");
            return RV_OK;
        }
    }
    if (a_kfast && b_kfast) hipLaunchKernelGGL((gemm_mfma_k<true, true>), grid, dim3(256), 0, st, a);
    else if (a_kfast && !b_kfast) hipLaunchKernelGGL((gemm_mfma_k<true, false>), grid, dim3(256), 0, st, a);
    else if (!a_kfast && b_kfast) hipLaunchKernelGGL((gemm_mfma_k<false, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_mfma_k<false, false>), grid, dim3(256), 0, st, a);
    RV_LAUNCH_CHECK("rv_gemm");
    if (splitk > 1 && splitk_ws) {
        hipLaunchKernelGGL(gemm_fold_k, dim3(cdiv(N, GBN), cdiv(M, GBM), batch), dim3(256), 0, st, a);
        RV_LAUNCH_CHECK("rv_gemm(fold)");
    }
    return RV_OK;
}

// Grouped form (see gemm_table_k): fill HOST entries one by one with the arguments of rv_gemm -- accumulate must be 1; the problems ADD
// into their destinations (gradient accumulation; two entries may share one), so the final adds are fp32 atomics.  splitk > 1 with a
// workspace (rv_gemm_splitk_workspace_bytes, one per entry): the k slices are PARKED and a second grouped launch folds them in k order --
// one atomic per output element instead of one per slice (the far atomics of a multi-XCD part are the slow half of an atomic split-K);
// without a workspace every slice adds atomically.  All entries of one table must have the same operand orientation
// (rv_gemm_table_fill returns it: bit 0 = A k-fast, bit 1 = B k-fast; negative = error) --, finalize (prefix sums of the workgroup
// counts of both launches), copy the table to the device and run it.
long rv_gemm_table_entry_bytes(void) { return (long)sizeof(GemmEntry); }

long rv_gemm_table_fill(void* entry_host, const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long scm,
                        long scn, const float* bias, int M, int N, int K, int splitk, int batch, long bsa, long bsb, long bsc,
                        float* a_rowsum, void* splitk_ws) {
    if (!entry_host) { rv_set_error("rv_gemm_table_fill: null entry"); return RV_EINVAL; }
    GemmEntry* e = (GemmEntry*)entry_host;
    const int rc = gemm_args_make(e->a, A, sam, sak, B, sbk, sbn, C, scm, scn, nullptr, 0, 0, bias, M, N, K, 0, 1, splitk, batch, bsa, bsb,
                                  bsc, a_rowsum, splitk > 1 ? splitk_ws : nullptr, nullptr);
    if (rc != RV_OK) return rc;
    e->a.atomic_out = 1;
    e->gx = cdiv(N, GBN); e->gy = cdiv(M, GBM);
    e->block0 = e->gx * e->gy * splitk * batch;                           // block0 / fold0: counts until finalize
    e->fold0 = (splitk > 1 && splitk_ws) ? e->gx * e->gy * batch : 0;
    return ((sak <= sam) ? 1 : 0) | ((sbk <= sbn) ? 2 : 0);
}

// -> the workgroups of the grouped launch; *fold_blocks (nullable) <- the workgroups of its fold launch (0: no entry parks)
long rv_gemm_table_finalize(void* table_host, int count, long* fold_blocks) {
    GemmEntry* t = (GemmEntry*)table_host;
    long total = 0, ftotal = 0;
    for (int i = 0; i < count; ++i) {
        const int n = t[i].block0; t[i].block0 = (int)total; total += n;
        const int f = t[i].fold0; t[i].fold0 = (int)ftotal; ftotal += f;
    }
    if (fold_blocks) *fold_blocks = ftotal;
    return total;
}

int rv_gemm_table_run(const void* table_dev, int count, long total_blocks, long fold_blocks, int orientation, void* stream) {
    RV_CHECK_ARG(count > 0 && count <= 64 && total_blocks > 0 && total_blocks < (1L << 31), "rv_gemm_table_run: empty or oversized table");
    RV_CHECK_ARG(fold_blocks >= 0 && fold_blocks < (1L << 31), "rv_gemm_table_run: bad fold grid");
    hipStream_t st = (hipStream_t)stream;
    const GemmEntry* t = (const GemmEntry*)table_dev;
    dim3 grid((unsigned)total_blocks);
    switch (orientation & 3) {
        case 3: hipLaunchKernelGGL((gemm_table_k<true, true>), grid, dim3(256), 0, st, t, count); break;
        case 1: hipLaunchKernelGGL((gemm_table_k<true, false>), grid, dim3(256), 0, st, t, count); break;
        case 2: hipLaunchKernelGGL((gemm_table_k<false, true>), grid, dim3(256), 0, st, t, count); break;
        default: hipLaunchKernelGGL((gemm_table_k<false, false>), grid, dim3(256), 0, st, t, count); break;
    }
    RV_LAUNCH_CHECK("rv_gemm_table_run");
    if (fold_blocks > 0) {
        hipLaunchKernelGGL(gemm_table_fold_k, dim3((unsigned)fold_blocks), dim3(256), 0, st, t, count);
        RV_LAUNCH_CHECK("rv_gemm_table_run(fold)");
    }
    return RV_OK;
}

}  // extern "C"
