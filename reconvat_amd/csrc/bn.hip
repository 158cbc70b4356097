// Train/eval BatchNorm2d + leaky-ReLU (+ residual add) on NHWC fp32 for gfx950.
//
// Reference semantics: nn.BatchNorm2d(C, momentum=0.1, eps=1e-5) followed by F.leaky_relu(0.01)
// (model/UNet_onset.py:183-201, :216-224).  HBM-bound: every kernel streams [P, C] once with
// coalesced channel-fastest 16-byte accesses; per-channel sums are carried in fp64 (fp32 per-thread
// partials over <= 16 pixels, fp64 across threads/blocks) so that var = E[x^2] - mean^2 is safe.
//
//   stats    : sum / sum-of-squares per channel                    (reads z)
//   apply    : y = lrelu(z*scale + shift) (+ residual)             (reads z [,res], writes y); every thread
//              derives its 4 channels' scale/shift from the fp64 sums, block 0 also does the "finalize" work
//              (coefficients for backward, running statistics, num_batches_tracked)
//   bwd_red  : sum(dzh), sum(dzh*xhat)  with dzh = dy * lrelu'(zh) (reads dy, z)
//   bwd_apply: dz = scale * (dzh - mean(dzh) - xhat*mean(dzh*xhat)) (reads dy, z, writes dz)
#include "common.h"

#define BN_PIX_PER_THREAD 16

// running <- (1 - momentum) * running + momentum * batch, with the rounding sequence pinned (one multiply, one fused
// multiply-add) so that the fused finalize, the per-layer replay and the table replay are bit-identical
__device__ __forceinline__ float bn_momentum_update(float running, float batch, float momentum) {
    return __fmaf_rn(momentum, batch, __fmul_rn(1.f - momentum, running));
}

struct BnArgs {
    const float* z; int z_ld;
    const float* dy; int dy_ld;
    float* out; int out_ld;
    const float* res; int res_ld;
    // fused skip conv (round 6): res[p][c] := skip(x)[p][c], the 1x1 convolution of an encoder block (model/UNet_onset.py:191,198) evaluated HERE instead of
    // being written by a conv launch and read back.  Bit-identical to that launch: v_mfma_f32_16x16x4_f32 is a chain of fused multiply-adds in k order
    // starting from C (tools/probes/mfma_f32_order.hip), so a vector-ALU fmaf chain in the conv kernel's k order reproduces it.
    //   sk_cin == 1 : rank-1 term fma(x[p], w[c], b[c])                      (block 1: conv_small_k<1, C, 1, 1, 1, 0>)
    //   sk_cin in {16, 32, 64}: dense, weights [C][sk_cin] staged transposed in LDS, chain order of conv_mfma_k<1, 1, 1, 0, 4, ...>: per 16-channel chunk
    //                 r = 0..3, g = 0..3 -> channel 16 chunk + 4 g + r; sk_ks != 0: the K-split form (family 0x5NM: four partial chains over the chunk
    //                 ranges of the four waves, added in wave order); then + bias
    const float* r1_x; int r1_x_ld; const float* r1_w; const float* r1_b; int sk_cin, sk_ks;
    long P; int C;
    double* sums;            // [2C] fp64 accumulators
    const float* coef;       // [4C]: mean, invstd, scale, shift
    float slope;
    int frozen;              // eval-mode statistics: no batch-mean terms in the backward
    float* dgamma; float* dbeta;
    int accumulate;
    int param_accumulate;    // dgamma/dbeta are added into the destination instead of overwriting it
    // forward finalize (fused into the apply kernel)
    const float* gamma; const float* beta;
    float* running_mean; float* running_var; long* nbt;
    float* coef_out;         // [4C] written by block 0 (saved for backward)
    float momentum, eps;
    int training;
    int ppt;                 // pixels per thread of the reduction kernels (multiple of 4)
};

// thread t -> channel quad t % (C/4), pixel lane t / (C/4); 16-byte loads, BN_PIX_PER_THREAD pixels each.
template <bool BWD>
__global__ __launch_bounds__(256) void bn_reduce_k(BnArgs a) {
    __shared__ float sh[256 * 8];
    const int C = a.C, C4 = C >> 2, PL = 256 / C4;
    const int t = threadIdx.x;
    const int c = (t % C4) * 4, pl = t / C4;
    f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (pl < PL) {
        const long p0 = (long)blockIdx.x * PL * a.ppt;
        f32x4 mean, invstd, scale, shift;
        if (BWD) {
            mean = *reinterpret_cast<const f32x4*>(a.coef + c);
            invstd = *reinterpret_cast<const f32x4*>(a.coef + C + c);
            scale = *reinterpret_cast<const f32x4*>(a.coef + 2 * C + c);
            shift = *reinterpret_cast<const f32x4*>(a.coef + 3 * C + c);
        }
        // four pixels per trip: all eight 16-byte loads are issued before the first use (a streaming kernel lives on
        // bytes in flight); out-of-range pixels load pixel 0 and are masked out of the sums
        for (int k = 0; k < a.ppt; k += 4) {
            f32x4 zv[4], dv[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long p = p0 + pl + (long)(k + u) * PL;
                ok[u] = p < a.P;
                const long pp = ok[u] ? p : 0;
                zv[u] = *reinterpret_cast<const f32x4*>(a.z + pp * a.z_ld + c);
                if (BWD) dv[u] = *reinterpret_cast<const f32x4*>(a.dy + pp * a.dy_ld + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!ok[u]) continue;
                const f32x4 z = zv[u];
                if (BWD) {
                    const f32x4 d = dv[u];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float zh = fmaf(z[q], scale[q], shift[q]);
                        float dd = zh > 0.f ? d[q] : d[q] * a.slope;
                        s0[q] += dd;
                        s1[q] += dd * ((z[q] - mean[q]) * invstd[q]);
                    }
                } else {
                    s0 += z;
#pragma unroll
                    for (int q = 0; q < 4; ++q) s1[q] = fmaf(z[q], z[q], s1[q]);
                }
            }
        }
    }
    // sh[(pl*C + ch)*2 + {0,1}]
    if (pl < PL) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sh[(pl * C + c + q) * 2] = s0[q];
            sh[(pl * C + c + q) * 2 + 1] = s1[q];
        }
    }
    __syncthreads();
    if (t < 2 * C) {
        const int ch = t >> 1, which = t & 1;
        double d = 0.0;
        for (int l = 0; l < PL; ++l) d += (double)sh[(l * C + ch) * 2 + which];
        atomicAdd(&a.sums[(blockIdx.x % RV_BN_NREP) * 2 * C + which * C + ch], d);
    }
}

// batch (training) or running (eval) mean / inverse std of channel c -- identical arithmetic in every thread
__device__ __forceinline__ void bn_coef(const BnArgs& a, const double* fold, int c, float& mean, float& invstd) {
    if (a.training) {
        const double n = (double)a.P;
        const double m = fold[c] / n;
        double var = fold[a.C + c] / n - m * m;
        if (var < 0.0) var = 0.0;
        mean = (float)m;
        invstd = (float)(1.0 / sqrt(var + (double)a.eps));
    } else {
        mean = a.running_mean[c];
        invstd = 1.0f / sqrtf(a.running_var[c] + a.eps);
    }
}

// elementwise, 4 channels per thread (C % 4 == 0).  The grid is a multiple of 3 workgroups, so the
// grid stride is a multiple of every C/4 in the model and a thread keeps ONE channel quad: its
// coefficients live in registers for the whole kernel.
template <bool BWD, int SKIP = 0>
__global__ __launch_bounds__(256) void bn_apply_k(BnArgs a) {
    static_assert(!BWD || SKIP == 0, "the fused skip conv exists in the forward kernel only");
    extern __shared__ __attribute__((aligned(16))) float skw[];        // SKIP: [SKIP][C] transposed skip weights
    const int C = a.C, C4 = C >> 2;
    const long total = a.P * C4;
    const long stride = (long)gridDim.x * blockDim.x;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool fixed = (stride % C4) == 0;
    int c = (int)(idx % C4) * 4;
    // the replicated sums are folded ONCE per workgroup (2C threads x 8 loads), not once per thread
    __shared__ double fold[256];
    if (BWD ? !a.frozen || a.dgamma : a.training) {
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) fold[i] = bn_sum_replicas(a.sums, C, i);
        __syncthreads();
    }
    f32x4 mean, invstd, scale, shift, k1, k2;
    auto fetch = [&](int cc) {
        if (BWD) {
            scale = *reinterpret_cast<const f32x4*>(a.coef + 2 * C + cc);
            shift = *reinterpret_cast<const f32x4*>(a.coef + 3 * C + cc);
            mean = *reinterpret_cast<const f32x4*>(a.coef + cc);
            invstd = *reinterpret_cast<const f32x4*>(a.coef + C + cc);
            const double invn = 1.0 / (double)a.P;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                k1[q] = a.frozen ? 0.f : (float)(fold[cc + q] * invn);
                k2[q] = a.frozen ? 0.f : (float)(fold[C + cc + q] * invn);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float m_, is_;
                bn_coef(a, fold, cc + q, m_, is_);
                scale[q] = a.gamma[cc + q] * is_;
                shift[q] = a.beta[cc + q] - m_ * scale[q];
            }
        }
    };
    fetch(c);
    f32x4 r1w = (f32x4){0.f, 0.f, 0.f, 0.f}, r1b = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto fetch_r1 = [&](int cc) {
        if (!BWD && a.r1_x) {
            if (SKIP == 0) r1w = *reinterpret_cast<const f32x4*>(a.r1_w + cc);
            if (a.r1_b) r1b = *reinterpret_cast<const f32x4*>(a.r1_b + cc);
        }
    };
    fetch_r1(c);
    if constexpr (SKIP > 0) {
        // weights [C][SKIP] (PyTorch layout) -> LDS [SKIP][C]: a thread's four channels of one k are one 16-byte read, the same for every thread of a channel quad
        for (int i = threadIdx.x; i < SKIP * C; i += blockDim.x) {
            const int cc = i / SKIP, kk = i - cc * SKIP;
            skw[kk * C + cc] = a.r1_w[i];
        }
        __syncthreads();
    }
    // skip(x)[p][c .. c+3] of the dense form (see BnArgs)
    auto skip_dense = [&](const long p, const int cc0) -> f32x4 {
        constexpr int NCH = SKIP > 0 ? SKIP / 16 : 1;
        const float* xp = a.r1_x + p * a.r1_x_ld;
        auto chunk = [&](f32x4& acc, const int ch) {
            f32x4 xg[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) xg[g] = *reinterpret_cast<const f32x4*>(xp + 16 * ch + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(&skw[(16 * ch + 4 * g + r) * C + cc0]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = fmaf(w4[q], xg[g][r], acc[q]);
                }
        };
        f32x4 tot = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!a.sk_ks) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) chunk(tot, ch);
        } else {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                f32x4 part = (f32x4){0.f, 0.f, 0.f, 0.f};
                for (int ch = (w * NCH) / 4; ch < ((w + 1) * NCH) / 4; ++ch) chunk(part, ch);
                if (w == 0) tot = part; else tot += part;
            }
        }
        return tot + r1b;                        // (r1b = the bias quad of this thread's channels, 0 without a bias)
    };
    auto one = [&](const f32x4& z, const f32x4& d, const f32x4& r) -> f32x4 {
        f32x4 o;
        if (!BWD) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float zh = fmaf(z[q], scale[q], shift[q]);
                o[q] = (zh > 0.f ? zh : zh * a.slope) + r[q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float zh = fmaf(z[q], scale[q], shift[q]);
                float dz = zh > 0.f ? d[q] : d[q] * a.slope;
                float xh = (z[q] - mean[q]) * invstd[q];
                dz = dz - k1[q] - xh * k2[q];
                o[q] = dz * scale[q];
            }
        }
        return o;
    };
    if (fixed) {
        // the thread keeps its channel quad: pixels p0, p0 + pstep, ... ; four pixels per trip, every load of the trip
        // issued before the first use, no division in the loop
        constexpr int UNR = BWD || SKIP > 0 ? 4 : 8;          // pixels in flight per thread
        const long pstep = stride / C4;
        long p = idx / C4;
        for (; p < a.P; p += UNR * pstep) {
            f32x4 zv[UNR], dv[UNR], rv[UNR];
            bool ok[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long pu = p + u * pstep;
                ok[u] = pu < a.P;
                const long pp = ok[u] ? pu : 0;
                zv[u] = *reinterpret_cast<const f32x4*>(a.z + pp * a.z_ld + c);
                if (BWD) dv[u] = *reinterpret_cast<const f32x4*>(a.dy + pp * a.dy_ld + c);
                else dv[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                rv[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (!BWD && a.res) rv[u] = *reinterpret_cast<const f32x4*>(a.res + pp * a.res_ld + c);
                if (!BWD && SKIP == 0 && a.r1_x) {
                    const float x1 = a.r1_x[pp * a.r1_x_ld];
#pragma unroll
                    for (int q = 0; q < 4; ++q) rv[u][q] = fmaf(x1, r1w[q], r1b[q]);       // bit-identical to conv_small_k<1, C, 1, 1, 1, 0>
                }
            }
            if constexpr (SKIP > 0) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) rv[u] = skip_dense(ok[u] ? p + u * pstep : 0, c);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (!ok[u]) continue;
                f32x4 o = one(zv[u], dv[u], rv[u]);
                float* dst = a.out + (p + u * pstep) * a.out_ld + c;
                if (a.accumulate) {
                    f32x4 old = *reinterpret_cast<f32x4*>(dst);
                    o += old;
                }
                *reinterpret_cast<f32x4*>(dst) = o;
            }
        }
        idx = total;                              // skip the generic loop below
    }
    for (; idx < total; idx += stride) {
        const long p = idx / C4;
        if (!fixed) { c = (int)(idx - p * C4) * 4; fetch(c); fetch_r1(c); }
        const f32x4 z = *reinterpret_cast<const f32x4*>(a.z + p * a.z_ld + c);
        f32x4 o;
        if (!BWD) {
            f32x4 r = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (a.res) r = *reinterpret_cast<const f32x4*>(a.res + p * a.res_ld + c);
            if (SKIP == 0 && a.r1_x) {
                const float x1 = a.r1_x[p * a.r1_x_ld];
#pragma unroll
                for (int q = 0; q < 4; ++q) r[q] = fmaf(x1, r1w[q], r1b[q]);
            }
            if constexpr (SKIP > 0) r = skip_dense(p, c);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float zh = fmaf(z[q], scale[q], shift[q]);
                o[q] = (zh > 0.f ? zh : zh * a.slope) + r[q];
            }
        } else {
            const f32x4 d = *reinterpret_cast<const f32x4*>(a.dy + p * a.dy_ld + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float zh = fmaf(z[q], scale[q], shift[q]);
                float dz = zh > 0.f ? d[q] : d[q] * a.slope;
                float xh = (z[q] - mean[q]) * invstd[q];
                dz = dz - k1[q] - xh * k2[q];
                o[q] = dz * scale[q];
            }
        }
        float* dst = a.out + p * a.out_ld + c;
        if (a.accumulate) {
            f32x4 old = *reinterpret_cast<f32x4*>(dst);
            o += old;
        }
        *reinterpret_cast<f32x4*>(dst) = o;
    }
    if (blockIdx.x == 0) {
        for (int cc = threadIdx.x; cc < a.C; cc += blockDim.x) {
            if (BWD) {
                if (a.dgamma) {
                    const float dg = (float)fold[a.C + cc], db = (float)fold[cc];
                    a.dgamma[cc] = a.param_accumulate ? a.dgamma[cc] + dg : dg;
                    a.dbeta[cc] = a.param_accumulate ? a.dbeta[cc] + db : db;
                }
            } else {
                // finalize: coefficients for the backward pass, running statistics, num_batches_tracked
                float m_, is_;
                bn_coef(a, fold, cc, m_, is_);
                const float sc = a.gamma[cc] * is_;
                a.coef_out[cc] = m_;
                a.coef_out[a.C + cc] = is_;
                a.coef_out[2 * a.C + cc] = sc;
                a.coef_out[3 * a.C + cc] = a.beta[cc] - m_ * sc;
                if (a.training) {
                    const double n = (double)a.P;
                    const double mm = fold[cc] / n;
                    double var = fold[a.C + cc] / n - mm * mm;
                    if (var < 0.0) var = 0.0;
                    const float unb = (float)(n > 1.0 ? var * n / (n - 1.0) : var);
                    a.coef_out[4 * a.C + cc] = unb;            // kept for a deferred running-stat update
                    if (a.training == 1) {
                        a.running_mean[cc] = bn_momentum_update(a.running_mean[cc], m_, a.momentum);
                        a.running_var[cc] = bn_momentum_update(a.running_var[cc], unb, a.momentum);
                    }
                }
            }
        }
        if (!BWD && threadIdx.x == 0 && a.training == 1 && a.nbt) *a.nbt += 1;
    }
}

// Grid of an apply launch: at most 3-4 workgroups per CU, each walking many pixels.  One workgroup per 256 quads (capped at 3072 = 1.5
// resident rounds of 8 per CU) paid the per-workgroup prologue (replica fold, fp64 coefficient math) twelve times per CU and left the
// second round half empty: 3072 -> 1023 is -0.2 ms per step in situ (2046 / 1536 / 1023 / 768 within noise of each other, 510 and
// 255 slower again: too few loads in flight).  The backward form (two input streams) likes 768 best: 1023 / 768 / 639 / 510 =
// 22.29 / 22.19 / 22.24 / 22.30 ms per step.  Round 5, with the re-tuned conv table, the forward form likes 768 / 639 better than 1023 as well
// (tools/knob_ab.sh, 6 interleaved runs each: 21.43 / 21.43 vs 21.52 ms; 510: 21.52) -> 768 for both.  RV_BN_MAXBLK / RV_BN_MAXBLK_BWD override (tuning only).
static int bn_apply_blocks(long total, bool bwd = false) {
    long b = (total + 255) / 256;
    static const long capf = getenv("RV_BN_MAXBLK") ? atol(getenv("RV_BN_MAXBLK")) : 768;
    static const long capb = getenv("RV_BN_MAXBLK_BWD") ? atol(getenv("RV_BN_MAXBLK_BWD")) : 768;
    const long cap = bwd ? capb : capf;
    if (b > cap) b = cap;
    b = ((b + 2) / 3) * 3;          // multiple of 3: see bn_apply_k
    return (int)b;
}

// Pixels per thread of a reduction launch.  Every workgroup ends with 2C fp64 atomics on the SAME 2C addresses, and those
// serialise at the memory side (~25 ns each, measured: 1 100 workgroups cost ~30 us of pure atomic tail on a 75 MB
// tensor), so the grid is sized to ~512 workgroups (two per CU keep enough loads in flight) instead of one per 16 pixels.
static int bn_ppt(long P, int PL) {
    long ppt = P / ((long)PL * 512);
    ppt = (ppt + 3) & ~3L;
    if (ppt < BN_PIX_PER_THREAD) ppt = BN_PIX_PER_THREAD;
    if (ppt > 256) ppt = 256;
    return (int)ppt;
}

int rv_internal_bn_stats(const float* z, int z_ld, long P, int C, double* sums, hipStream_t st) {
    RV_CHECK_ARG(C % 4 == 0 && C <= 128 && (z_ld % 4) == 0, "bn statistics: C=%d must be a multiple of 4 and <= 128", C);
    BnArgs a = {};
    a.z = z; a.z_ld = z_ld; a.P = P; a.C = C; a.sums = sums;
    const int PL = 256 / (C / 4);
    a.ppt = bn_ppt(P, PL);
    hipLaunchKernelGGL(bn_reduce_k<false>, dim3(cdiv(P, (long)PL * a.ppt)), dim3(256), 0, st, a);
    RV_LAUNCH_CHECK("bn statistics");
    return RV_OK;
}

int rv_internal_bn_bwd_stats(const float* dy, int dy_ld, const float* z, int z_ld, long P, int C, const float* coef, float slope,
                             double* sums, hipStream_t st) {
    RV_CHECK_ARG(C % 4 == 0 && C <= 128 && (z_ld % 4) == 0 && (dy_ld % 4) == 0, "bn backward statistics: C=%d unsupported", C);
    BnArgs a = {};
    a.z = z; a.z_ld = z_ld; a.dy = dy; a.dy_ld = dy_ld; a.P = P; a.C = C; a.sums = sums; a.coef = coef; a.slope = slope;
    const int PL = 256 / (C / 4);
    a.ppt = bn_ppt(P, PL);
    hipLaunchKernelGGL(bn_reduce_k<true>, dim3(cdiv(P, (long)PL * a.ppt)), dim3(256), 0, st, a);
    RV_LAUNCH_CHECK("bn backward statistics");
    return RV_OK;
}

extern "C" {

// Bytes of workspace the BN entry points need: RV_BN_NREP replicas of 2*C fp64 sums (common.h).  CONTRACT: the workspace must be ALL-ZERO on entry (the
// host hands out slices of one arena that is cleared once per step -- a per-call memset costs a launch, and a
// last-workgroup self-clean costs ~4 ns of serialized atomics per workgroup); it holds the sums on exit.
long rv_bn_workspace_bytes(int C) { return (long)RV_BN_NREP * (2 * C) * 8; }

// coef: [5C] floats (mean, invstd, scale, shift, unbiased batch variance) -- saved for backward.
// training == 1: batch statistics, running stats / num_batches_tracked updated in place;
// training == 2: batch statistics, running stats untouched (apply them later with rv_bn_running_update);
// training == 0: running statistics (eval mode).
// sums_ready != 0: the workspace already holds this tensor's sums (rv_conv_fwd(..., bn_sums = workspace) produced z).
// y = leaky_relu(bn(z), slope) (+ res).  slope = 1 -> no activation.
int rv_bn_lrelu_fwd(const float* z, int z_ld, long P, int C, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, long* num_batches_tracked, float momentum, float eps, int training, float slope,
                    const float* res, int res_ld, float* y, int y_ld, float* coef, void* workspace, int sums_ready,
                    void* stream) {
    hipStream_t st = (hipStream_t)stream;
    RV_CHECK_ARG(C % 4 == 0 && C <= 128, "rv_bn_lrelu_fwd: C=%d must be a multiple of 4 and <= 128", C);
    RV_CHECK_ARG((z_ld % 4) == 0 && (y_ld % 4) == 0 && (!res || (res_ld % 4) == 0), "rv_bn_lrelu_fwd: strides must be multiples of 4");
    BnArgs a = {};
    a.z = z; a.z_ld = z_ld; a.P = P; a.C = C; a.sums = (double*)workspace; a.slope = slope;
    a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var; a.nbt = num_batches_tracked;
    a.coef_out = coef; a.momentum = momentum; a.eps = eps; a.training = training;
    if (training && !sums_ready) {
        const int rc = rv_internal_bn_stats(z, z_ld, P, C, a.sums, st);
        if (rc != RV_OK) return rc;
    }
    a.out = y; a.out_ld = y_ld; a.res = res; a.res_ld = res_ld;
    hipLaunchKernelGGL(bn_apply_k<false>, dim3(bn_apply_blocks(P * (C / 4))), dim3(256), 0, st, a);
    RV_LAUNCH_CHECK("rv_bn_lrelu_fwd(apply)");
    return RV_OK;
}

// The same with the residual evaluated in place: y = leaky_relu(bn(z), slope) + skip(x), skip = the 1x1 convolution of an encoder block
// (model/UNet_onset.py:191,198: `x12 += self.skip(x)`), x [P][cin] at pixel stride x_ld, w [C][cin] (the PyTorch weight), b [C] nullable.  cin = 1 (block 1, the
// single-channel spectrogram: one fma per element) or 16 / 32 / 64 (blocks 2-4: an fmaf chain per element in the k order of the MFMA conv kernel; ksplit != 0
// reproduces the K-split form of that kernel, algo family 0x5NM).  Results are BIT-IDENTICAL to rv_conv_fwd(mode 1) + rv_bn_lrelu_fwd(res = its output).
int rv_bn_lrelu_fwd_skip(const float* z, int z_ld, long P, int C, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, long* num_batches_tracked, float momentum, float eps, int training, float slope,
                         const float* x, int x_ld, int cin, const float* w, const float* b, int ksplit, float* y, int y_ld, float* coef,
                         void* workspace, int sums_ready, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    RV_CHECK_ARG(C % 4 == 0 && C <= 128, "rv_bn_lrelu_fwd_skip: C=%d must be a multiple of 4 and <= 128", C);
    RV_CHECK_ARG(cin == 1 || cin == 16 || cin == 32 || cin == 64, "rv_bn_lrelu_fwd_skip: cin=%d (1, 16, 32 or 64)", cin);
    RV_CHECK_ARG((z_ld % 4) == 0 && (y_ld % 4) == 0 && x && w && x_ld >= cin && (cin == 1 || ((x_ld % 4) == 0 && (((uintptr_t)x) & 15) == 0)),
                 "rv_bn_lrelu_fwd_skip: bad strides / alignment / null skip operands");
    BnArgs a = {};
    a.z = z; a.z_ld = z_ld; a.P = P; a.C = C; a.sums = (double*)workspace; a.slope = slope;
    a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var; a.nbt = num_batches_tracked;
    a.coef_out = coef; a.momentum = momentum; a.eps = eps; a.training = training;
    if (training && !sums_ready) {
        const int rc = rv_internal_bn_stats(z, z_ld, P, C, a.sums, st);
        if (rc != RV_OK) return rc;
    }
    a.out = y; a.out_ld = y_ld; a.r1_x = x; a.r1_x_ld = x_ld; a.r1_w = w; a.r1_b = b; a.sk_cin = cin; a.sk_ks = ksplit;
    const dim3 grid(bn_apply_blocks(P * (C / 4)));
    const size_t lds = (size_t)cin * C * sizeof(float);
    if (cin == 1) hipLaunchKernelGGL((bn_apply_k<false, 0>), grid, dim3(256), 0, st, a);
    else if (cin == 16) hipLaunchKernelGGL((bn_apply_k<false, 16>), grid, dim3(256), lds, st, a);
    else if (cin == 32) hipLaunchKernelGGL((bn_apply_k<false, 32>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((bn_apply_k<false, 64>), grid, dim3(256), lds, st, a);
    RV_LAUNCH_CHECK("rv_bn_lrelu_fwd_skip(apply)");
    return RV_OK;
}

// dz (and dgamma/dbeta when non-null) from dy; z and coef are the forward's.  frozen != 0: eval-mode BN.
// param_accumulate != 0: dgamma/dbeta are added into their destinations (gradient accumulation).
// sums_ready != 0: the workspace already holds the reduction (rv_conv_fwd(..., bn_sums, bn_z, ...) produced dy).
int rv_bn_lrelu_bwd(const float* dy, int dy_ld, const float* z, int z_ld, long P, int C, const float* coef, float slope,
                    int frozen, float* dz, int dz_ld, float* dgamma, float* dbeta, int param_accumulate, void* workspace,
                    int sums_ready, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    RV_CHECK_ARG(C % 4 == 0 && C <= 128, "rv_bn_lrelu_bwd: C=%d must be a multiple of 4 and <= 128", C);
    BnArgs a = {};
    a.z = z; a.z_ld = z_ld; a.dy = dy; a.dy_ld = dy_ld; a.P = P; a.C = C; a.sums = (double*)workspace;
    a.coef = coef; a.slope = slope; a.frozen = frozen; a.out = dz; a.out_ld = dz_ld; a.dgamma = dgamma; a.dbeta = dbeta; a.param_accumulate = param_accumulate;
    if ((!frozen || dgamma) && !sums_ready) {
        const int rc = rv_internal_bn_bwd_stats(dy, dy_ld, z, z_ld, P, C, coef, slope, a.sums, st);
        if (rc != RV_OK) return rc;
    }
    hipLaunchKernelGGL(bn_apply_k<true>, dim3(bn_apply_blocks(P * (C / 4), true)), dim3(256), 0, st, a);
    RV_LAUNCH_CHECK("rv_bn_lrelu_bwd(apply)");
    return RV_OK;
}

// running_mean/var <- (1-m)*running + m*batch (batch mean / unbiased variance taken from a forward's coef), and
// num_batches_tracked += 1: the update a training == 2 forward skipped, applied where the caller needs it in the
// sequence of updates (nn.BatchNorm2d semantics are order-dependent).
__global__ void bn_running_update_k(float* rm, float* rv, long* nbt, const float* coef, int C, float momentum) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    rm[c] = bn_momentum_update(rm[c], coef[c], momentum);
    rv[c] = bn_momentum_update(rv[c], coef[4 * C + c], momentum);
}

// Table form: every deferred update of a training step in ONE launch.  `table` (device, int64 words) holds, per BatchNorm
// layer, {running_mean, running_var, num_batches_tracked, C, first, count} and then a flat list of coef pointers; workgroup
// l applies layer l's `count` updates in list order (the order matters: momentum updates do not commute).
__global__ __launch_bounds__(128) void bn_running_update_table_k(const long* table, int nlayers, float momentum) {
    const long* e = table + (long)blockIdx.x * 6;
    float* rm = (float*)e[0];
    float* rv = (float*)e[1];
    long* nbt = (long*)e[2];
    const int C = (int)e[3];
    const long first = e[4], count = e[5];
    const long* coefs = table + (long)nlayers * 6 + first;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float m = rm[c], v = rv[c];
        for (long k = 0; k < count; ++k) {
            const float* coef = (const float*)coefs[k];
            m = bn_momentum_update(m, coef[c], momentum);
            v = bn_momentum_update(v, coef[4 * C + c], momentum);
        }
        rm[c] = m; rv[c] = v;
    }
    if (threadIdx.x == 0 && nbt) *nbt += count;
}

int rv_bn_running_update_table(const long* table, int nlayers, float momentum, void* stream) {
    RV_CHECK_ARG(table && nlayers > 0, "rv_bn_running_update_table: empty table");
    hipLaunchKernelGGL(bn_running_update_table_k, dim3(nlayers), dim3(128), 0, (hipStream_t)stream, table, nlayers, momentum);
    RV_LAUNCH_CHECK("rv_bn_running_update_table");
    return RV_OK;
}

int rv_bn_running_update(float* running_mean, float* running_var, long* num_batches_tracked, const float* coef, int C,
                         float momentum, void* stream) {
    hipLaunchKernelGGL(bn_running_update_k, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, running_mean, running_var,
                       num_batches_tracked, coef, C, momentum);
    RV_LAUNCH_CHECK("rv_bn_running_update");
    return RV_OK;
}

}  // extern "C"
