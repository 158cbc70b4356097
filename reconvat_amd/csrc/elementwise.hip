// VAT perturbation primitives, losses, optimiser and small reductions for gfx950.
// All HBM-bound: one pass per tensor, wavefront (64-lane) reductions over the 229-/88-wide rows.
//
// Reference anchors:
//   _l2_normalize / r = XI*d/||d|| / clamp(0,1) ......... model/UNet_onset.py:130-131,145-151,165-171
//   F.binary_cross_entropy (soft targets, log clamp -100) .. model/UNet_onset.py:136-137,157-158,473-476
//   F.mse_loss ............................................ model/UNet_onset.py:472
//   Adam + StepLR ......................................... train_UNet_Onset_VAT.py:113,124
#include "common.h"

// ---------------------------------------------------------------------------------------------
// VAT: x_adv = clamp(x + scale * (prescale*d) / ||prescale*d||_row, 0, 1)
// ---------------------------------------------------------------------------------------------
struct PerturbArgs {
    const float* x; const float* d; const float* g;   // g: grad wrt x_adv (bwd only)
    float* x_adv; float* r_out; float* dn_out; float* gd;
    long rows; int n;
    float prescale, scale;
    int* nan_flag;
};

template <bool BWD>
__global__ __launch_bounds__(256) void vat_perturb_k(PerturbArgs a) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const float* d = a.d + row * a.n;
    const float* x = a.x + row * a.n;
    float ss = 0.f;
    for (int i = lane; i < a.n; i += 64) { float v = d[i] * a.prescale; ss = fmaf(v, v, ss); }
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    const float inv = 1.0f / nrm;        // 0/0 -> NaN exactly like the reference (no epsilon)
    if (!BWD) {
        bool bad = false;
        for (int i = lane; i < a.n; i += 64) {
            float dn = (d[i] * a.prescale) / nrm;
            float r = a.scale * dn;
            bad |= (r != r);
            float xa = x[i] + r;
            xa = fminf(fmaxf(xa, 0.f), 1.f);
            a.x_adv[row * a.n + i] = xa;
            if (a.r_out) a.r_out[row * a.n + i] = r;
            if (a.dn_out) a.dn_out[row * a.n + i] = dn;
        }
        if (a.nan_flag && __any(bad) && lane == 0) atomicOr(a.nan_flag, 1);
    } else {
        const float* g = a.g + row * a.n;
        float dot = 0.f;
        for (int i = lane; i < a.n; i += 64) {
            float dn = (d[i] * a.prescale) / nrm;
            float xa = x[i] + a.scale * dn;
            float gr = (xa >= 0.f && xa <= 1.f) ? g[i] * a.scale : 0.f;
            dot = fmaf(dn, gr, dot);
        }
        dot = wave_sum(dot);
        for (int i = lane; i < a.n; i += 64) {
            float dn = (d[i] * a.prescale) / nrm;
            float xa = x[i] + a.scale * dn;
            float gr = (xa >= 0.f && xa <= 1.f) ? g[i] * a.scale : 0.f;
            a.gd[row * a.n + i] = (gr - dn * dot) * inv * a.prescale;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// mean-reduced losses: two-stage deterministic reduction (per-block partials, fp64 fold)
// ---------------------------------------------------------------------------------------------
enum { RED_BCE = 0, RED_MSE = 1, RED_ABS = 2, RED_SUMSQ = 3 };

template <int KIND>
__device__ __forceinline__ float red_term(float p, float t) {
    if (KIND == RED_BCE) {
        float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);   // torch clamps log at -100
        return -(t * lp + (1.f - t) * lq);
    } else if (KIND == RED_MSE) {
        float e = p - t;
        return e * e;
    } else if (KIND == RED_ABS) {
        return fabsf(p);
    } else {
        return p * p;
    }
}

// ticket (nullable): a zeroed device word.  With it the LAST workgroup to finish also does the final fold (fixed order, so
// the result does not depend on which workgroup that is) and re-zeroes the ticket: one launch per loss instead of two.
template <int KIND>
__global__ __launch_bounds__(256) void reduce_partial_k(const float* p, const float* t, long n, float* part, unsigned* ticket,
                                                        double denom, float* out, int sqrt_out) {
    __shared__ float sh[4];
    __shared__ double shd[4];
    __shared__ bool last;
    float s = 0.f;
    const long base = (long)blockIdx.x * 2048;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        long i = base + k * 256 + threadIdx.x;
        if (i < n) s += red_term<KIND>(p[i], (KIND == RED_BCE || KIND == RED_MSE) ? t[i] : 0.f);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    if (!ticket) return;
    if (threadIdx.x == 0) {
        __threadfence();                                      // the partial is visible before the ticket is taken
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    double d = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) d += (double)__hip_atomic_load(&part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    d = wave_sum_d(d);
    if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double v = ((shd[0] + shd[1]) + (shd[2] + shd[3])) / denom;
        *out = (float)(sqrt_out ? sqrt(v) : v);
        *ticket = 0u;
    }
}

__global__ __launch_bounds__(256) void reduce_final_k(const float* part, int nparts, double denom, float* out, int sqrt_out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += (double)part[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = ((sh[0] + sh[1]) + (sh[2] + sh[3])) / denom;
        *out = (float)(sqrt_out ? sqrt(v) : v);
    }
}

// grad of the mean-reduced loss wrt its first argument, times the upstream scalar *gout
template <int KIND>
__global__ __launch_bounds__(256) void loss_bwd_k(const float* p, const float* t, long n, const float* gout, float* gp) {
    const float go = *gout / (float)n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float g;
        if (KIND == RED_BCE) g = (p[i] - t[i]) / fmaxf((1.f - p[i]) * p[i], 1e-12f);   // torch's BCE backward
        else g = 2.f * (p[i] - t[i]);
        gp[i] = g * go;
    }
}

// ---------------------------------------------------------------------------------------------
// sigmoid backward with up to two upstream gradients (strided): dz = (g1 + g2) * y * (1 - y)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sigmoid_bwd_k(const float* g1, int ld1, const float* g2, int ld2, const float* y,
                                                     int ldy, float* dz, int ldz, long M, int N) {
    const long total = M * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long m = i / N; int n = (int)(i - m * N);
        float g = 0.f;
        if (g1) g += g1[m * ld1 + n];
        if (g2) g += g2[m * ld2 + n];
        float yy = y[m * ldy + n];
        dz[m * ldz + n] = g * yy * (1.f - yy);
    }
}

// column sums of a [M, N] matrix (bias gradients): out[n] (+)= sum_m x[m*ld + n]
// generic version: 64 columns x 4 row lanes per workgroup
__global__ __launch_bounds__(256) void colsum_k(const float* x, int ld, long M, int N, float* out, int rows_per_block) {
    __shared__ float sh[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(r0 + rows_per_block, M);
    float s = 0.f;
    if (col < N)
        for (long r = r0 + rl; r < r1; r += 4) s += x[r * ld + col];
    sh[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && col < N) atomicAdd(&out[col], (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]));
}

// narrow matrices (N % 4 == 0, N <= 256; the up-conv bias gradients over millions of pixels): 16-byte loads,
// thread -> (column quad, row lane), 32 rows per thread, LDS fold, one atomic per column per workgroup
__global__ __launch_bounds__(256) void colsum_vec_k(const float* x, int ld, long M, int N, float* out, int rpt) {
    __shared__ float sh[256 * 4];
    const int C4 = N >> 2, PL = 256 / C4;
    const int t = threadIdx.x;
    const int c = (t % C4) * 4, pl = t / C4;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (pl < PL) {
        const long p0 = (long)blockIdx.x * PL * rpt;
#pragma unroll 4
        for (int k = 0; k < rpt; ++k) {
            const long p = p0 + pl + (long)k * PL;
            if (p >= M) break;
            s += *reinterpret_cast<const f32x4*>(x + p * ld + c);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) sh[pl * N + c + q] = s[q];
    }
    __syncthreads();
    if (t < N) {
        float d = 0.f;
        for (int l = 0; l < PL; ++l) d += sh[l * N + t];
        atomicAdd(&out[t], d);
    }
}

// ---------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults) + StepLR on one flat parameter buffer
// ---------------------------------------------------------------------------------------------
struct AdamArgs {
    float* p; const float* g; float* m; float* v; long n;
    const long* step;      // number of optimiser steps already taken (device scalar)
    float lr0, decay_rate; long decay_steps;
    float b1, b2, eps, grad_scale;
    const int* skip;       // nullable: a non-zero word means "this step's gradients are invalid" -> leave p, m, v untouched
};

__global__ __launch_bounds__(256) void adam_k(AdamArgs a) {
    if (a.skip && *a.skip != 0) return;
    const long step = *a.step;
    const double t = (double)(step + 1);
    const float lr = a.lr0 * (float)pow((double)a.decay_rate, (double)(step / a.decay_steps));
    const float bc1 = (float)(1.0 - pow((double)a.b1, t));
    const float bc2s = (float)sqrt(1.0 - pow((double)a.b2, t));
    const float step_size = lr / bc1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long)gridDim.x * blockDim.x) {
        float g = a.g[i] * a.grad_scale;
        float m = a.m[i] * a.b1 + (1.f - a.b1) * g;
        float v = a.v[i] * a.b2 + (1.f - a.b2) * g * g;
        a.m[i] = m; a.v[i] = v;
        float denom = sqrtf(v) / bc2s + a.eps;
        a.p[i] -= step_size * (m / denom);
    }
}

__global__ void counter_add_k(long* c, long inc, const int* skip) { if (!skip || *skip == 0) *c += inc; }

__global__ __launch_bounds__(256) void scale_by_clip_k(float* g, long n, const float* total_norm, float max_norm) {
    // torch.nn.utils.clip_grad_norm_: coef = clamp(max_norm / (total_norm + 1e-6), max=1)
    float coef = fminf(max_norm / (*total_norm + 1e-6f), 1.0f);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) g[i] *= coef;
}

static int grid_for(long n) { long b = (n + 255) / 256; return (int)(b < 4096 ? (b > 0 ? b : 1) : 4096); }

extern "C" {

// x_adv = clamp(x + scale*normalise_rows(prescale*d), 0, 1); optional r (=scale*dn) and dn outputs;
// nan_flag (device int, may be null) is OR-ed with 1 when any r is NaN (the reference's assert).
int rv_vat_perturb_fwd(const float* x, const float* d, long rows, int n, float prescale, float scale, float* x_adv,
                       float* r_out, float* dn_out, int* nan_flag, void* stream) {
    PerturbArgs a = {};
    a.x = x; a.d = d; a.x_adv = x_adv; a.r_out = r_out; a.dn_out = dn_out; a.rows = rows; a.n = n;
    a.prescale = prescale; a.scale = scale; a.nan_flag = nan_flag;
    hipLaunchKernelGGL(vat_perturb_k<false>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, a);
    RV_LAUNCH_CHECK("rv_vat_perturb_fwd");
    return RV_OK;
}

// gd = d(loss)/d(d) given g = d(loss)/d(x_adv): through clamp, the scale and the row L2-normalisation.
int rv_vat_perturb_bwd(const float* g, const float* x, const float* d, long rows, int n, float prescale, float scale,
                       float* gd, void* stream) {
    PerturbArgs a = {};
    a.x = x; a.d = d; a.g = g; a.gd = gd; a.rows = rows; a.n = n; a.prescale = prescale; a.scale = scale;
    hipLaunchKernelGGL(vat_perturb_k<true>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, a);
    RV_LAUNCH_CHECK("rv_vat_perturb_bwd");
    return RV_OK;
}

long rv_reduce_workspace_bytes(long n) { return ((n + 2047) / 2048) * 4; }

// kind: 0 = BCE mean (p vs soft/hard target t), 1 = MSE mean, 2 = mean |p|, 3 = sqrt(sum p^2) (L2 norm)
// ticket (nullable): a ZEROED device word (left zero again): single-launch form, see reduce_partial_k
int rv_reduce_mean(int kind, const float* p, const float* t, long n, float* out, void* workspace, unsigned* ticket, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int nparts = cdiv(n, 2048);
    float* part = (float*)workspace;
    const double denom = kind == RED_SUMSQ ? 1.0 : (double)n;
    const int sq = kind == RED_SUMSQ ? 1 : 0;
    switch (kind) {
        case RED_BCE: hipLaunchKernelGGL(reduce_partial_k<RED_BCE>, dim3(nparts), dim3(256), 0, st, p, t, n, part, ticket, denom, out, sq); break;
        case RED_MSE: hipLaunchKernelGGL(reduce_partial_k<RED_MSE>, dim3(nparts), dim3(256), 0, st, p, t, n, part, ticket, denom, out, sq); break;
        case RED_ABS: hipLaunchKernelGGL(reduce_partial_k<RED_ABS>, dim3(nparts), dim3(256), 0, st, p, t, n, part, ticket, denom, out, sq); break;
        case RED_SUMSQ: hipLaunchKernelGGL(reduce_partial_k<RED_SUMSQ>, dim3(nparts), dim3(256), 0, st, p, t, n, part, ticket, denom, out, sq); break;
        default: rv_set_error("rv_reduce_mean: bad kind %d", kind); return RV_EINVAL;
    }
    RV_LAUNCH_CHECK("rv_reduce_mean(partial)");
    if (ticket) return RV_OK;
    hipLaunchKernelGGL(reduce_final_k, dim3(1), dim3(256), 0, st, part, nparts, denom, out, sq);
    RV_LAUNCH_CHECK("rv_reduce_mean(final)");
    return RV_OK;
}

// gp = *gout * d(mean loss)/dp ; kind 0 = BCE, 1 = MSE
int rv_loss_bwd(int kind, const float* p, const float* t, long n, const float* gout, float* gp, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (kind == RED_BCE) hipLaunchKernelGGL(loss_bwd_k<RED_BCE>, dim3(grid_for(n)), dim3(256), 0, st, p, t, n, gout, gp);
    else if (kind == RED_MSE) hipLaunchKernelGGL(loss_bwd_k<RED_MSE>, dim3(grid_for(n)), dim3(256), 0, st, p, t, n, gout, gp);
    else { rv_set_error("rv_loss_bwd: bad kind %d", kind); return RV_EINVAL; }
    RV_LAUNCH_CHECK("rv_loss_bwd");
    return RV_OK;
}

int rv_sigmoid_bwd(const float* g1, int ld1, const float* g2, int ld2, const float* y, int ldy, float* dz, int ldz, long M,
                   int N, void* stream) {
    hipLaunchKernelGGL(sigmoid_bwd_k, dim3(grid_for(M * N)), dim3(256), 0, (hipStream_t)stream, g1, ld1, g2, ld2, y, ldy, dz,
                       ldz, M, N);
    RV_LAUNCH_CHECK("rv_sigmoid_bwd");
    return RV_OK;
}

__global__ void zero_floats_k(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

// out[n] = sum_m x[m*ld + n]   (out is overwritten unless accumulate)
// out[i] (+)= sum over the RV_BN_NREP replicas of sums[rep][c0 + i], i < n, where `sums` is a BatchNorm statistics workspace of C channels
// ([rep][2][C] fp64) that a conv's fused epilogue filled (rv_conv_fwd bn_sums): the per-channel sums of what the conv stored ARE the
// column sums of its output -- the bias gradient of the layer that consumes that output as dY, without another pass over it.
__global__ void sums_fold_k(const double* sums, int C, int c0, int n, float* out, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < RV_BN_NREP; ++r) s += sums[(long)r * 2 * C + c0 + i];
    out[i] = accumulate ? out[i] + (float)s : (float)s;
}

int rv_sums_fold(const double* sums, int C, int c0, int n, float* out, int accumulate, void* stream) {
    RV_CHECK_ARG(sums && out && n > 0 && c0 >= 0 && c0 + n <= C, "rv_sums_fold: bad channel range");
    hipLaunchKernelGGL(sums_fold_k, dim3(cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, sums, C, c0, n, out, accumulate);
    RV_LAUNCH_CHECK("rv_sums_fold");
    return RV_OK;
}

int rv_colsum(const float* x, int ld, long M, int N, float* out, int accumulate, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) hipLaunchKernelGGL(zero_floats_k, dim3(cdiv(N, 256)), dim3(256), 0, st, out, N);   // (a kernel: memset nodes proved unreliable across hipGraph replays)
    if ((N & 3) == 0 && N <= 256 && (ld & 3) == 0 && ((((uintptr_t)x) & 15) == 0) && M >= 1024) {
        // rows per thread: ~512 workgroups -- enough to fill the chip on the 5 120-row linear layers (32 rows per thread
        // left 15 workgroups there), few enough that the per-column atomics (serialised, ~23 ns each) stay a short tail
        const int PL = 256 / (N / 4);
        long rpt = M / ((long)PL * 512);
        if (rpt < 4) rpt = 4;
        if (rpt > 64) rpt = 64;
        hipLaunchKernelGGL(colsum_vec_k, dim3(cdiv(M, (long)PL * rpt)), dim3(256), 0, st, x, ld, M, N, out, (int)rpt);
    } else {
        int rpb = 256;
        dim3 grid(cdiv(N, 64), cdiv(M, rpb));
        hipLaunchKernelGGL(colsum_k, grid, dim3(256), 0, st, x, ld, M, N, out, rpb);
    }
    RV_LAUNCH_CHECK("rv_colsum");
    return RV_OK;
}

// Deterministic form of rv_colsum (RV_DETERMINISTIC=1 of the host side): no atomics.  Pass 1: row block b leaves its per-column sums in
// ws[b][N] (plain stores, a fixed row order inside the block); pass 2: one thread per column folds the blocks in index order and
// writes / accumulates.  ws: rv_colsum_ordered_workspace_bytes(M, N), uninitialised.
__global__ __launch_bounds__(256) void colsum_part_k(const float* x, int ld, long M, int N, float* ws, long rows_per_block) {
    __shared__ float sh[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(r0 + rows_per_block, M);
    float s = 0.f;
    if (col < N)
        for (long r = r0 + rl; r < r1; r += 4) s += x[r * ld + col];
    sh[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && col < N) ws[(long)blockIdx.y * N + col] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void colsum_fold_k(const float* ws, int nblk, int N, float* out, int accumulate) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= N) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += ws[(long)b * N + col];
    out[col] = accumulate ? out[col] + s : s;
}
static long colsum_ordered_blocks(long M) {
    long nblk = (M + 255) / 256;
    return nblk > 1024 ? 1024 : (nblk < 1 ? 1 : nblk);
}
long rv_colsum_ordered_workspace_bytes(long M, int N) { return colsum_ordered_blocks(M) * (long)N * 4; }
int rv_colsum_ordered(const float* x, int ld, long M, int N, float* out, int accumulate, void* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    RV_CHECK_ARG(ws, "rv_colsum_ordered: null workspace");
    const long nblk = colsum_ordered_blocks(M);
    const long rpb = (M + nblk - 1) / nblk;
    hipLaunchKernelGGL(colsum_part_k, dim3(cdiv(N, 64), (unsigned)nblk), dim3(256), 0, st, x, ld, M, N, (float*)ws, rpb);
    hipLaunchKernelGGL(colsum_fold_k, dim3(cdiv(N, 256)), dim3(256), 0, st, (const float*)ws, (int)nblk, N, out, accumulate);
    RV_LAUNCH_CHECK("rv_colsum_ordered");
    return RV_OK;
}

// One Adam step on flat buffers; lr = lr0 * decay_rate^(step // decay_steps) (StepLR); *step is NOT modified
// (call rv_counter_add afterwards, with the same skip word).  grad_scale multiplies g on the fly (1/world_size after an all-reduce sum).
int rv_adam_step(float* p, const float* g, float* m, float* v, long n, const long* step, float lr0, long decay_steps,
                 float decay_rate, float beta1, float beta2, float eps, float grad_scale, const int* skip, void* stream) {
    AdamArgs a;
    a.skip = skip;
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.step = step; a.lr0 = lr0; a.decay_steps = decay_steps;
    a.decay_rate = decay_rate; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.grad_scale = grad_scale;
    hipLaunchKernelGGL(adam_k, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a);
    RV_LAUNCH_CHECK("rv_adam_step");
    return RV_OK;
}

// skip (nullable): the per-device "this step is invalid" word -- while it is set the counter does not move either, so a step
// whose update rv_adam_step skipped advances neither StepLR nor the bias correction.
int rv_counter_add(long* counter, long inc, const int* skip, void* stream) {
    hipLaunchKernelGGL(counter_add_k, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, inc, skip);
    RV_LAUNCH_CHECK("rv_counter_add");
    return RV_OK;
}

// g *= min(1, max_norm / (*total_norm + 1e-6))  -- the reference's post-step clip_grad_norm_
int rv_clip_scale(float* g, long n, const float* total_norm, float max_norm, void* stream) {
    hipLaunchKernelGGL(scale_by_clip_k, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, n, total_norm, max_norm);
    RV_LAUNCH_CHECK("rv_clip_scale");
    return RV_OK;
}

}  // extern "C"
