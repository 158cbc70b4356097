// Fused 31-frame local multi-head attention (reference MutliHeadAttention1D.forward,
// model/UNet_onset.py:56-91) for gfx950 -- no [B, L, F, 31] unfold is ever materialised.
//
//   energy[b,t,g,w] = sum_f q[b,t,g,f] * (k[b,t+w-15,g,f] + rel[g*dh+f, w])   (k = 0 outside [0,L): the
//                     reference zero-pads x and W_k has no bias, so padded slots score q.rel -- unmasked)
//   att = softmax_w(energy)          (no 1/sqrt(d) scaling)
//   out[b,t,g,f]    = sum_w att[b,t,g,w] * v[b,t+w-15,g,f]
//
// q, k, v, out: [B, L, F] with F = G*dh contiguous; rel: [F, 31]; att: [B, L, G, 31].
// One workgroup = (batch b, 16-frame tile, head g): the 46-row K (then V) window is staged once in LDS
// (rows padded to dh+1 floats -> conflict-free for both the row-strided and the column-strided phase).
// HBM/L2-bound: ~0.5 GFLOP per forward against ~60 MB of q/k/v/out traffic.
#include "common.h"

#define AT_W 31
#define AT_P 15
#define AT_TT 16
#define AT_WIN (AT_TT + 2 * AT_P)   // 46 rows

struct AttnArgs {
    const float* q; const float* k; const float* v; const float* rel;
    float* out; float* att;
    const float* dout; const float* de_in;
    float* dq; float* dk; float* dv; float* de;
    int B, L, G, dh;
};

__device__ __forceinline__ void load_rows(float* dst, int ldd, const float* src, int F, int col0, int dh, int row0,
                                          int nrows, int L) {
    // dst[r][f] = src[(row0 + r)*F + col0 + f] (0 outside [0, L))
    for (int idx = threadIdx.x; idx < nrows * dh; idx += blockDim.x) {
        int r = idx / dh, f = idx - r * dh;
        int t = row0 + r;
        dst[r * ldd + f] = (t >= 0 && t < L) ? src[(long)t * F + col0 + f] : 0.f;
    }
}

__global__ __launch_bounds__(256) void attn_fwd_k(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int dh = a.dh, ld = dh + 1, F = a.G * dh;
    float* qs = smem;                       // [TT][ld]
    float* ws = qs + AT_TT * ld;            // [WIN][ld]
    float* es = ws + AT_WIN * ld;           // [TT][32]
    const int ntile = (a.L + AT_TT - 1) / AT_TT;
    const int b = blockIdx.x / ntile, t0 = (blockIdx.x - b * ntile) * AT_TT;
    const int g = blockIdx.y;
    const long boff = (long)b * a.L * F;
    load_rows(qs, ld, a.q + boff, F, g * dh, dh, t0, AT_TT, a.L);
    load_rows(ws, ld, a.k + boff, F, g * dh, dh, t0 - AT_P, AT_WIN, a.L);
    __syncthreads();
    const float* rel = a.rel + (long)g * dh * AT_W;
    for (int idx = threadIdx.x; idx < AT_TT * AT_W; idx += blockDim.x) {
        int t = idx / AT_W, w = idx - t * AT_W;
        const float* qr = qs + t * ld;
        const float* kr = ws + (t + w) * ld;
        float e = 0.f;
        for (int f = 0; f < dh; ++f) e = fmaf(qr[f], kr[f] + rel[f * AT_W + w], e);
        es[t * 32 + w] = e;
    }
    __syncthreads();
    if (threadIdx.x < AT_TT) {
        int t = threadIdx.x;
        float mx = -INFINITY;
        for (int w = 0; w < AT_W; ++w) mx = fmaxf(mx, es[t * 32 + w]);
        float s = 0.f;
        for (int w = 0; w < AT_W; ++w) { float e = expf(es[t * 32 + w] - mx); es[t * 32 + w] = e; s += e; }
        for (int w = 0; w < AT_W; ++w) es[t * 32 + w] = es[t * 32 + w] / s;
    }
    // stage V over the K window (everyone is done reading K once the softmax barrier is passed)
    __syncthreads();
    load_rows(ws, ld, a.v + boff, F, g * dh, dh, t0 - AT_P, AT_WIN, a.L);
    if (a.att) {
        for (int idx = threadIdx.x; idx < AT_TT * AT_W; idx += blockDim.x) {
            int t = idx / AT_W, w = idx - t * AT_W;
            if (t0 + t < a.L) a.att[(((long)b * a.L + t0 + t) * a.G + g) * AT_W + w] = es[t * 32 + w];
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < AT_TT * dh; idx += blockDim.x) {
        int t = idx / dh, f = idx - t * dh;
        if (t0 + t >= a.L) continue;
        float o = 0.f;
#pragma unroll
        for (int w = 0; w < AT_W; ++w) o = fmaf(es[t * 32 + w], ws[(t + w) * ld + f], o);
        a.out[boff + (long)(t0 + t) * F + g * dh + f] = o;
    }
}

// backward, query side: de (softmax backward) and dq for a 16-frame tile
__global__ __launch_bounds__(256) void attn_bwd_q_k(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int dh = a.dh, ld = dh + 1, F = a.G * dh;
    float* dos = smem;                      // [TT][ld]
    float* ws = dos + AT_TT * ld;           // [WIN][ld]  V then K
    float* as = ws + AT_WIN * ld;           // [TT][32]   att
    float* ds = as + AT_TT * 32;            // [TT][32]   datt -> de
    const int ntile = (a.L + AT_TT - 1) / AT_TT;
    const int b = blockIdx.x / ntile, t0 = (blockIdx.x - b * ntile) * AT_TT;
    const int g = blockIdx.y;
    const long boff = (long)b * a.L * F;
    load_rows(dos, ld, a.dout + boff, F, g * dh, dh, t0, AT_TT, a.L);
    load_rows(ws, ld, a.v + boff, F, g * dh, dh, t0 - AT_P, AT_WIN, a.L);
    for (int idx = threadIdx.x; idx < AT_TT * AT_W; idx += blockDim.x) {
        int t = idx / AT_W, w = idx - t * AT_W;
        as[t * 32 + w] = (t0 + t < a.L) ? a.att[(((long)b * a.L + t0 + t) * a.G + g) * AT_W + w] : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < AT_TT * AT_W; idx += blockDim.x) {
        int t = idx / AT_W, w = idx - t * AT_W;
        const float* dr = dos + t * ld;
        const float* vr = ws + (t + w) * ld;
        float s = 0.f;
        for (int f = 0; f < dh; ++f) s = fmaf(dr[f], vr[f], s);
        ds[t * 32 + w] = s;
    }
    __syncthreads();
    if (threadIdx.x < AT_TT) {
        int t = threadIdx.x;
        float dot = 0.f;
        for (int w = 0; w < AT_W; ++w) dot = fmaf(as[t * 32 + w], ds[t * 32 + w], dot);
        for (int w = 0; w < AT_W; ++w) ds[t * 32 + w] = as[t * 32 + w] * (ds[t * 32 + w] - dot);
    }
    __syncthreads();
    load_rows(ws, ld, a.k + boff, F, g * dh, dh, t0 - AT_P, AT_WIN, a.L);
    for (int idx = threadIdx.x; idx < AT_TT * AT_W; idx += blockDim.x) {
        int t = idx / AT_W, w = idx - t * AT_W;
        if (t0 + t < a.L) a.de[(((long)b * a.L + t0 + t) * a.G + g) * AT_W + w] = ds[t * 32 + w];
    }
    __syncthreads();
    const float* rel = a.rel + (long)g * dh * AT_W;
    for (int idx = threadIdx.x; idx < AT_TT * dh; idx += blockDim.x) {
        int t = idx / dh, f = idx - t * dh;
        if (t0 + t >= a.L) continue;
        float o = 0.f;
#pragma unroll
        for (int w = 0; w < AT_W; ++w) o = fmaf(ds[t * 32 + w], ws[(t + w) * ld + f] + rel[f * AT_W + w], o);
        a.dq[boff + (long)(t0 + t) * F + g * dh + f] = o;
    }
}

// backward, key/value side for a 16-frame tile of window rows s:
//   dv[s] = sum_w att[s-w+15][w] * dout[s-w+15],   dk[s] = sum_w de[s-w+15][w] * q[s-w+15]
__global__ __launch_bounds__(256) void attn_bwd_kv_k(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int dh = a.dh, ld = dh + 1, F = a.G * dh;
    float* ws = smem;                       // [WIN][ld]  dout then q
    float* as = ws + AT_WIN * ld;           // [WIN][32]  att then de
    const int ntile = (a.L + AT_TT - 1) / AT_TT;
    const int b = blockIdx.x / ntile, s0 = (blockIdx.x - b * ntile) * AT_TT;
    const int g = blockIdx.y;
    const long boff = (long)b * a.L * F;
    for (int pass = 0; pass < 2; ++pass) {
        const float* rows = pass == 0 ? a.dout : a.q;
        const float* coef = pass == 0 ? a.att : a.de_in;
        float* dst = pass == 0 ? a.dv : a.dk;
        if (pass) __syncthreads();
        load_rows(ws, ld, rows + boff, F, g * dh, dh, s0 - AT_P, AT_WIN, a.L);
        for (int idx = threadIdx.x; idx < AT_WIN * AT_W; idx += blockDim.x) {
            int u = idx / AT_W, w = idx - u * AT_W;
            int t = s0 - AT_P + u;
            as[u * 32 + w] = (t >= 0 && t < a.L) ? coef[(((long)b * a.L + t) * a.G + g) * AT_W + w] : 0.f;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < AT_TT * dh; idx += blockDim.x) {
            int sl = idx / dh, f = idx - sl * dh;
            if (s0 + sl >= a.L) continue;
            float o = 0.f;
#pragma unroll
            for (int w = 0; w < AT_W; ++w) {
                int u = sl - w + 2 * AT_P;      // t = s - w + 15  ->  u = t - (s0 - 15)
                o = fmaf(as[u * 32 + w], ws[u * ld + f], o);
            }
            dst[boff + (long)(s0 + sl) * F + g * dh + f] = o;
        }
    }
}

extern "C" {

int rv_local_attn_fwd(const float* q, const float* k, const float* v, const float* rel, float* out, float* att, int B, int L,
                      int G, int dh, void* stream) {
    RV_CHECK_ARG(dh >= 1 && dh <= 256, "rv_local_attn_fwd: head dim %d unsupported", dh);
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.rel = rel; a.out = out; a.att = att; a.B = B; a.L = L; a.G = G; a.dh = dh;
    const int ntile = (L + AT_TT - 1) / AT_TT;
    size_t lds = ((size_t)(AT_TT + AT_WIN) * (dh + 1) + AT_TT * 32) * sizeof(float);
    hipLaunchKernelGGL(attn_fwd_k, dim3(B * ntile, G), dim3(256), lds, (hipStream_t)stream, a);
    RV_LAUNCH_CHECK("rv_local_attn_fwd");
    return RV_OK;
}

// Inputs: dout and the forward's q, k, v, att.  Outputs: dq, dk, dv [B,L,F] and de [B,L,G,31] (the energy
// gradient; drel = sum_{b,t} de (x) q is a plain GEMM the host issues through rv_gemm).
int rv_local_attn_bwd(const float* dout, const float* q, const float* k, const float* v, const float* rel, const float* att,
                      float* dq, float* dk, float* dv, float* de, int B, int L, int G, int dh, void* stream) {
    RV_CHECK_ARG(dh >= 1 && dh <= 256, "rv_local_attn_bwd: head dim %d unsupported", dh);
    hipStream_t st = (hipStream_t)stream;
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.rel = rel; a.att = (float*)att; a.dout = dout; a.dq = dq; a.dk = dk; a.dv = dv; a.de = de;
    a.de_in = de; a.B = B; a.L = L; a.G = G; a.dh = dh;
    const int ntile = (L + AT_TT - 1) / AT_TT;
    size_t lds1 = ((size_t)(AT_TT + AT_WIN) * (dh + 1) + 2 * AT_TT * 32) * sizeof(float);
    hipLaunchKernelGGL(attn_bwd_q_k, dim3(B * ntile, G), dim3(256), lds1, st, a);
    RV_LAUNCH_CHECK("rv_local_attn_bwd(q)");
    size_t lds2 = ((size_t)AT_WIN * (dh + 1) + AT_WIN * 32) * sizeof(float);
    hipLaunchKernelGGL(attn_bwd_kv_k, dim3(B * ntile, G), dim3(256), lds2, st, a);
    RV_LAUNCH_CHECK("rv_local_attn_bwd(kv)");
    return RV_OK;
}

}  // extern "C"
