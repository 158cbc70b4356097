// Convolution family for the ReconVAT U-Nets on gfx950 (CDNA4), fp32 in / fp32 accumulate.
//
// All activations are NHWC fp32 with an explicit pixel stride ("ld", floats per pixel) so that a
// tensor may be a channel slice of a wider buffer (the decoder's concat buffers).
//
// Everything 3x3 / 1x1 / 2x2-stride-2 in the U-Net (reference model/UNet_onset.py:173-224) is one
// of two implicit GEMMs on v_mfma_f32_16x16x4_f32 (exact f32, 157 TF/s peak):
//
//   gather : out[b,oy,ox,co] = bias[co] + sum_{ky,kx,ci} in[b, oy*S-P+ky, ox*S-P+kx, ci] * Wm[ky,kx,ci,co]
//            Conv2d 3x3 (fwd, dgrad), ConvTranspose2d 3x3 s1 p1 (fwd, dgrad), 1x1 skip (fwd, dgrad),
//            2x2/s2 down-conv fwd, 2x2/s2 up-conv dgrad.
//   scatter: out[b,2iy+ky,2ix+kx,co] = bias[co] + sum_ci in[b,iy,ix,ci] * Wm[ci,(ky,kx,co)]
//            ConvTranspose2d 2x2/s2 fwd (with output_size -> bias-only padding row/col,
//            model/UNet_onset.py:212-219) and down-conv dgrad.
//
// MFMA roles: A operand = weights  A[i=cout][k=cin],  B operand = activations B[k=cin][j=pixel],
// so the accumulator lane (j = lane&15, g = lane>>4) holds 4 CONSECUTIVE output channels
// (4g..4g+3) of ONE pixel -> one 16-byte store per lane, whole-wave stores are contiguous.
// Activation fragments are read straight from global/L2 as 16-byte (R=4) or 8-byte (R=2) vectors:
// lane (j,g) reads channels [c*4R + g*R, +R) of pixel j; MFMA r uses channel g*R+r as k=g.
// Weight fragments are pre-packed in exactly this lane order (rv_pack_weights) so a wave reads one
// contiguous 1 KiB line per fragment.  M is the flattened output-pixel index (no per-row tile waste).
//
// Weight gradients are pixel-reduction GEMMs (wgrad_mfma): per-wave partial sums are written to a
// workspace and folded by a deterministic second pass (no atomics).
#include "conv_shared.h"
#include <mutex>

// ------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------
struct PackArgs {
    const float* w;
    float* out;
    int taps, kdim, ndim;     // logical Wm[tap][k][n]
    int R, ntile_n, nchunk;   // fragment geometry
    long s_k, s_n;            // element strides of k and n in the PyTorch tensor (tap stride is 1)
    int flip;                 // spatial flip of the tap index
    int scatter_cmid;         // >0: n = tap4*cmid + c, taps==1, source = k*s_k + c*s_n + tap4
    long total;
    long wino0;               // >0: elements [wino0, total) are the Winograd F(2x2,3x3) section U = G g G^T (16 "taps", same fragment order)
};

// U[xi = 4a + b][k][n] = sum_{ky,kx} G[a][ky] G[b][kx] g[ky][kx][k][n],  G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]
__device__ __forceinline__ void pack_wino_elem(const PackArgs& a, long idx) {
    const int R = a.R;
    int r = (int)(idx % R);
    long t = idx / R;
    int lane = (int)(t % 64); t /= 64;
    int nt = (int)(t % a.ntile_n); t /= a.ntile_n;
    int c = (int)(t % a.nchunk);
    int xi = (int)(t / a.nchunk);
    int k = c * 4 * R + (lane >> 4) * R + r;
    int n = nt * 16 + (lane & 15);
    float v = 0.f;
    if (k < a.kdim && n < a.ndim) {
        const float* w = a.w + (long)k * a.s_k + (long)n * a.s_n;
        const int ra = xi >> 2, cb = xi & 3;
        float col[3];                                    // (g G^T)[ky][cb]
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            float g0 = w[a.flip ? 8 - (ky * 3 + 0) : ky * 3 + 0], g1 = w[a.flip ? 8 - (ky * 3 + 1) : ky * 3 + 1],
                  g2 = w[a.flip ? 8 - (ky * 3 + 2) : ky * 3 + 2];
            col[ky] = cb == 0 ? g0 : (cb == 1 ? 0.5f * (g0 + g1 + g2) : (cb == 2 ? 0.5f * (g0 - g1 + g2) : g2));
        }
        v = ra == 0 ? col[0] : (ra == 1 ? 0.5f * (col[0] + col[1] + col[2]) : (ra == 2 ? 0.5f * (col[0] - col[1] + col[2]) : col[2]));
    }
    a.out[a.wino0 + idx] = v;
}

__device__ __forceinline__ void pack_frag_elem(const PackArgs& a, long idx) {
    if (a.wino0 > 0 && idx >= a.wino0) { pack_wino_elem(a, idx - a.wino0); return; }
    const int R = a.R;
    int r = (int)(idx % R);
    long t = idx / R;
    int lane = (int)(t % 64); t /= 64;
    int nt = (int)(t % a.ntile_n); t /= a.ntile_n;
    int c = (int)(t % a.nchunk);
    int tap = (int)(t / a.nchunk);
    int k = c * 4 * R + (lane >> 4) * R + r;
    int n = nt * 16 + (lane & 15);
    float v = 0.f;
    if (k < a.kdim && n < a.ndim) {
        if (a.scatter_cmid > 0) {
            int tap4 = n / a.scatter_cmid, cc = n % a.scatter_cmid;
            v = a.w[(long)k * a.s_k + (long)cc * a.s_n + tap4];
        } else {
            int tt = a.flip ? (a.taps - 1 - tap) : tap;
            v = a.w[(long)k * a.s_k + (long)n * a.s_n + tt];
        }
    }
    a.out[idx] = v;
}

__global__ void pack_frag_k(PackArgs a) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < a.total) pack_frag_elem(a, idx);
}

// plain [tap][k][n] copy for the small-channel VALU kernels
__device__ __forceinline__ void pack_plain_elem(const PackArgs& a, long idx) {
    int n = (int)(idx % a.ndim);
    long t = idx / a.ndim;
    int k = (int)(t % a.kdim);
    int tap = (int)(t / a.kdim);
    int tt = a.flip ? (a.taps - 1 - tap) : tap;
    a.out[idx] = a.w[(long)k * a.s_k + (long)n * a.s_n + tt];
}

__global__ void pack_plain_k(PackArgs a) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < a.total) pack_plain_elem(a, idx);
}

// Every weight of the model in ONE launch (the weights change once per optimiser step, and ~100 separate 5 us pack
// launches per step cost more than the packing itself).  tab[e].block0 = first workgroup of entry e (prefix sum);
// a workgroup finds its entry by bisection.
struct PackEntry {
    PackArgs a;
    long block0;
};
__global__ __launch_bounds__(256) void pack_table_k(const PackEntry* tab, int count) {
    __shared__ PackEntry ent;
    if (threadIdx.x == 0) {
        int lo = 0, hi = count - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tab[mid].block0 <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        ent = tab[lo];
    }
    __syncthreads();
    const long idx = ((long)blockIdx.x - ent.block0) * 256 + threadIdx.x;
    if (idx >= ent.a.total) return;
    if (ent.a.R == 0) pack_plain_elem(ent.a, idx);
    else pack_frag_elem(ent.a, idx);
}

// ------------------------------------------------------------------------------------------
// implicit-GEMM convolution on MFMA
// ------------------------------------------------------------------------------------------

// KS = 4 (family 0x5NM): the four waves of a workgroup share ONE set of MT x NT tiles and split the K loop (taps x channel chunks)
// four ways; the partial accumulators are folded through LDS and the (m, n) tiles are dealt round-robin to the waves for the
// epilogue.  The deep 2x2 / 1x1 layers (K = 256 .. 512, only a few hundred pixel tiles) are otherwise ONE dependent chain of up to
// 32 [global load -> MFMA] steps per wave at about one wave per SIMD: latency-bound at 17-25 us for < 1 GFLOP.
template <int KH, int KW, int S, int P, int R, int NT, int MT, bool SCATTER, int KS = 1>
__global__ __launch_bounds__(256) void conv_mfma_k(ConvArgs a) {
    typedef typename VecR<R>::T vec;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int bx = xcd_remap(blockIdx.x, gridDim.x);
    const int nt0 = blockIdx.y * NT;
    const long tile0 = KS == 1 ? ((long)bx * 4 + wave) * MT : (long)bx * MT;
    if (tile0 * 16 >= a.npix) return;

    int py[MT], px[MT], pb[MT];
    long ibase[MT];
    bool pv[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        long p = (tile0 + m) * 16 + j;
        pv[m] = p < a.npix;
        const unsigned pp = pv[m] ? (unsigned)p : 0u;
        const int b = (int)fastdiv(pp, a.fd_plane);
        const unsigned rem = pp - (unsigned)b * (unsigned)(a.Ph * a.Pw);
        py[m] = (int)fastdiv(rem, a.fd_pw);
        px[m] = (int)rem - py[m] * a.Pw;
        pb[m] = b;
        ibase[m] = (((long)b * a.H + py[m] * S - P) * a.W + px[m] * S - P) * a.in_ld + g * R;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* wp = a.wpack + ((long)nt0 * 64 + lane) * R;
    const long wstep = (long)a.ntile_n * 64 * R;
    const int nchunk = a.nchunk;
    const int nsteps = KH * KW * nchunk;
    const int s_lo = KS == 1 ? 0 : (wave * nsteps) / KS, s_hi = KS == 1 ? nsteps : ((wave + 1) * nsteps) / KS;

    // loader state (one tap ahead of the MFMAs at most)
    int l_tap = s_lo / nchunk, l_c = s_lo - l_tap * nchunk;
    bool ok[MT];
    long toff = 0;
    auto set_tap = [&](int tap) {
        const int ky = tap / KW, kx = tap - ky * KW;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            int iy = py[m] * S - P + ky, ix = px[m] * S - P + kx;
            ok[m] = pv[m] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        }
        toff = ((long)ky * a.W + kx) * a.in_ld;
    };
    auto load = [&](vec (&wf)[NT], vec (&xf)[MT], int s) {
#pragma unroll
        for (int n = 0; n < NT; ++n) wf[n] = *reinterpret_cast<const vec*>(wp + s * wstep + (long)n * 64 * R);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            vec v;
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = 0.f;
            if (ok[m]) v = *reinterpret_cast<const vec*>(a.in + ibase[m] + toff + (long)l_c * 4 * R);
            xf[m] = v;
        }
    };

    vec wcur[NT], xcur[MT], wnxt[NT], xnxt[MT];
    set_tap(l_tap);
    if (s_lo < s_hi) load(wcur, xcur, s_lo);
    for (int s = s_lo; s < s_hi; ++s) {
        if (s + 1 < s_hi) {
            if (++l_c == nchunk) { l_c = 0; ++l_tap; set_tap(l_tap); }
            load(wnxt, xnxt, s + 1);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[n][r], xcur[m][r], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n) wcur[n] = wnxt[n];
#pragma unroll
        for (int m = 0; m < MT; ++m) xcur[m] = xnxt[m];
    }

    if constexpr (KS > 1) {
        // fold the K slices: every wave parks its MT x NT accumulators, then tile q = m * NT + n belongs to wave q % KS
        __shared__ f32x4 part[KS][MT * NT][64];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) part[wave][m * NT + n][lane] = acc[m][n];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int q = m * NT + n;
                if (q % KS != wave) continue;
                f32x4 sum = part[0][q][lane];
#pragma unroll
                for (int w = 1; w < KS; ++w) sum += part[w][q][lane];
                acc[m][n] = sum;
            }
    }
    // epilogue: lane holds channels co0..co0+3 of pixel j of every (m, n) tile
    f32x4 bv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int ncol = (nt0 + n) * 16 + 4 * g;
        const int cb = SCATTER ? ncol % a.Cout : ncol;
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[n][r] = (a.bias && cb + r < a.Cout) ? a.bias[cb + r] : 0.f;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (!pv[m]) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if (KS > 1 && (m * NT + n) % KS != wave) continue;
            const int ncol = (nt0 + n) * 16 + 4 * g;      // GEMM row index (cout')
            int co0;
            long opix;
            bool inside = true;
            if (SCATTER) {
                const int tap4 = ncol / a.Cout;
                co0 = ncol - tap4 * a.Cout;
                const int oy = py[m] * 2 + (tap4 >> 1), ox = px[m] * 2 + (tap4 & 1);
                inside = tap4 < 4 && oy < a.Ho && ox < a.Wo;
                opix = ((long)pb[m] * a.Ho + oy) * a.Wo + ox;
            } else {
                co0 = ncol;
                opix = (tile0 + m) * 16 + j;
            }
            if (!inside || co0 >= a.Cout) continue;
            float* o = a.out + opix * a.out_ld + co0;
            f32x4 v = acc[m][n] + bv[n];
            if (a.vec_store && co0 + 3 < a.Cout) {
                if (a.accumulate) {
                    f32x4 old = *reinterpret_cast<f32x4*>(o);
                    v += old;
                }
                *reinterpret_cast<f32x4*>(o) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co0 + r < a.Cout) o[r] = a.accumulate ? o[r] + v[r] : v[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// LDS-resident, DMA-pipelined 3x3 (stride 1, pad 1) implicit GEMM: forward and input-gradient of every
// 3x3 Conv2d / ConvTranspose2d with Cin % 8 == 0.
//
// A persistent workgroup walks a run of vertically adjacent bands (TH output rows x full width) for NT
// n-tiles of output channels.  Unit of work = (band, 4R-channel chunk of Cin): its TH+2 input rows (one
// pixel of zero halo left/right) and its 9*NT packed weight fragments sit in LDS, so the tap loop is
// ds_read_b128 + MFMA only and every input element is fetched once per band instead of once per tap.
// Units are DOUBLE-BUFFERED in LDS and filled by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip,
// no ds_write pass): unit u+1 is in flight while unit u is multiplied; one s_waitcnt vmcnt(0) + barrier
// per unit.  M-tiles are 16 consecutive pixels of the flattened band (no waste on the odd widths);
// tile t belongs to wave t % 4.  Accumulator layout / epilogue as conv_mfma_k.
// ------------------------------------------------------------------------------------------
// Timing experiments only (tools/bench_conv.py with RV_ABLATE=bits; results are wrong by design): compile with
// -DRV_ABLATION to honour ConvLdsArgs::ablate.  Bits: 1 barrier, 2 staging, 4 MFMAs, 8 stores, 16 LDS reads,
// 32/64/128 early returns (launch floor / prologue / before the first DMA).
// 1 KiB of zeros in global memory: conv3x3_lds_k's DMA source for rows above / below the image (no row branch around a store loop).
// The Winograd kernels stage through buffer resources instead, where out-of-range offsets read as zeros (rv_buf_lds16 above).
__device__ __attribute__((aligned(16))) const float rv_zero_piece[256] = {0.f};

// BF (R == 4 only; opt-in experiment, BASELINE config 3): the operands are rounded to bf16 (round-to-nearest-even,
// v_cvt_pk_bf16_f32) on their way from LDS to the matrix pipe and ONE v_mfma_f32_16x16x16_bf16 replaces the four f32 MFMAs of a
// 16-channel chunk (a lane's four channels 4g .. 4g+3 are exactly the four consecutive k the bf16 MFMA wants from lane (j, g));
// accumulation, bias, statistics and stores stay fp32.  Everything else -- staging, layouts, epilogue -- is shared.
// (compiler-visible conversions, not inline asm: the VALU-write -> MFMA-read hazard slots are the compiler's to insert)
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 to_bf16x4(const f32x4& v) {
    const bf16x2_t lo = __builtin_convertvector((f32x2){v[0], v[1]}, bf16x2_t);      // v_cvt_pk_bf16_f32 (round to nearest even)
    const bf16x2_t hi = __builtin_convertvector((f32x2){v[2], v[3]}, bf16x2_t);
    const bf16x4_t q = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
    return __builtin_bit_cast(s16x4, q);
}

template <int R, int NT, int MTW, int NW, bool BF = false>
__global__ __launch_bounds__(NW * 64) void conv3x3_lds_k(ConvLdsArgs aa) {
    static_assert(!BF || R == 4, "the bf16 path needs 16-channel chunks");
    constexpr int NTHR = NW * 64;
    typedef typename VecR<R>::T vec;
    constexpr int KC = 4 * R;                    // channels per chunk
    constexpr int Q = R;                         // float4 per staged pixel (KC/4)
    constexpr int PPI = 64 / Q;                  // pixels per DMA wave-instruction
    constexpr int LPF = 16 * R;                  // lanes (float4) per weight fragment
    constexpr int FPI = 64 / LPF;                // fragments per DMA wave-instruction
    constexpr int WFLOATS = 9 * NT * 64 * R;
    const ConvArgs& a = aa.c;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int W = a.W, H = a.H, W2 = W + 2, TH = aa.TH;
    const int nrow = TH + 2;
    const int xfloats = nrow * W2 * KC;
    // XCD-aware placement (1-D grid): every XCD gets a contiguous run of (band group, n-split) pairs with the
    // splits fastest, so the workgroups that re-read the same input rows -- the n-splits of one band group and
    // the neighbouring band groups (halo rows) -- share one L2 instead of pulling the rows into several.
    if (ABL(aa) & 32) return;
    const int vid = aa.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int grp = vid / aa.nsplit, split = vid - grp * aa.nsplit;
    const int nt0 = split * NT;
    const int nbuf = aa.nbuf;
    float* xs0 = smem;                                   // [nbuf][nrow][W2][KC]
    float* ws0 = smem + nbuf * xfloats;                  // [nbuf][9][NT][64][R]
    const int band_lo = grp * aa.bands_per_wg;
    const int band_hi = min(band_lo + aa.bands_per_wg, aa.total_bands);
    if (band_lo >= band_hi) return;
    const int nchunk = a.nchunk;
    const int nunits = (band_hi - band_lo) * nchunk;
    const int ipr = (W + PPI - 1) / PPI;                 // DMA instructions per input row

    // ---- staging plan.  Staging is instruction-issue bound and does not hide under the other waves' MFMAs, so
    // everything that does not depend on the unit is computed ONCE per wave: per DMA slot the lane's byte offset
    // from the unit's first input row / from the chunk's first weight fragment, the LDS destination and the lane
    // mask.  Per unit that leaves two scalar base pointers and one DMA instruction per slot. ----
    constexpr int TXF = 4;                                // x slots planned in registers (any further: generic loop)
    constexpr int NWF = (9 * NT + FPI - 1) / FPI;         // weight DMA instructions per unit (whole workgroup)
    constexpr int TW = (NWF + NW - 1) / NW;               // ... per wave
    const int nx = nrow * ipr;
    int xs_row[TXF], xs_ldst[TXF];
    unsigned xs_goff[TXF];
    bool xs_lane[TXF];
#pragma unroll
    for (int t = 0; t < TXF; ++t) {
        const int i = wave + NW * t;
        const int row = i / ipr, k = i - row * ipr;
        const int px = k * PPI + lane / Q, q = lane - (lane / Q) * Q;
        xs_row[t] = row;
        xs_ldst[t] = (row * W2 + 1 + k * PPI) * KC;
        xs_goff[t] = (unsigned)((row * W + px) * a.in_ld + q * 4) * 4u;
        xs_lane[t] = i < nx && px < W;
    }
    unsigned w_off[TW];
    bool w_lane[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        const int i = wave + NW * t;
        const int f = i * FPI + lane / LPF, o = lane - (lane / LPF) * LPF;
        const int tap = f / NT, n = f - tap * NT;
        w_lane[t] = i < NWF && f < 9 * NT;
        w_off[t] = (unsigned)((tap * nchunk * a.ntile_n + nt0 + n) * 64 * R + o * 4) * 4u;
    }
    const char* zsrc = reinterpret_cast<const char*>(rv_zero_piece) + lane * 16;
    const int b_first = band_lo / aa.nbands, y_first = (band_lo - b_first * aa.nbands) * TH;
    int sg_u = 0, sg_buf = 0, sg_b = b_first, sg_y0 = y_first, sg_c = 0;      // staging cursor (units in order)

    // issues the DMA of the next unit; returns how many VMEM instructions this wave issued (wave-uniform)
    auto stage = [&]() -> int {
        int issued = 0;
        float* xb = xs0 + sg_buf * xfloats;
        const char* src = reinterpret_cast<const char*>(a.in + ((long)(sg_b * H + sg_y0 - 1) * W) * a.in_ld + sg_c * KC);
#pragma unroll
        for (int t = 0; t < TXF; ++t) {
            if (wave + NW * t >= nx) break;
            const int gy = sg_y0 - 1 + xs_row[t];
            float* ldst = xb + xs_ldst[t];                               // wave-uniform
            // rows above / below the image fetch zeros (rv_zero_piece) instead of branching into a store: one select, one DMA
            const bool rowok = (unsigned)gy < (unsigned)H;
            if (xs_lane[t]) glds16(reinterpret_cast<const float*>(rowok ? src + xs_goff[t] : zsrc), ldst);
            ++issued;
        }
        for (int i = wave + NW * TXF; i < nx; i += NW) {                 // narrow workgroups on wide rows
            const int row = i / ipr, k = i - row * ipr;
            const int gy = sg_y0 - 1 + row;
            const int px = k * PPI + lane / Q, q = lane - (lane / Q) * Q;
            float* ldst = xb + (row * W2 + 1 + k * PPI) * KC;
            const bool rowok = (unsigned)gy < (unsigned)H;
            if (px < W)
                glds16(rowok ? reinterpret_cast<const float*>(src) + ((long)row * W + px) * a.in_ld + q * 4 : reinterpret_cast<const float*>(zsrc), ldst);
            ++issued;
        }
        if (nchunk > 1 || sg_u == 0) {
            float* wb = ws0 + ((nchunk > 1) ? sg_buf : 0) * WFLOATS;
            const char* wsrc = reinterpret_cast<const char*>(a.wpack + (long)sg_c * a.ntile_n * 64 * R);
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                if (wave + NW * t >= NWF) break;
                if (w_lane[t]) glds16(reinterpret_cast<const float*>(wsrc + w_off[t]), wb + (wave + NW * t) * 256);
                ++issued;
            }
        }
        ++sg_u;
        if (++sg_buf == nbuf) sg_buf = 0;
        if (++sg_c == nchunk) {
            sg_c = 0;
            sg_y0 += TH;
            if (sg_y0 >= H) { sg_y0 = 0; ++sg_b; }
        }
        return issued;
    };

    // halo columns of both buffers are zero for the whole kernel (the DMA never touches them)
    for (int k = tid; k < nbuf * nrow * 2 * KC; k += NTHR) {
        const int buf = k / (nrow * 2 * KC), rem = k - buf * (nrow * 2 * KC);
        const int row = rem / (2 * KC), rem2 = rem - row * (2 * KC);
        const int side = rem2 / KC, ch = rem2 - side * KC;
        xs0[buf * xfloats + (row * W2 + (side ? W + 1 : 0)) * KC + ch] = 0.f;
    }

    f32x4 acc[MTW][NT];
    int lbase[MTW];
    bool pv[MTW];
    f32x4 bv[NT];
    __shared__ __attribute__((aligned(16))) float cf[4 * 64];   // mean | invstd | scale | shift of this workgroup's channels
    if (a.bn_z) {
        for (int idx = tid; idx < 4 * NT * 16; idx += NTHR) {
            const int k = idx / (NT * 16), cl = idx - k * (NT * 16), ch = nt0 * 16 + cl;
            cf[k * 64 + cl] = ch < a.Cout ? a.bn_coef[k * a.Cout + ch] : 0.f;
        }
    }
    f32x4 st1[NT], st2[NT];                      // BatchNorm statistics of this thread's outputs (a.bn_sums)
#pragma unroll
    for (int n = 0; n < NT; ++n) st1[n] = st2[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int cb = (nt0 + n) * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[n][r] = (a.bias && cb + r < a.Cout) ? a.bias[cb + r] : 0.f;
    }
    // units 0 .. nbuf-2 are in flight before the first multiply; `ahead` = VMEM instructions of this wave that
    // belong to LATER units than the one about to be multiplied (they may stay outstanding: loads return in order)
    if (ABL(aa) & 128) return;
    stage();
    int ahead = 0;
    if (nbuf > 2 && nunits > 1) ahead = stage();
    if (ABL(aa) & 64) return;
    int cu_buf = 0, cu_b = b_first, cu_y0 = y_first, cu_c = 0;                // compute cursor
    for (int u = 0; u < nunits; ++u) {
        const int b = cu_b, y0 = cu_y0, c = cu_c, buf = cu_buf;
        if (++cu_buf == nbuf) cu_buf = 0;
        if (++cu_c == nchunk) {
            cu_c = 0;
            cu_y0 += TH;
            if (cu_y0 >= H) { cu_y0 = 0; ++cu_b; }
        }
        const int th = min(TH, H - y0);
        const int npx = th * W;
        const int ntile = (npx + 15) >> 4;
        if (c == 0) {
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int t = wave + NW * m;
                const int p = t * 16 + j;
                pv[m] = t < ntile && p < npx;
                const unsigned pp = pv[m] ? (unsigned)p : 0u;
                const int ty = (int)fastdiv(pp, a.fd_w), tx = (int)pp - ty * W;
                lbase[m] = (ty * W2 + tx) * KC + g * R;
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        if (nbuf == 3) wait_vmcnt_le(ahead);                // this wave's share of unit u has landed
        // two buffers: nothing but unit u's DMA is in flight, wait for all of it.  EXPLICIT: the tap loop's ds_reads are inline asm, invisible
        // to the compiler's wait-count insertion, so the landing of the LDS-DMA must not hang on what it happens to emit for the barrier
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(ABL(aa) & 1)) __syncthreads();              // ... and everyone's; unit u-1's buffer is free
        // The waves that share a SIMD stage at opposite ends of the unit (prefetch distance 2 only), so one wave's
        // address arithmetic / DMA issue overlaps the other's MFMAs instead of both idling the matrix pipe together.
        const bool late = aa.skew && ((wave >> 2) & 1);
        if (!late && !(ABL(aa) & 2)) ahead = (sg_u < nunits) ? stage() : 0;   // DMA runs under the MFMAs below
        const float* xs = xs0 + buf * xfloats;
        const float* ws = ws0 + ((nchunk > 1) ? buf : 0) * WFLOATS;
        // Software-pipelined tap loop: the fragments of tap t+1 are in flight while tap t is multiplied.  The
        // ds_reads and their s_waitcnt are issued by hand: with LDS-DMA pending the compiler's own wait-count
        // model degrades every LDS wait to lgkmcnt(0), which would serialise the prefetch again.
        vec wf[2][NT], xf[2][MTW];
        const unsigned ws_a = lds_addr(ws) + lane * (R * 4);
        unsigned xs_a[MTW];
#pragma unroll
        for (int m = 0; m < MTW; ++m) xs_a[m] = lds_addr(xs) + lbase[m] * 4;
        const unsigned rowb = (unsigned)W2 * KC * 4;
        // fragment reads of one tap: the weight offsets and the column part of the tap shift are instruction immediates, the row
        // part is one add per (tile, kernel row)
        auto ldtap = [&](auto bufc, auto tapc) {
            constexpr int bsel = decltype(bufc)::value, tap = decltype(tapc)::value;
            sfor<0, NT>([&](auto nc) {
                constexpr int n = decltype(nc)::value;
                lds_read_o<(tap * NT + n) * 64 * R * 4>(wf[bsel][n], ws_a);
            });
#pragma unroll
            for (int m = 0; m < MTW; ++m) lds_read_o<(tap % 3) * KC * 4>(xf[bsel][m], xs_a[m] + (tap / 3) * rowb);
        };
        ldtap(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        if (!(ABL(aa) & 4))
        sfor<0, 9>([&](auto tapc) {
            constexpr int tap = decltype(tapc)::value;
#ifdef RV_ABLATION
            if ((ABL(aa) & 256) && tap >= 4) return;         // Winograd F(2x2,3x3) cost probe: 4 MFMA groups per pixel tile ...
            if (ABL(aa) & 512) {                             // ... plus its input-transform adds (8 per fragment and group)
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int q = 0; q < 8; ++q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(xf[tap & 1][m][q % R]) : "v"(xf[tap & 1][m][(q + 1) % R]));
            }
#endif
            if constexpr (tap + 1 < 9) {
                if (!(ABL(aa) & 16)) ldtap(std::integral_constant<int, (tap + 1) & 1>{}, std::integral_constant<int, tap + 1>{});
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NT + MTW) : "memory");   // tap's own fragments have landed
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (BF) {
                s16x4 wb[NT], xb[MTW];
#pragma unroll
                for (int n = 0; n < NT; ++n) wb[n] = to_bf16x4(wf[tap & 1][n]);
#pragma unroll
                for (int m = 0; m < MTW; ++m) xb[m] = to_bf16x4(xf[tap & 1][m]);
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wb[n], xb[m], acc[m][n], 0, 0, 0);
            } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[tap & 1][n][r], xf[tap & 1][m][r], acc[m][n], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (late && !(ABL(aa) & 2)) ahead = (sg_u < nunits) ? stage() : 0;
        if (c != nchunk - 1) continue;
        // epilogue of this band
        const long pix0 = ((long)b * H + y0) * W;
        f32x4 zreg[MTW][NT];                     // fused BatchNorm backward: all z loads of the band in flight at once
        if (a.bn_z) {
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const long opix = pix0 + (wave + NW * m) * 16 + j;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int co0 = (nt0 + n) * 16 + 4 * g;
                    zreg[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (!pv[m] || co0 >= a.Cout) continue;
                    const float* zp = a.bn_z + opix * a.bn_z_ld + co0;
                    if ((a.bn_z_ld & 3) == 0 && co0 + 3 < a.Cout) zreg[m][n] = *reinterpret_cast<const f32x4*>(zp);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (co0 + r < a.Cout) zreg[m][n][r] = zp[r];
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            if (!pv[m] || (ABL(aa) & 8)) continue;
            const long opix = pix0 + (wave + NW * m) * 16 + j;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int co0 = (nt0 + n) * 16 + 4 * g;
                if (co0 >= a.Cout) continue;
                float* o = a.out + opix * a.out_ld + co0;
                f32x4 v = acc[m][n] + bv[n];
                if (a.vec_store && co0 + 3 < a.Cout) {
                    if (a.accumulate) { f32x4 old = *reinterpret_cast<f32x4*>(o); v += old; }
                    *reinterpret_cast<f32x4*>(o) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co0 + r < a.Cout) {
                            if (a.accumulate) v[r] += o[r];
                            o[r] = v[r];
                        }
                }
                if (a.bn_z) {
                    // backward statistics: coefficients of these 4 channels from LDS, z of this pixel prefetched above
                    const f32x4 mean4 = *reinterpret_cast<const f32x4*>(&cf[0 * 64 + n * 16 + 4 * g]);
                    const f32x4 inv4 = *reinterpret_cast<const f32x4*>(&cf[1 * 64 + n * 16 + 4 * g]);
                    const f32x4 sc4 = *reinterpret_cast<const f32x4*>(&cf[2 * 64 + n * 16 + 4 * g]);
                    const f32x4 sh4 = *reinterpret_cast<const f32x4*>(&cf[3 * 64 + n * 16 + 4 * g]);
                    const f32x4 z4 = zreg[m][n];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float zh = fmaf(z4[r], sc4[r], sh4[r]);
                        const float dd = zh > 0.f ? v[r] : v[r] * a.bn_slope;
                        st1[n][r] += dd;
                        st2[n][r] = fmaf(dd, (z4[r] - mean4[r]) * inv4[r], st2[n][r]);
                    }
                } else {
                    st1[n] += v;
                    st2[n] += v * v;
                }
            }
        }
    }
    // Fused BatchNorm statistics: the consumer's per-channel sum / sum of squares leave with the conv instead of
    // costing another pass over the output.  16 pixel lanes -> one lane (xor shuffles), waves -> LDS, then ONE fp64
    // atomic per channel and workgroup (a persistent grid: a few hundred atomics per address at most).
    if (a.bn_sums) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = st1[n][r], q = st2[n][r];
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) {
                    u += __shfl_xor(u, d, 64);
                    q += __shfl_xor(q, d, 64);
                }
                st1[n][r] = u; st2[n][r] = q;
            }
        __syncthreads();                                   // every wave is done with the unit buffers
        float* red = smem;                                 // [NW][NT*16][2]
        if (j == 0) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    red[((wave * NT + n) * 16 + 4 * g + r) * 2] = st1[n][r];
                    red[((wave * NT + n) * 16 + 4 * g + r) * 2 + 1] = st2[n][r];
                }
        }
        __syncthreads();
        for (int t = tid; t < NT * 16 * 2; t += NTHR) {
            const int cl = t >> 1, which = t & 1, ch = nt0 * 16 + cl;
            double dsum = 0.0;
            for (int w = 0; w < NW; ++w) dsum += (double)red[((w * NT) * 16 + cl) * 2 + which];
            if (ch < a.Cout) atomicAdd(&a.bn_sums[(blockIdx.x % RV_BN_NREP) * 2 * a.Cout + which * a.Cout + ch], dsum);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) form of the persistent 3x3 kernel (algo families 0x6NM, with HALF 0xANM / 0xCNM; 16-channel chunks only).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A    per 4x4 input patch d (stride 2) and 2x2 output tile Y
//
// 16 element-wise products per tile and (cin, cout) pair instead of 36: the matrix pipe does 2.25x fewer MFMAs.  The "pixel"
// dimension of the MFMA is a run of 16 TILES of the band (flattened (tile row, tile column) index, as the direct kernel flattens
// pixels); lane (j, g) reads the 4x4 patch of tile j for channels 4g..4g+3 of the chunk straight from the staged band (16
// ds_read_b128), transforms it in registers (32 adds per channel) and feeds V[xi] to the 16 x NT x 4 MFMAs of the chunk; the
// transformed weights U[xi] = G g G^T come pre-packed (rv_pack_weights appends them behind the nine tap fragments).  The
// accumulators hold M[xi] for xi = 0..15; the band epilogue applies A^T . A in registers and shares bias / statistics / fused
// BatchNorm-backward / store logic with the direct form.  Unit staging (LDS-DMA, two buffers) is the direct kernel's, with the
// band padded to an even width.
// ------------------------------------------------------------------------------------------
// HALF (families 0xANM: 8 waves, 0xCNM: 12 waves): the patch is read and transformed one HALF chunk (8 channels, two k-steps) at a time -- 32 instead of 64 patch
// registers, the transform in packed math -- which is what lets the NT = 2 tile (two n-tiles share one patch: half the patch
// reads / transforms / x staging / barriers per MFMA) fit the 256 registers of two waves per SIMD.  With NT = 1 the kernel needs 165
// registers: three waves per SIMD (12-wave workgroups).
// LDS image of a staged band (round 4).  A row is a sequence of 1 KiB PIECES of 16 pixels x 16 channels; piece k holds the padded
// columns X = 16k .. 16k+15 (X = x + 1: X = 0 is the left halo column).  One LDS-DMA instruction fills one piece (lane i -> bytes
// [16 i, 16 i + 16) of the piece -- fixed by the hardware), but WHICH (pixel, channel quad) a lane fetches is free, so the position of a
// quad inside its piece is a permutation chosen for the READ side: lane (j, g) of the multiplying wave reads, for each of the 16 patch
// elements, pixel X = 2 (tx_j + const) + const' -- neighbouring lanes are TWO pixels (128 bytes in a pixel-major image) apart, which
// put the 16 lanes of a ds_read_b128 lane group on 4 of the 16 slots of a bank row (4-way conflict, 61 % of all LDS cycles in the
// round-3 kernel).  Layout 1: slot = parity * 32 + (quad >> 1) * 16 + (pixel >> 1) * 2 + (quad & 1): the eight tiles x two quads
// (g, g ^ 1) of a lane group land on sixteen different slots -- conflict-free.  Layout 0: even / odd pixel planes, pixel-major inside
// a plane (slot = parity * 32 + (pixel >> 1) * 4 + quad): 2-way, every DMA lane quad still fetches 64 contiguous bytes.
// BNZ: the launch carries the fused BatchNorm-backward reduction (a.bn_z != NULL); its epilogue needs ~30 more registers, so the plain
// launches get their own instance
template <int NT, int MTW, int NW, bool HALF = false, int LAY = 1, bool BNZ = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void conv3x3_wino_k(ConvLdsArgs aa) {     // (four waves: the half-CU experiment family 0xE, two workgroups per CU -> <= 256 registers)
    constexpr int NTHR = NW * 64;
    constexpr int KC = 16;                       // channels per chunk
    constexpr int WFLOATS = 16 * NT * 256;       // 16 xi x NT fragments x 64 lanes x 4 floats
    const ConvArgs& a = aa.c;
    if (ABL(aa) & 64) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int W = a.W, H = a.H, WT = (W + 1) >> 1, TH = aa.TH;
    const int NP = (2 * WT + 2 + 15) >> 4;               // pieces per row
    const int RP = NP * 256;                             // floats per row
    const int nrow = TH + 2;
    const int xfloats = nrow * RP;
    const int vid = aa.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int grp = vid / aa.nsplit, split = vid - grp * aa.nsplit;
    const int nt0 = split * NT;
    float* xs0 = smem;                                   // [2][nrow][NP][256]
    float* ws0 = smem + 2 * xfloats;                     // [nchunk if resident, else 2 (1 if one chunk)][16][NT][64][4]
    const int band_lo = grp * aa.bands_per_wg;
    const int band_hi = min(band_lo + aa.bands_per_wg, aa.total_bands);
    if (band_lo >= band_hi) return;
    const int nchunk = a.nchunk;
    const bool wres = aa.wres != 0;                      // all chunks' weights stay in LDS for the whole kernel (loaded with unit 0)
    const int nunits = (band_hi - band_lo) * nchunk;

    // ---- staging plan: slot i = (row, piece) of the unit's input rows, dealt round-robin over the waves; per slot the lane's byte
    // offset from the unit's first input row and whether it fetches a real pixel (else: zeros).  Per unit that leaves one scalar row
    // test, one select and one DMA instruction per slot. ----
    constexpr int TXF = NW == 12 ? 3 : 4;                // slots planned in registers (any further: generic loop)
    constexpr int NWF = 16 * NT;                         // weight fragments per chunk
    constexpr int TW = (NWF + NW - 1) / NW;
    const int nx = nrow * NP;
    int lp, lq;
    wino_lane<LAY>(lane, lp, lq);
    // The input rows are fetched through a BUFFER RESOURCE over the input view: a lane whose byte offset lies outside it gets ZEROS
    // written to its LDS slot (measured: tools/probes/buffer_lds_oob.hip).  So "this lane is a halo column" and "this row is above /
    // below the image" are both just an offset bump of OOB = 0x40000000 (the view is < 0x3f000000 bytes: host check) -- no per-lane
    // mask, no pointer select, no 64-bit address arithmetic; per slot: one scalar row test, one scalar select, one vector add, the DMA.
    constexpr unsigned OOB = 0x40000000u;
    const rv_rsrc_t rs_in = rv_make_rsrc(a.in, aa.in_bytes);
    int xs_row[TXF], xs_ldst[TXF];
    unsigned xs_goff[TXF];
#pragma unroll
    for (int t = 0; t < TXF; ++t) {
        const int i = wave + NW * t;
        const int row = i / NP, k = i - row * NP;
        const int px = k * 16 + lp - 1;
        xs_row[t] = row;
        xs_ldst[t] = (row * RP + k * 256) * 4;               // bytes
        xs_goff[t] = (unsigned)px < (unsigned)W ? (unsigned)((row * W + px) * a.in_ld + lq * 4) * 4u : OOB;
    }
    unsigned w_off[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        const int f = wave + NW * t;                     // fragment = xi * NT + n
        const int xi = f / NT, n = f - xi * NT;
        w_off[t] = (unsigned)(((xi * nchunk * a.ntile_n) + nt0 + n) * 256 + lane * 4) * 4u;
    }
    const float* wino = a.wpack + (long)9 * nchunk * a.ntile_n * 256;      // the Winograd section of the packed weights
    const int b_first = band_lo / aa.nbands, y_first = (band_lo - b_first * aa.nbands) * TH;
    int sg_u = 0, sg_buf = 0, sg_b = b_first, sg_y0 = y_first, sg_c = 0;

    auto stage = [&]() -> int {
        int issued = 0;
        float* xb = xs0 + sg_buf * xfloats;
        // byte offset of the unit's first staged row (row y0 - 1: may be "negative" -- all arithmetic is mod 2^32 and the sum with a real
        // lane's offset is the true offset whenever the row is inside the image)
        const unsigned ubase = (unsigned)((((sg_b * H + sg_y0 - 1) * W) * a.in_ld + sg_c * KC) * 4);
#pragma unroll
        for (int t = 0; t < TXF; ++t) {
            if (wave + NW * t >= nx) break;
            const unsigned add = (unsigned)(sg_y0 - 1 + xs_row[t]) < (unsigned)H ? ubase : ubase + OOB;
            rv_buf_lds16(rs_in, reinterpret_cast<char*>(xb) + xs_ldst[t], xs_goff[t] + add);
            ++issued;
        }
        for (int i = wave + NW * TXF; i < nx; i += NW) {                    // narrow workgroups on wide rows
            const int row = i / NP, k = i - row * NP;
            const int px = k * 16 + lp - 1;
            const unsigned add = (unsigned)(sg_y0 - 1 + row) < (unsigned)H ? ubase : ubase + OOB;
            const unsigned goff = (unsigned)px < (unsigned)W ? (unsigned)((row * W + px) * a.in_ld + lq * 4) * 4u : OOB;
            rv_buf_lds16(rs_in, xb + row * RP + k * 256, goff + add);
            ++issued;
        }
        if (wres) {
            // resident weights: chunk c arrives with unit c of the first band (not all of them ahead of unit 0: the first wait of
            // the kernel is then for one chunk's weights, not for nchunk)
            if (sg_u < nchunk) {
                const char* wsrc = reinterpret_cast<const char*>(wino + (long)sg_u * a.ntile_n * 256);
#pragma unroll
                for (int t = 0; t < TW; ++t) {
                    if (wave + NW * t >= NWF) break;
                    glds16(reinterpret_cast<const float*>(wsrc + w_off[t]), ws0 + sg_u * WFLOATS + (wave + NW * t) * 256);
                    ++issued;
                }
            }
        } else if (nchunk > 1 || sg_u == 0) {
            float* wb = ws0 + ((nchunk > 1) ? sg_buf : 0) * WFLOATS;
            const char* wsrc = reinterpret_cast<const char*>(wino + (long)sg_c * a.ntile_n * 256);
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                if (wave + NW * t >= NWF) break;
                glds16(reinterpret_cast<const float*>(wsrc + w_off[t]), wb + (wave + NW * t) * 256);
                ++issued;
            }
        }
        ++sg_u;
        sg_buf ^= 1;
        if (++sg_c == nchunk) {
            sg_c = 0;
            sg_y0 += TH;
            if (sg_y0 >= H) { sg_y0 = 0; ++sg_b; }
        }
        return issued;
    };

    typedef typename VecR<HALF ? 2 : 4>::T pvec;         // what one patch / weight read delivers
    constexpr int NH = HALF ? 2 : 1;                     // passes over the chunk
    constexpr int WD = (NT == 1 && NW <= 8) ? 3 : 2;     // weight-fragment ring: prefetch distance WD - 1 (register budget)
    f32x4 acc[MTW][NT][16];
    int lb0[MTW], lb1[MTW], tyx[MTW];
    bool tv[MTW];
    __shared__ __attribute__((aligned(16))) float cf[4 * 64];
    if (BNZ) {
        for (int idx = tid; idx < 4 * NT * 16; idx += NTHR) {
            const int k = idx / (NT * 16), cl = idx - k * (NT * 16), ch = nt0 * 16 + cl;
            cf[k * 64 + cl] = ch < a.Cout ? a.bn_coef[k * a.Cout + ch] : 0.f;
        }
    }
    f32x4 st1[NT], st2[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) st1[n] = st2[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (ABL(aa) & 128) return;
    stage();
    if (ABL(aa) & 1024) { wait_vmcnt_le(0); return; }
    int cu_buf = 0, cu_b = b_first, cu_y0 = y_first, cu_c = 0;
    for (int u = 0; u < nunits; ++u) {
        const int b = cu_b, y0 = cu_y0, c = cu_c, buf = cu_buf;
        cu_buf ^= 1;
        if (++cu_c == nchunk) {
            cu_c = 0;
            cu_y0 += TH;
            if (cu_y0 >= H) { cu_y0 = 0; ++cu_b; }
        }
        const int th = min(TH, H - y0);
        const int ntiles = ((th + 1) >> 1) * WT;
        if (c == 0) {
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int t = (wave + NW * m) * 16 + j;
                tv[m] = t < ntiles;
                const unsigned tt = tv[m] ? (unsigned)t : 0u;
                const int ty = (int)fastdiv(tt, a.fd_pw), tx = (int)tt - ty * WT;       // fd_pw divides by WT here
                tyx[m] = (ty << 16) | tx;
                lb0[m] = (2 * ty) * RP * 4 + wino_pair_off<LAY>(tx, g);
                lb1[m] = (2 * ty) * RP * 4 + wino_pair_off<LAY>(tx + 1, g);
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int xi = 0; xi < 16; ++xi) acc[m][n][xi] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        // unit u's DMA has landed (two buffers: nothing else is in flight).  EXPLICIT wait: the patch / weight ds_reads below are inline asm,
        // invisible to the compiler's wait-count insertion -- the barrier must not depend on the vmcnt(0) it happens to emit today
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(ABL(aa) & 1)) __syncthreads();
        if (!(ABL(aa) & 2) && sg_u < nunits) stage();
        const unsigned xs_a = lds_addr(xs0 + buf * xfloats);
        const unsigned ws_a = lds_addr(ws0 + (wres ? c : ((nchunk > 1) ? buf : 0)) * WFLOATS) + lane * 16;
        const unsigned rp4 = (unsigned)RP * 4u;
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            if ((wave + NW * m) * 16 >= ntiles) continue;                   // wave-uniform: this tile group is beyond the band
            sfor<0, NH>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                pvec d[16];
                pvec wf[WD][NT];
                unsigned ra[4][2];                                            // per patch row: address of the (even) pixel pairs tx, tx + 1
#pragma unroll
                for (int er = 0; er < 4; ++er) {
                    ra[er][0] = xs_a + lb0[m] + er * rp4;
                    ra[er][1] = xs_a + lb1[m] + er * rp4;
                }
                // patch element (er, ec): pixel X = 2 tx + ec -> pair tx + (ec >> 1); the odd pixel of a pair sits 512 bytes behind the
                // even one; HALF: second half-chunk 8 bytes into the quad.  All offsets are instruction immediates.
#pragma unroll
                for (int er = 0; er < 4; ++er) {
                    lds_read_o<h * 8>(d[er * 4 + 0], ra[er][0]);
                    lds_read_o<512 + h * 8>(d[er * 4 + 1], ra[er][0]);
                    lds_read_o<h * 8>(d[er * 4 + 2], ra[er][1]);
                    lds_read_o<512 + h * 8>(d[er * 4 + 3], ra[er][1]);
                }
                // weight fragments: WD - 1 xi ahead of the multiplies
                auto ldw = [&](auto xic) {
                    constexpr int xi = decltype(xic)::value;
                    sfor<0, NT>([&](auto nc) {
                        constexpr int n = decltype(nc)::value;
                        lds_read_o<(xi * NT + n) * 1024 + h * 8>(wf[xi % WD][n], ws_a);
                    });
                };
                sfor<0, WD - 1>([&](auto xic) { ldw(xic); });
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((WD - 1) * NT) : "memory");      // the patch has landed
                __builtin_amdgcn_sched_barrier(0);
                // V = B^T d B, in place (rows, then columns); pk_add / pk_sub work on channel PAIRS: one v_pk_add_f32 each
                if (!(ABL(aa) & 16)) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const pvec e0 = d[cc], e1 = d[4 + cc], e2 = d[8 + cc], e3 = d[12 + cc];
                        d[cc] = pk_sub(e0, e2); d[4 + cc] = pk_add(e1, e2); d[8 + cc] = pk_sub(e2, e1); d[12 + cc] = pk_sub(e1, e3);
                    }
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const pvec e0 = d[4 * rr], e1 = d[4 * rr + 1], e2 = d[4 * rr + 2], e3 = d[4 * rr + 3];
                        d[4 * rr] = pk_sub(e0, e2); d[4 * rr + 1] = pk_add(e1, e2); d[4 * rr + 2] = pk_sub(e2, e1); d[4 * rr + 3] = pk_sub(e1, e3);
                    }
                }
                sfor<0, 16>([&](auto xic) {
                    constexpr int xi = decltype(xic)::value;
                    if constexpr (xi + WD - 1 < 16) {
                        if (!(ABL(aa) & 32)) ldw(std::integral_constant<int, xi + WD - 1>{});
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((WD - 1) * NT) : "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((15 - xi) * NT) : "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(ABL(aa) & 4))
#pragma unroll
                    for (int r = 0; r < (HALF ? 2 : 4); ++r)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[m][n][xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[xi % WD][n][r], d[xi][r], acc[m][n][xi], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        }
        if (c != nchunk - 1 || (ABL(aa) & 8)) continue;
        // ---- epilogue of this band: Y = A^T M A (packed math on channel pairs), then bias / statistics / store per output pixel ----
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            if (!tv[m]) continue;
            const int ty = tyx[m] >> 16, tx = tyx[m] & 0xffff;
            const int oy = 2 * ty, ox = 2 * tx;
            const long pix00 = ((long)b * H + y0 + oy) * W + ox;
            const bool vy1 = oy + 1 < th, vx1 = ox + 1 < W;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int co0 = (nt0 + n) * 16 + 4 * g;
                if (co0 >= a.Cout) continue;
                f32x4 y[4];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f32x2 s0[4], s1[4];
                    f32x2 bq;                       // bias of this lane's channel pair (once per band: not worth registers across the units)
#pragma unroll
                    for (int r = 0; r < 2; ++r) bq[r] = (a.bias && co0 + 2 * q + r < a.Cout) ? a.bias[co0 + 2 * q + r] : 0.f;
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) {
                        const f32x2 m0 = (f32x2){acc[m][n][bb][2 * q], acc[m][n][bb][2 * q + 1]};
                        const f32x2 m1 = (f32x2){acc[m][n][4 + bb][2 * q], acc[m][n][4 + bb][2 * q + 1]};
                        const f32x2 m2 = (f32x2){acc[m][n][8 + bb][2 * q], acc[m][n][8 + bb][2 * q + 1]};
                        const f32x2 m3 = (f32x2){acc[m][n][12 + bb][2 * q], acc[m][n][12 + bb][2 * q + 1]};
                        s0[bb] = m0 + m1 + m2;
                        s1[bb] = m1 - m2 - m3;
                    }
                    const f32x2 y0_ = s0[0] + s0[1] + s0[2] + bq, y1_ = s0[1] - s0[2] - s0[3] + bq;
                    const f32x2 y2_ = s1[0] + s1[1] + s1[2] + bq, y3_ = s1[1] - s1[2] - s1[3] + bq;
                    y[0][2 * q] = y0_[0]; y[0][2 * q + 1] = y0_[1]; y[1][2 * q] = y1_[0]; y[1][2 * q + 1] = y1_[1];
                    y[2][2 * q] = y2_[0]; y[2][2 * q + 1] = y2_[1]; y[3][2 * q] = y3_[0]; y[3][2 * q + 1] = y3_[1];
                }
                f32x4 z4[4];
                if (BNZ) {
                    __builtin_amdgcn_sched_barrier(0);       // the accumulators are dead from here: keep the z loads behind the transform
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        z4[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (((p >> 1) && !vy1) || ((p & 1) && !vx1)) continue;
                        const float* zp = a.bn_z + (pix00 + (p >> 1) * W + (p & 1)) * a.bn_z_ld + co0;
                        if ((a.bn_z_ld & 3) == 0 && co0 + 3 < a.Cout) z4[p] = *reinterpret_cast<const f32x4*>(zp);
                        else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (co0 + r < a.Cout) z4[p][r] = zp[r];
                        }
                    }
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if (((p >> 1) && !vy1) || ((p & 1) && !vx1)) continue;
                    float* o = a.out + (pix00 + (p >> 1) * W + (p & 1)) * a.out_ld + co0;
                    f32x4 v = y[p];
                    if (a.vec_store && co0 + 3 < a.Cout) {
                        if (a.accumulate) { f32x4 old = *reinterpret_cast<f32x4*>(o); v += old; }
                        *reinterpret_cast<f32x4*>(o) = v;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (co0 + r < a.Cout) {
                                if (a.accumulate) v[r] += o[r];
                                o[r] = v[r];
                            }
                    }
                    if (BNZ) {
                        const f32x4 mean4 = *reinterpret_cast<const f32x4*>(&cf[0 * 64 + n * 16 + 4 * g]);
                        const f32x4 inv4 = *reinterpret_cast<const f32x4*>(&cf[1 * 64 + n * 16 + 4 * g]);
                        const f32x4 sc4 = *reinterpret_cast<const f32x4*>(&cf[2 * 64 + n * 16 + 4 * g]);
                        const f32x4 sh4 = *reinterpret_cast<const f32x4*>(&cf[3 * 64 + n * 16 + 4 * g]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float zh = fmaf(z4[p][r], sc4[r], sh4[r]);
                            const float dd = zh > 0.f ? v[r] : v[r] * a.bn_slope;
                            st1[n][r] += dd;
                            st2[n][r] = fmaf(dd, (z4[p][r] - mean4[r]) * inv4[r], st2[n][r]);
                        }
                    } else {
                        st1[n] += v;
                        st2[n] += v * v;
                    }
                }
            }
        }
    }
    if (a.bn_sums && !(ABL(aa) & 4096)) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = st1[n][r], q = st2[n][r];
#pragma unroll
                for (int dd = 1; dd < 16; dd <<= 1) {
                    u += __shfl_xor(u, dd, 64);
                    q += __shfl_xor(q, dd, 64);
                }
                st1[n][r] = u; st2[n][r] = q;
            }
        __syncthreads();
        float* red = smem;                                 // [NW][NT*16][2]
        if (j == 0) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    red[((wave * NT + n) * 16 + 4 * g + r) * 2] = st1[n][r];
                    red[((wave * NT + n) * 16 + 4 * g + r) * 2 + 1] = st2[n][r];
                }
        }
        __syncthreads();
        for (int t = tid; t < NT * 16 * 2; t += NTHR) {
            const int cl = t >> 1, which = t & 1, ch = nt0 * 16 + cl;
            double dsum = 0.0;
            for (int w = 0; w < NW; ++w) dsum += (double)red[((w * NT) * 16 + cl) * 2 + which];
            if (ch < a.Cout) atomicAdd(&a.bn_sums[(blockIdx.x % RV_BN_NREP) * 2 * a.Cout + which * a.Cout + ch], dsum);
        }
    }
}

// ------------------------------------------------------------------------------------------
// small-channel convolution on the VALU (Cin==1, or Cout<=2): pure bandwidth kernels
// ------------------------------------------------------------------------------------------
struct SmallArgs {
    const float* in;  int in_ld;  int H, W;
    float* out;       int out_ld; int Ho, Wo;
    int B;
    const float* wplain;   // [tap][CIN][COUT]
    const float* bias;
    long npix;
    int accumulate;
    FastDiv fd_plane, fd_wo;   // divide by Ho*Wo, by Wo (npix * COUT/4 < 2^31)
    double* bn_sums;           // optional fused BatchNorm statistics of the output (COUT >= 4 instances)
    int rows;                  // conv_narrow_out_k / conv_cin12_k: output rows a workgroup walks down its column strip
    const float* bn_z; int bn_z_ld; const float* bn_coef; float bn_slope;    // conv_cin12_k: fused BatchNorm backward reduction
};

// COUT >= 4: four output channels per thread (COUT/4 threads per pixel) so that a wave's stores are
// contiguous 16-byte pieces; the thread's KH*KW*CIN*4 weights and its bias stay in registers while it
// walks pixels with a grid stride (the stride is a multiple of COUT/4, so its channel quad never changes).
// COUT < 4: one thread per pixel, weights through the scalar cache.
// output channels per thread: all of them when the input is 1-2 channels wide (the thread's few input taps are then
// loaded once instead of once per channel quad), else a quad
__host__ __device__ constexpr int small_cpt(int cin, int cout, int taps) {
    return cout < 4 ? cout : ((cin <= 2 && taps > 1) ? cout : 4);
}

// 3x3 (stride 1, pad 1) convolution from CIN in {8, 16} channels down to COUT <= 2 (the last decoder layer 8 -> 2, the
// reconstructor's 16 -> 1 head layer and the input-gradients of the first layers): pure bandwidth kernels whose cost is the
// number of CACHE LINES a wave-instruction touches, not its bytes.  With one thread per pixel every 16-byte load of a wave
// lands in a different line (pixel stride 32-64 B: 64 lines per instruction, each line touched by CIN/4 instructions per
// tap) and the vector L1's tag rate bounds the kernel at 1.2-1.7 TB/s.  Here the CIN/4 channel quads of a pixel sit on
// NEIGHBOURING lanes: a wave-instruction reads 1 KiB of contiguous memory (8 lines), every lane keeps the 9*4*COUT weights of
// its quad in registers, and the per-pixel sum is two xor-shuffles over the quad lanes.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv_narrow_out_k(SmallArgs a) {
    // A workgroup owns a strip of 256 / LPP pixel columns and walks a.rows output rows down it with a SLIDING 3-row window
    // in registers: per output row a lane loads only the three quads of the new bottom row (the other six move up), so every
    // input quad is requested 3 times instead of 9 -- and by the same CU.  (A flat grid-stride loop over pixels put the rows
    // above / below a pixel on other XCDs: the L2s fetched the input 2.6x from HBM, profiles/r02_pmc_traffic.json.)
    constexpr int LPP = CIN / 4;                     // lanes per pixel
    static_assert(LPP == 2 || LPP == 4, "quad lanes must divide the wave");
    constexpr int PXW = 256 / LPP;                   // pixel columns per workgroup
    const int q = (int)(threadIdx.x % LPP);
    const int col = (int)(threadIdx.x / LPP);
    float w[9][4][COUT];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int co = 0; co < COUT; ++co) w[tap][c][co] = a.wplain[(tap * CIN + 4 * q + c) * COUT + co];
    float bias[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) bias[co] = a.bias ? a.bias[co] : 0.f;
    const int nstrip = (a.W + PXW - 1) / PXW;
    const int nband = (a.H + a.rows - 1) / a.rows;
    // blockIdx -> (image, row band, column strip), strips fastest: neighbouring strips (shared halo columns) and bands follow
    int bid = blockIdx.x;
    const int strip = bid % nstrip; bid /= nstrip;
    const int band = bid % nband;
    const int b = bid / nband;
    const int ox = strip * PXW + col;
    const bool colok = ox < a.W;                      // lanes past the row end stay in the loop (shuffles) and do not store
    const int y0 = band * a.rows, y1 = min(y0 + a.rows, a.H);
    const int cx[3] = {min(max(ox - 1, 0), a.W - 1), min(ox, a.W - 1), min(ox + 1, a.W - 1)};
    const bool okx[3] = {ox - 1 >= 0 && ox - 1 < a.W, colok, ox + 1 < a.W};
    const float* img = a.in + (long)b * a.H * a.W * a.in_ld + 4 * q;
    auto load_raw = [&](f32x4 (&r)[3], int iy) {     // the three quads of input row iy (clamped address: always a valid load)
        const float* rowp = img + (long)min(max(iy, 0), a.H - 1) * a.W * a.in_ld;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) r[kx] = *reinterpret_cast<const f32x4*>(rowp + (long)cx[kx] * a.in_ld);
    };
    auto mask_row = [&](f32x4 (&r)[3], int iy) {     // taps outside the image become zero (applied when the row is first used)
        const bool oky = iy >= 0 && iy < a.H;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int c = 0; c < 4; ++c) r[kx][c] = (oky && okx[kx]) ? r[kx][c] : 0.f;
    };
    f32x4 win[3][3], nxt[3];                          // rows oy-1, oy, oy+1; nxt = row oy+2 in flight under the FMAs of row oy
    // (round 6: a prefetch distance of three rows, which pays in conv_cin12_k and wgrad_cin1_k, LOSES here -- 8 -> 2: 18.8 -> 23.2 us, 16 -> 1: 31.1 -> 33.5 --:
    // 36 more registers per lane for the queue of 16-byte quads)
    load_raw(win[0], y0 - 1);
    load_raw(win[1], y0);
    load_raw(win[2], y0 + 1);
    mask_row(win[0], y0 - 1);
    mask_row(win[1], y0);
    mask_row(win[2], y0 + 1);
    for (int oy = y0; oy < y1; ++oy) {
        load_raw(nxt, oy + 2);
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int co = 0; co < COUT; ++co) acc[co] = fmaf(win[ky][kx][c], w[ky * 3 + kx][c][co], acc[co]);
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
#pragma unroll
            for (int d = 1; d < LPP; d <<= 1) acc[co] += __shfl_xor(acc[co], d, 64);
            acc[co] += bias[co];
        }
        if (colok && q == 0) {
            float* o = a.out + (((long)b * a.H + oy) * a.W + ox) * a.out_ld;
#pragma unroll
            for (int co = 0; co < COUT; ++co) o[co] = a.accumulate ? o[co] + acc[co] : acc[co];
        }
        mask_row(nxt, oy + 2);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) { win[0][kx] = win[1][kx]; win[1][kx] = win[2][kx]; win[2][kx] = nxt[kx]; }
    }
}

// 3x3 convolution of a ONE- or TWO-channel input (the first encoder layer 1 -> 16 / 1 -> 8, the input gradient 2 -> 8 of the last
// decoder layer): COUT / 4 lanes per pixel, each owning four output channels with its 36 (72) weights in registers; a workgroup walks
// a.rows output rows down a strip of 1024 / COUT pixel columns with a sliding 3-row window of the input (three new pixels per row,
// shared by the lanes of a pixel), so a wave's stores are 1 KiB contiguous without an LDS transpose and the taps cost 3 loads per
// pixel instead of 9.  Fused BatchNorm statistics of the output (per-lane sums -> pixel lanes by xor shuffles -> waves through LDS
// -> one fp64 atomic per channel and workgroup); with a.bn_z the sums are the BACKWARD reduction of the BatchNorm whose output
// gradient this conv produces (z of the next row prefetched like the input), which saves the standalone pass over dx and z.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv_cin12_k(SmallArgs a) {
    constexpr int LPP = COUT / 4;                    // lanes per pixel
    constexpr int PXW = 256 / LPP;                   // pixel columns per workgroup
    const int q = (int)(threadIdx.x % LPP);
    const int col = (int)(threadIdx.x / LPP);
    float w[9][CIN][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int c = 0; c < 4; ++c) w[tap][ci][c] = a.wplain[(tap * CIN + ci) * COUT + 4 * q + c];
    f32x4 bias = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias = *reinterpret_cast<const f32x4*>(a.bias + 4 * q);
    const int nstrip = (a.W + PXW - 1) / PXW;
    const int nband = (a.H + a.rows - 1) / a.rows;
    int bid = blockIdx.x;
    const int strip = bid % nstrip; bid /= nstrip;
    const int band = bid % nband;
    const int b = bid / nband;
    const int ox = strip * PXW + col;
    const bool colok = ox < a.W;
    const int y0 = band * a.rows, y1 = min(y0 + a.rows, a.H);
    const int cx[3] = {min(max(ox - 1, 0), a.W - 1), min(ox, a.W - 1), min(ox + 1, a.W - 1)};
    const bool okx[3] = {ox - 1 >= 0 && ox - 1 < a.W, colok, ox + 1 < a.W};
    const float* img = a.in + (long)b * a.H * a.W * a.in_ld;
    auto load_raw = [&](float (&r)[3][CIN], int iy) {
        const float* rowp = img + (long)min(max(iy, 0), a.H - 1) * a.W * a.in_ld;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            if constexpr (CIN == 2) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(rowp + (long)cx[kx] * a.in_ld);
                r[kx][0] = v[0]; r[kx][1] = v[1];
            } else {
                r[kx][0] = rowp[(long)cx[kx] * a.in_ld];
            }
        }
    };
    auto mask_row = [&](float (&r)[3][CIN], int iy) {
        const bool oky = iy >= 0 && iy < a.H;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) r[kx][ci] = (oky && okx[kx]) ? r[kx][ci] : 0.f;
    };
    const bool bwd = a.bn_z != nullptr;
    f32x4 c_mean = (f32x4){0.f, 0.f, 0.f, 0.f}, c_inv = c_mean, c_sc = c_mean, c_sh = c_mean;
    if (bwd) {
        c_mean = *reinterpret_cast<const f32x4*>(a.bn_coef + 4 * q);
        c_inv = *reinterpret_cast<const f32x4*>(a.bn_coef + COUT + 4 * q);
        c_sc = *reinterpret_cast<const f32x4*>(a.bn_coef + 2 * COUT + 4 * q);
        c_sh = *reinterpret_cast<const f32x4*>(a.bn_coef + 3 * COUT + 4 * q);
    }
    auto load_z = [&](int y) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(a.bn_z + (((long)b * a.H + min(y, a.H - 1)) * a.W + min(ox, a.W - 1)) * a.bn_z_ld + 4 * q);
    };
    // input rows travel PD rows ahead of their use (round 6: with one row of distance an iteration's ~150 cycles of fmas did not cover an L2 round trip at five
    // waves per SIMD: 1 -> 16 @640x229 26.4 -> 22.7 us; the arithmetic and its order are untouched, so results stay bit-identical)
    constexpr int PD = 3;
    float win[3][3][CIN], nxt[3][CIN], pre[PD][3][CIN];
    load_raw(win[0], y0 - 1); load_raw(win[1], y0); load_raw(win[2], y0 + 1);
#pragma unroll
    for (int d = 0; d < PD; ++d) load_raw(pre[d], y0 + 2 + d);
    f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f}, zn = z4;
    if (bwd) z4 = load_z(y0);
    mask_row(win[0], y0 - 1); mask_row(win[1], y0); mask_row(win[2], y0 + 1);
    f32x4 st1 = (f32x4){0.f, 0.f, 0.f, 0.f}, st2 = st1;
    const bool vec = (a.out_ld & 3) == 0 && ((((uintptr_t)a.out) & 15) == 0);
    for (int oy = y0; oy < y1; ++oy) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                nxt[kx][ci] = pre[0][kx][ci];
#pragma unroll
                for (int d = 0; d + 1 < PD; ++d) pre[d][kx][ci] = pre[d + 1][kx][ci];
            }
        load_raw(pre[PD - 1], oy + 2 + PD);
        if (bwd) zn = load_z(oy + 1);
        f32x4 acc = bias;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = fmaf(win[ky][kx][ci], w[ky * 3 + kx][ci][c], acc[c]);
        if (colok) {
            float* o = a.out + (((long)b * a.H + oy) * a.W + ox) * a.out_ld + 4 * q;
            if (vec) {
                if (a.accumulate) acc += *reinterpret_cast<const f32x4*>(o);
                *reinterpret_cast<f32x4*>(o) = acc;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) { if (a.accumulate) acc[c] += o[c]; o[c] = acc[c]; }
            }
            if (bwd) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float zh = fmaf(z4[c], c_sc[c], c_sh[c]);
                    const float dd = zh > 0.f ? acc[c] : acc[c] * a.bn_slope;
                    st1[c] += dd;
                    st2[c] = fmaf(dd, (z4[c] - c_mean[c]) * c_inv[c], st2[c]);
                }
            } else {
                st1 += acc;
                st2 += acc * acc;
            }
        }
        mask_row(nxt, oy + 2);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) { win[0][kx][ci] = win[1][kx][ci]; win[1][kx][ci] = win[2][kx][ci]; win[2][kx][ci] = nxt[kx][ci]; }
        z4 = zn;
    }
    if (a.bn_sums) {
        __shared__ float red[4][COUT * 2];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float u = st1[c], v = st2[c];
#pragma unroll
            for (int o = 32; o >= LPP; o >>= 1) { u += __shfl_xor(u, o, 64); v += __shfl_xor(v, o, 64); }
            if (lane < LPP) { red[wave][(4 * q + c) * 2] = u; red[wave][(4 * q + c) * 2 + 1] = v; }
        }
        __syncthreads();
        if (threadIdx.x < COUT * 2) {
            const int ch = threadIdx.x >> 1, which = threadIdx.x & 1;
            const double d = (double)red[0][threadIdx.x] + (double)red[1][threadIdx.x] + (double)red[2][threadIdx.x] + (double)red[3][threadIdx.x];
            atomicAdd(&a.bn_sums[(blockIdx.x % RV_BN_NREP) * 2 * COUT + which * COUT + ch], d);
        }
    }
}

template <int CIN, int COUT, int KH, int KW, int S, int P>
__global__ __launch_bounds__(256) void conv_small_k(SmallArgs a) {
    constexpr int CPT = small_cpt(CIN, COUT, KH * KW);   // output channels per thread
    constexpr int TPP = COUT / CPT;                  // threads per pixel
    constexpr bool WREG = (TPP > 1) && (KH * KW * CIN * CPT <= 80);
    const unsigned tstride = gridDim.x * blockDim.x;
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned tend = (unsigned)a.npix * TPP;
    const int c0 = (int)(t % TPP) * CPT;
    float wreg[WREG ? KH * KW * CIN * CPT : 1];
    if constexpr (WREG) {
#pragma unroll
        for (int tap = 0; tap < KH * KW; ++tap)
#pragma unroll
            for (int c = 0; c < CIN; ++c)
#pragma unroll
                for (int co = 0; co < CPT; ++co) wreg[(tap * CIN + c) * CPT + co] = a.wplain[(tap * CIN + c) * COUT + c0 + co];
    }
    float bias[CPT];
#pragma unroll
    for (int co = 0; co < CPT; ++co) bias[co] = a.bias ? a.bias[c0 + co] : 0.f;
    float st1[CPT], st2[CPT];                        // BatchNorm statistics of this thread's channel quad
#pragma unroll
    for (int co = 0; co < CPT; ++co) st1[co] = st2[co] = 0.f;
    for (; t < tend; t += tstride) {
        const unsigned p = t / TPP;
        const int b = (int)fastdiv(p, a.fd_plane);
        const unsigned rem = p - (unsigned)b * (unsigned)(a.Ho * a.Wo);
        const int oy = (int)fastdiv(rem, a.fd_wo), ox = (int)rem - oy * a.Wo;
        float acc[CPT];
#pragma unroll
        for (int co = 0; co < CPT; ++co) acc[co] = bias[co];
        // Branch-free taps: an out-of-image tap reads a clamped (valid) address and its values are zeroed by a select.
        // With a divergent `continue` per tap the compiler waits for every tap's loads before it issues the next ones --
        // nine dependent L2 round trips per pixel, which is what bounded these layers at 1.2-2.3 TB/s; straight-line code
        // lets it put a whole row of taps (or all nine) in flight at once.
        const float* centre = a.in + (((long)b * a.H + oy * S) * a.W + ox * S) * a.in_ld;
#pragma unroll
        for (int ky = 0; ky < KH; ++ky) {
            float xin[KW][CIN];
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                const int dy = ky - P, dx = kx - P;
                const int iy = oy * S + dy, ix = ox * S + dx;
                const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const int cy = min(max(iy, 0), a.H - 1) - oy * S, cx = min(max(ix, 0), a.W - 1) - ox * S;
                const float* src = centre + (cy * a.W + cx) * a.in_ld;
                if constexpr (CIN % 4 == 0) {
#pragma unroll
                    for (int c = 0; c < CIN; c += 4) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
                        xin[kx][c] = ok ? v[0] : 0.f; xin[kx][c + 1] = ok ? v[1] : 0.f;
                        xin[kx][c + 2] = ok ? v[2] : 0.f; xin[kx][c + 3] = ok ? v[3] : 0.f;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < CIN; ++c) { const float v = src[c]; xin[kx][c] = ok ? v : 0.f; }
                }
            }
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                const float* w = a.wplain + (ky * KW + kx) * CIN * COUT + c0;
#pragma unroll
                for (int c = 0; c < CIN; ++c)
#pragma unroll
                    for (int co = 0; co < CPT; ++co) {
                        const float wv = WREG ? wreg[((ky * KW + kx) * CIN + c) * CPT + co] : w[c * COUT + co];
                        acc[co] = fmaf(xin[kx][c], wv, acc[co]);
                    }
            }
        }
        float* o = a.out + (long)p * a.out_ld + c0;
        if (a.accumulate) {
#pragma unroll
            for (int co = 0; co < CPT; ++co) acc[co] += o[co];
        }
        if constexpr (TPP == 1 && CPT % 4 == 0) {
            // the wave's 64 pixels are contiguous in memory: transpose through LDS so that every store instruction
            // writes 1 KiB contiguous instead of 64 scattered 16-byte pieces
            __shared__ __attribute__((aligned(16))) float tr[4][64 * CPT];
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const unsigned pw = p - lane;                       // first pixel of the wave
            const bool whole = a.out_ld == COUT && pw + 64 <= (unsigned)a.npix;   // wave-uniform
            if (whole) {
#pragma unroll
                for (int co = 0; co < CPT; co += 4)
                    *reinterpret_cast<f32x4*>(&tr[wave][lane * CPT + co]) = (f32x4){acc[co], acc[co + 1], acc[co + 2], acc[co + 3]};
                float* ob = a.out + (long)pw * COUT;
#pragma unroll
                for (int k = 0; k < CPT / 4; ++k)
                    *reinterpret_cast<f32x4*>(ob + k * 256 + lane * 4) = *reinterpret_cast<const f32x4*>(&tr[wave][k * 256 + lane * 4]);
            } else if ((a.out_ld & 3) == 0) {
#pragma unroll
                for (int co = 0; co < CPT; co += 4)
                    *reinterpret_cast<f32x4*>(o + co) = (f32x4){acc[co], acc[co + 1], acc[co + 2], acc[co + 3]};
            } else {
#pragma unroll
                for (int co = 0; co < CPT; ++co) o[co] = acc[co];
            }
        } else if (CPT % 4 == 0 && (a.out_ld & 3) == 0) {
#pragma unroll
            for (int co = 0; co < CPT; co += 4)
                *reinterpret_cast<f32x4*>(o + co) = (f32x4){acc[co], acc[co + 1], acc[co + 2], acc[co + 3]};
        } else {
#pragma unroll
            for (int co = 0; co < CPT; ++co) o[co] = acc[co];
        }
#pragma unroll
        for (int co = 0; co < CPT; ++co) { st1[co] += acc[co]; st2[co] = fmaf(acc[co], acc[co], st2[co]); }
    }
    if constexpr (COUT >= 4) {
        // fused BatchNorm statistics: fold over the threads that own the same channels (wave shuffles when a thread owns
        // all channels, LDS otherwise), then one fp64 atomic per channel and workgroup (the host caps the grid)
        if (a.bn_sums) {
            __shared__ float red[TPP > 1 ? 256 * 2 * CPT : 4 * 2 * COUT];
            if constexpr (TPP > 1) {
#pragma unroll
                for (int co = 0; co < CPT; ++co) {
                    red[(threadIdx.x * CPT + co) * 2] = st1[co];
                    red[(threadIdx.x * CPT + co) * 2 + 1] = st2[co];
                }
            } else {
                const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
                for (int co = 0; co < CPT; ++co) {
                    const float u = wave_sum(st1[co]), q = wave_sum(st2[co]);
                    if (lane == 0) { red[(wave * COUT + co) * 2] = u; red[(wave * COUT + co) * 2 + 1] = q; }
                }
            }
            __syncthreads();
            if (threadIdx.x < COUT * 2) {
                const int ch = threadIdx.x >> 1, which = threadIdx.x & 1;
                double d = 0.0;
                if constexpr (TPP > 1) {
                    const int quad = ch / CPT, co = ch - quad * CPT;
                    for (int k = quad; k < 256; k += TPP) d += (double)red[(k * CPT + co) * 2 + which];
                } else {
                    for (int w = 0; w < 4; ++w) d += (double)red[(w * COUT + ch) * 2 + which];
                }
                atomicAdd(&a.bn_sums[(blockIdx.x % RV_BN_NREP) * 2 * COUT + which * COUT + ch], d);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// weight gradient: G[tap][a][b] = sum_p U[f(p,tap)][a] * V[p][b]   (+ colsum of V for the bias)
// ------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* U; int u_ld, Hu, Wu, Ca;    // gathered tensor (conv input, or dY for the up-conv)
    const float* V; int v_ld, Hv, Wv, Cb;    // dense tensor, one MFMA k per V pixel
    int B;
    float* part; long pstride;               // partial sums [nparts][taps*Ca*Cb (+Cb)]
    int rows_per_wave, nparts, ngb;          // ngb = number of b-groups
    int want_bias;
    int ablate;                              // -DRV_ABLATION builds only (timing experiments, wrong results): 1 no MFMAs,
                                             // 2 no fragment reads, 4 stage the first row only, 8 no row barriers
    unsigned long long* dbg;                 // -DRV_ABLATION: s_memtime stamps of workgroup (0,0) / wave 0 (RV_DBG_PTR)
    FastDiv fd_vplane, fd_wv;                // divide by Hv*Wv, by Wv (small-channel kernel)
    unsigned u_bytes, v_bytes;               // wgrad_wino_k: bytes of ONE SEGMENT's U / V view (buffer-resource ranges of its staging loads; < 0x3f000000)
    // Segments (rv_conv_wgrad_seg): the B images are nseg runs of Bseg images, run s at Useg[s] / Vseg[s] (same geometry and strides) --
    // the same layer's (input, dY) pairs of several backward passes of one step reduced by ONE launch.  nseg == 1: Useg[0] = U, Bseg = B.
    const float* Useg[4]; const float* Vseg[4];
    int nseg, Bseg;
};

// image b -> (segment, image inside it); b is wave-uniform
__device__ __forceinline__ int wgrad_seg_of(const WgradArgs& a, int b, int& bl) {
    const int sg = (b >= a.Bseg ? 1 : 0) + (b >= 2 * a.Bseg ? 1 : 0) + (b >= 3 * a.Bseg ? 1 : 0);
    bl = b - sg * a.Bseg;
    return sg;
}

// LDS-staged pixel-reduction GEMM.  One workgroup owns a run of V rows (b, y) and one (a-group, b-group)
// of up to 32x32 channels.  Per V row it stages that row (CB channels) and the KH input rows it touches
// (CA channels, front/back zero padding so the tap shifts need no predicates) into LDS with 16-byte
// loads -- for the 3x3 convs the KH-row window slides, so every input row is fetched once per workgroup
// -- and the four waves walk the row in 4-pixel MFMA k-steps (wave w takes steps w, w+4, ...), reading
// A/B fragments with ds_read_b32 (lane (i, g): channel i of pixel x0+g, which is what the f32 MFMA wants).
// Accumulators stay in registers across all rows; at the end the four waves are folded through LDS and
// ONE partial per workgroup goes to the workspace (deterministic second pass: wgrad_reduce_k).
// BF (opt-in experiment, BASELINE config 3): both operands are rounded to bf16 on their way from LDS to the matrix pipe and one
// v_mfma_f32_16x16x16_bf16 covers a k-step of SIXTEEN pixels: lane (i, g) supplies channel i of pixels x0 + g + 4e, e = 0..3 (any
// k <-> pixel assignment works as long as both operands use the same one; this one keeps the bank pattern of the f32 kernel).  Rows
// are padded to a multiple of 16 pixels in LDS; accumulation, the bias gradient and the fold stay fp32.
template <int KH, int KW, int S, int P, int TA, int TB, int NW, bool BF = false>
__global__ __launch_bounds__(NW * 64) void wgrad_mfma_k(WgradArgs a) {
    constexpr int TAPS = KH * KW, CA = TA * 16, CB = TB * 16;
    // 8-wave workgroups put two waves on every SIMD.  With the full 32x32 channel group the accumulators of one
    // wave would not leave room for a second one, so there the waves are split 4 (x groups) x 2 (halves of the
    // a-tiles); otherwise all NW waves are x groups.
    constexpr int NH = (NW == 8 && TA == 2 && TB == 2) ? 2 : 1;
    constexpr int TAW = TA / NH;             // a-tiles owned by one wave
    constexpr int NXG = NW / NH;             // x groups
    constexpr int NTHR = NW * 64;
    constexpr int C4A = CA / 4, C4B = CB / 4;
    constexpr int NSLOT = (S == 1) ? KH + 1 : 2 * KH;      // input-row ring: one slot ahead of the rows in use
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int ga = blockIdx.y / a.ngb, gb = blockIdx.y - ga * a.ngb;
    const int a0 = ga * CA, b0 = gb * CB;
    const int nrows = a.B * a.Hv;
    const int row0 = blockIdx.x * a.rows_per_wave;          // rows per WORKGROUP here
    const int row1 = min(row0 + a.rows_per_wave, nrows);
    const int Wv4 = BF ? ((a.Wv + 15) & ~15) : ((a.Wv + 3) & ~3);
    const int UP = S * (Wv4 - 1) + KW;                       // pixels per staged input row (incl. zero padding)
    float* ubuf = smem;                                      // [NSLOT][UP][CA]
    float* vbuf = smem + NSLOT * UP * CA;                    // [2][Wv4][CB]
    const int stage_floats = NSLOT * UP * CA + 2 * Wv4 * CB;

    // The waves of one SIMD are wave, wave + 4: with two halves the second half's x groups are rotated by two, so that the x
    // groups that own one k-step more per row (8/7/7/7 of the 29 steps of a 114-pixel row) do not share a SIMD (-2 % there).
    const int hp = wave / NXG, xg = (wave + (NH == 2 ? 2 * hp : 0)) % NXG;
    f32x4 acc[TAPS][TAW][TB];
    f32x4 accb[TB];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int x = 0; x < TAW; ++x)
#pragma unroll
            for (int y = 0; y < TB; ++y) acc[t][x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int y = 0; y < TB; ++y) accb[y] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = a.want_bias && ga == 0 && hp == 0;
    float accbs[TB];                                         // per-lane partial column sums of V (lane (i, g): pixels x0+g)
#pragma unroll
    for (int y = 0; y < TB; ++y) accbs[y] = 0.f;

#ifdef RV_ABLATION
    const bool stamp = a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
    unsigned long long t_wait = 0, t_comp = 0, t_mark = 0;
#define TS(k) do { if (stamp) a.dbg[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TS(k) do { } while (0)
#endif
    TS(0);
    if (ABL(a) & 64) return;
    if (!(ABL(a) & 16))
    for (int k = tid; k < stage_floats; k += NTHR) smem[k] = 0.f;   // padding / unused channels stay zero
    __syncthreads();

    TS(1);
    const int ca_valid = min(CA, a.Ca - a0), cb_valid = min(CB, a.Cb - b0);
    const int npxu = min(a.Wu, UP - P);

    // LDS-DMA of one input row (image b, row r) into ring slot `slot`; rows outside the image are zero-filled
    auto dma_urow = [&](int b, int r, int slot) {
        constexpr int PPI = 64 / C4A;
        float* dst0 = ubuf + slot * UP * CA + P * CA;
        const bool inside = r >= 0 && r < a.Hu;
        int bl;
        const int sg = wgrad_seg_of(a, b, bl);
        const float* src = a.Useg[sg] + ((long)bl * a.Hu + (inside ? r : 0)) * a.Wu * a.u_ld + a0;
        const int px_l = lane / C4A, c4 = (lane - px_l * C4A) * 4;
        for (int k = wave; k * PPI < npxu; k += NW) {
            const int px = k * PPI + px_l;
            float* ldst = dst0 + k * PPI * CA;                           // wave-uniform
            if (px < npxu && c4 < ca_valid) {
                if (inside) glds16(src + (long)px * a.u_ld + c4, ldst);
                else *reinterpret_cast<f32x4*>(ldst + lane * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto dma_vrow = [&](int b, int y, int vb) {
        constexpr int PPI = 64 / C4B;
        float* dst0 = vbuf + vb * Wv4 * CB;
        int bl;
        const int sg = wgrad_seg_of(a, b, bl);
        const float* src = a.Vseg[sg] + ((long)bl * a.Hv + y) * a.Wv * a.v_ld + b0;
        const int px_l = lane / C4B, c4 = (lane - px_l * C4B) * 4;
        for (int k = wave; k * PPI < a.Wv; k += NW) {
            const int px = k * PPI + px_l;
            if (px < a.Wv && c4 < cb_valid) glds16(src + (long)px * a.v_ld + c4, dst0 + k * PPI * CB);
        }
    };
    auto slot_of_row = [&](int y, int ky) -> int {
        if (S == 1) return (y - P + ky + 4 * NSLOT) % NSLOT;
        return (y & 1) * KH + ky;
    };
    auto stage_full = [&](int b, int y, int vb) {
#pragma unroll
        for (int ky = 0; ky < KH; ++ky) dma_urow(b, y * S - P + ky, slot_of_row(y, ky));
        dma_vrow(b, y, vb);
    };
    auto stage_next = [&](int b, int y, int vb) {            // rows of (b, y) not already resident for (b, y-1)
        if (S == 1) dma_urow(b, y - P + KH - 1, slot_of_row(y, KH - 1));
        else {
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) dma_urow(b, y * S - P + ky, slot_of_row(y, ky));
        }
        dma_vrow(b, y, vb);
    };

    bool prefetched = false;
    for (int row = row0; row < row1; ++row) {
        const int b = row / a.Hv, y = row - b * a.Hv;
        const int vb = row & 1;
#ifdef RV_ABLATION
        if (stamp) t_mark = __builtin_amdgcn_s_memtime();
#endif
        if (!prefetched && !((ABL(a) & 4) && row > row0)) {
            if (!(ABL(a) & 8)) __syncthreads();               // nobody still reads the ring
            stage_full(b, y, vb);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(ABL(a) & 8)) __syncthreads();                   // row (b, y) is resident for every wave
        prefetched = (row + 1 < row1) && (y + 1 < a.Hv);
#ifdef RV_ABLATION
        if (stamp) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_wait += t - t_mark; t_mark = t; if (row == row0) a.dbg[2] = t; }
#endif
        if (prefetched && !(ABL(a) & 4)) stage_next(b, y + 1, vb ^ 1);         // DMA under the MFMAs below
#ifdef RV_ABLATION
        if (stamp) { const unsigned long long t = __builtin_amdgcn_s_memtime(); a.dbg[8] += t - t_mark; t_mark = t; }
#endif
        int slot_of[KH];
#pragma unroll
        for (int ky = 0; ky < KH; ++ky) slot_of[ky] = slot_of_row(y, ky);
        const float* vrow = vbuf + vb * Wv4 * CB;
        // k-steps of 4 pixels, software-pipelined in the source: the fragments of step s+1 are loaded (plain LDS loads the
        // compiler counts) into the other register set while step s is multiplied, and sched_group_barrier pins an
        // MFMA / ds_read interleave.  Left alone, the compiler put every ds_read directly in front of the MFMAs that consume
        // it with s_waitcnt lgkmcnt(0) in between -- six exposed LDS round trips per 18 MFMAs, SQ_WAIT_ANY 33 %, matrix pipe
        // busy 49 % (profiles/r02_pmc_wgrad.txt).
        float vfA[TB], ufA[TAPS][TAW], vfB[TB], ufB[TAPS][TAW];
        auto ld = [&](float (&vf)[TB], float (&uf)[TAPS][TAW], const int x0) {
            const int x = x0 + g;
            if (ABL(a) & 2) {
#pragma unroll
                for (int tb = 0; tb < TB; ++tb) vf[tb] = (float)x0;
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int ta = 0; ta < TAW; ++ta) uf[t][ta] = (float)(x0 + t);
                return;
            }
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) vf[tb] = vrow[x * CB + tb * 16 + i];
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) {
                const float* ur = ubuf + slot_of[ky] * UP * CA + (S * x) * CA + i;
#pragma unroll
                for (int kx = 0; kx < KW; ++kx)
#pragma unroll
                    for (int ta = 0; ta < TAW; ++ta) {
#ifdef RV_ABLATION
                        if ((ABL(a) & 256) && ky * KW + kx >= 4) { uf[ky * KW + kx][ta] = 0.f; continue; }   // probe: 4 + TB reads per k-step
#endif
                        uf[ky * KW + kx][ta] = ur[kx * CA + (hp * TAW + ta) * 16];
                    }
            }
        };
        auto mm = [&](const float (&vf)[TB], const float (&uf)[TAPS][TAW]) {
            if (ABL(a) & 1) {                                 // keep the operands live without the matrix pipe
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int ta = 0; ta < TAW; ++ta) acc[t][ta][0][0] += uf[t][ta] * vf[0] + vf[TB - 1];
                return;
            }
#ifdef RV_ABLATION
            if (ABL(a) & 256) {                               // Winograd F(3x3, 2x2) cost probe: 4 of 9 MFMA groups per k-step (16 per 4 tiles) ...
                float vv[TB];
#pragma unroll
                for (int tb = 0; tb < TB; ++tb) vv[tb] = vf[tb];
                if (ABL(a) & 512) {                           // ... plus the transform adds of both operands (56 per 4 tiles = 14 per k-step)
#pragma unroll
                    for (int q = 0; q < 14; ++q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(vv[q % TB]) : "v"(uf[q % TAPS][0]));
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int ta = 0; ta < TAW; ++ta)
#pragma unroll
                        for (int tb = 0; tb < TB; ++tb)
                            acc[t][ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[t][ta], vv[tb], acc[t][ta][tb], 0, 0, 0);
                return;
            }
#endif
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
#pragma unroll
                for (int ta = 0; ta < TAW; ++ta)
#pragma unroll
                    for (int tb = 0; tb < TB; ++tb)
                        acc[t][ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[t][ta], vf[tb], acc[t][ta][tb], 0, 0, 0);
            if (do_bias) {
                // bias gradient on the VALU (co-issues with the MFMAs): only the ga == 0 workgroups carry it, and as two
                // extra MFMAs per step it made exactly those workgroups -- hence the whole launch -- ~10 % longer
#pragma unroll
                for (int tb = 0; tb < TB; ++tb) accbs[tb] += vf[tb];
            }
        };
        auto interleave = [&]() {                             // one ds_read after each of the first MFMAs of the block
#pragma unroll
            for (int k = 0; k < TAPS * TAW * TB; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (k < TB + TAPS * TAW) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        };
        constexpr int DX = 4 * NXG;
        if constexpr (BF) {
            for (int xb0 = xg * 16; xb0 < Wv4; xb0 += 16 * NXG) {
                const int x = xb0 + g;
                float vraw[TB][4], uraw[TAPS][TAW][4];
#pragma unroll
                for (int tb = 0; tb < TB; ++tb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) vraw[tb][e] = vrow[(x + 4 * e) * CB + tb * 16 + i];
#pragma unroll
                for (int ky = 0; ky < KH; ++ky) {
                    const float* ur = ubuf + slot_of[ky] * UP * CA + (S * x) * CA + i;
#pragma unroll
                    for (int kx = 0; kx < KW; ++kx)
#pragma unroll
                        for (int ta = 0; ta < TAW; ++ta)
#pragma unroll
                            for (int e = 0; e < 4; ++e) uraw[ky * KW + kx][ta][e] = ur[(S * 4 * e + kx) * CA + (hp * TAW + ta) * 16];
                }
                s16x4 vb16[TB], ub16[TAPS][TAW];
#pragma unroll
                for (int tb = 0; tb < TB; ++tb) vb16[tb] = to_bf16x4((f32x4){vraw[tb][0], vraw[tb][1], vraw[tb][2], vraw[tb][3]});
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int ta = 0; ta < TAW; ++ta)
                        ub16[t][ta] = to_bf16x4((f32x4){uraw[t][ta][0], uraw[t][ta][1], uraw[t][ta][2], uraw[t][ta][3]});
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int ta = 0; ta < TAW; ++ta)
#pragma unroll
                        for (int tb = 0; tb < TB; ++tb)
                            acc[t][ta][tb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ub16[t][ta], vb16[tb], acc[t][ta][tb], 0, 0, 0);
                if (do_bias) {
#pragma unroll
                    for (int tb = 0; tb < TB; ++tb) accbs[tb] += (vraw[tb][0] + vraw[tb][1]) + (vraw[tb][2] + vraw[tb][3]);
                }
            }
            continue;
        }
        // the prefetch of a step past the end of the row reads the row's last (valid, staged) k-step instead: no branch
        // around the loads, so each [prefetch ; multiply] pair stays one straight-line block the scheduler can interleave
        const int xlast = Wv4 - 4;
        int x0 = xg * 4;
        if (x0 < Wv4) {
            ld(vfA, ufA, x0);
            for (;;) {
                const int x1 = x0 + DX;
                ld(vfB, ufB, min(x1, xlast));
                mm(vfA, ufA);
                interleave();
                if (x1 >= Wv4) break;
                x0 = x1 + DX;
                ld(vfA, ufA, min(x0, xlast));
                mm(vfB, ufB);
                interleave();
                if (x0 >= Wv4) break;
            }
        }
#ifdef RV_ABLATION
        if (stamp) t_comp += __builtin_amdgcn_s_memtime() - t_mark;
#endif
    }
#ifdef RV_ABLATION
    if (stamp) { a.dbg[6] = t_wait; a.dbg[7] = t_comp; }
#endif
    TS(3);
    if (do_bias) {                                           // pixel lanes g -> every lane of a channel holds the sum
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            float sb = accbs[tb];
            sb += __shfl_xor(sb, 16, 64);
            sb += __shfl_xor(sb, 32, 64);
            accb[tb] = (f32x4){sb, sb, sb, sb};
        }
    }
    if (ABL(a) & 32) { if (acc[0][0][0][0] == 123.456f) a.part[0] = accbs[0]; return; }
    // fold the x groups through LDS as a binary tree: in every round the upper half of the surviving x groups publishes, the lower
    // half accumulates (log2(NXG) rounds of two barriers instead of NXG - 1: the linear fold was 2.8 us of every launch)
    constexpr int NVEC = TAPS * TAW * TB + TB;              // accumulator vectors per lane: one 16-byte LDS access each
    static_assert((NXG & (NXG - 1)) == 0, "x groups must be a power of two");
#pragma unroll
    for (int half = NXG / 2; half >= 1; half >>= 1) {
        f32x4* fold = reinterpret_cast<f32x4*>(smem) + (hp * (NXG / 2) + (xg & (half - 1))) * NVEC * 64 + lane;
        __syncthreads();
        if (xg >= half && xg < 2 * half) {
            int q = 0;
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
#pragma unroll
                for (int ta = 0; ta < TAW; ++ta)
#pragma unroll
                    for (int tb = 0; tb < TB; ++tb) fold[(q++) * 64] = acc[t][ta][tb];
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) fold[(q++) * 64] = accb[tb];
        }
        __syncthreads();
        if (xg < half) {
            int q = 0;
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
#pragma unroll
                for (int ta = 0; ta < TAW; ++ta)
#pragma unroll
                    for (int tb = 0; tb < TB; ++tb) acc[t][ta][tb] += fold[(q++) * 64];
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) accb[tb] += fold[(q++) * 64];
        }
    }
    TS(4);
    if (xg != 0) return;
    // D[row = a_local = 4g+r][col = b_local = i]
    float* dst = a.part + (long)blockIdx.x * a.pstride;
#ifdef RV_ABLATION
    if (ABL(a) & 2048) {                                     // cost probe: every row partition ADDS into partial 0 (fp32 atomics) instead of storing its own
        dst = a.part;
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int ta = 0; ta < TAW; ++ta)
#pragma unroll
                for (int tb = 0; tb < TB; ++tb) {
                    const int bb = b0 + tb * 16 + i;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int aa = a0 + (hp * TAW + ta) * 16 + 4 * g + r;
                        if (aa < a.Ca && bb < a.Cb) atomicAdd(&dst[((long)t * a.Ca + aa) * a.Cb + bb], acc[t][ta][tb][r]);
                    }
                }
        return;
    }
#endif
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int ta = 0; ta < TAW; ++ta)
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                const int bb = b0 + tb * 16 + i;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int aa = a0 + (hp * TAW + ta) * 16 + 4 * g + r;
                    if (aa < a.Ca && bb < a.Cb) dst[((long)t * a.Ca + aa) * a.Cb + bb] = acc[t][ta][tb][r];
                }
            }
    if (do_bias && g == 0) {
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            const int bb = b0 + tb * 16 + i;
            if (bb < a.Cb) dst[(long)TAPS * a.Ca * a.Cb + bb] = accb[tb][0];
        }
    }
    TS(5);
#undef TS
}

// ------------------------------------------------------------------------------------------
// Winograd F(3x3, 2x2) form of the 3x3 weight gradient (round 4; plan code nw = 24 of rv_conv_wgrad_set_plan).
//
//   dW = G^T [ sum_tiles (B^T d B) .* (A dY A^T) ] G        per 4x4 input patch d and 2x2 tile dY of the output gradient
//
// 16 products per tile (4 output pixels) and channel pair instead of 36: 2.25x fewer MFMAs than the direct form.  The MFMA contraction
// runs over TILES (k = 4 consecutive tiles of a tile row per instruction): lane (i, g) holds channel i of tile g, reads the 16 patch
// values of its (input channel, tile) and the 4 dY values of its (output channel, tile) from LDS (4-byte reads), transforms both in
// registers (32 + 12 adds) and issues one MFMA per xi and (a-tile, b-tile) pair.  The accumulators hold M[xi] (16 x 16 per tile pair);
// the 4x4 -> 3x3 transform G^T M G is lane-local and applied before the fold, so the partial sums are the direct kernel's [9][Ca][Cb].
// Work partition, staging ring, fold and partial-sum stores follow wgrad_mfma_k, with ROW PAIRS instead of rows: one barrier interval
// covers two output rows (one tile row); the input ring holds 6 rows (4 in use, 2 arriving), dY is double-buffered two rows at a time.
// LDS images are permuted inside each 1 KiB DMA piece so that the two tiles of a 32-lane read group sit on different bank halves
// (pixel-major images put them exactly 128 or 256 bytes apart: a 2-way conflict on every read, as in the direct kernel):
//   32-channel groups: the 16-byte quad q of pixel X is stored at quad position q ^ (4 * ((X >> 1) & 1))
//   16-channel groups: pixel X is stored in pixel slot X ^ ((X >> 1) & 1)
// Rows / columns outside the image and unused channels are out-of-range offsets of the buffer resources the rows are staged through (the
// hardware writes zeros for them): every piece of every staged row is written by one unmasked DMA instruction, nothing is zero-filled by stores.
// ------------------------------------------------------------------------------------------
template <int C> __device__ __forceinline__ int wgw_pos(int X, int c) {          // float offset of channel c of pixel X inside a staged row
    return C == 32 ? X * 32 + ((((c >> 4) ^ ((X >> 1) & 1)) << 4) | (c & 15)) : (X ^ ((X >> 1) & 1)) * 16 + c;
}

template <int TA, int TB, int NW>
__global__ __launch_bounds__(NW * 64) void wgrad_wino_k(WgradArgs a) {
    constexpr int CA = TA * 16, CB = TB * 16;
    constexpr int NH = (NW == 8 && TA == 2 && TB == 2) ? 2 : 1;
    constexpr int TAW = TA / NH, NXG = NW / NH;
    constexpr int C4A = CA / 4, C4B = CB / 4;
    constexpr int PPA = 64 / C4A, PPB = 64 / C4B;            // pixels per DMA piece
    constexpr int NSLOT = 6;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int ga = blockIdx.y / a.ngb, gb = blockIdx.y - ga * a.ngb;
    const int a0 = ga * CA, b0 = gb * CB;
    const int H2 = a.Hv >> 1;                                // tile rows per image (Hv even: checked by the host)
    const int npairs = a.B * H2;
    const int pair0 = blockIdx.x * a.rows_per_wave;          // row PAIRS per workgroup here
    const int pair1 = min(pair0 + a.rows_per_wave, npairs);
    const int WT = (a.Wv + 1) >> 1, WT4 = (WT + 3) & ~3;     // tiles per row, rounded to whole k-steps
    const int UPp = ((2 * WT4 + 2 + PPA - 1) / PPA) * PPA;   // staged input pixels per row (X = x + 1 = 0 .. 2 WT4 + 1), whole pieces
    const int VPp = ((2 * WT4 + PPB - 1) / PPB) * PPB;       // staged dY pixels per row
    float* ubuf = smem;                                      // [NSLOT][UPp][CA]
    float* vbuf = smem + NSLOT * UPp * CA;                   // [2][2][VPp][CB]
    const int hp = wave / NXG, xg = (wave + (NH == 2 ? 2 * hp : 0)) % NXG;

    f32x4 acc[16][TAW][TB];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int x = 0; x < TAW; ++x)
#pragma unroll
            for (int y = 0; y < TB; ++y) acc[xi][x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 accb[TB];
    float accbs[TB];
#pragma unroll
    for (int y = 0; y < TB; ++y) { accb[y] = (f32x4){0.f, 0.f, 0.f, 0.f}; accbs[y] = 0.f; }
    const bool do_bias = a.want_bias && ga == 0 && hp == 0;
    const int ca_valid = min(CA, a.Ca - a0), cb_valid = min(CB, a.Cb - b0);

    // ---- staging: which (pixel, quad) this lane fetches for piece k of a row (inverse of the LDS permutation).  Rows travel through
    // buffer resources over the U / V views (as in conv3x3_wino_k): a halo column, a padded channel quad and a row outside the image are
    // all just an offset bump of OOB, which the hardware turns into zeros in LDS.  The per-lane byte offsets of a wave's pieces
    // (k = wave, wave + NW, ...) do not depend on the row: computed once. ----
    constexpr unsigned OOB = 0x40000000u;
    constexpr int MAXP = 4;                                  // pieces per row and wave planned in registers (any further: generic loop)
    const int upl = lane / C4A, uql = lane - upl * C4A;      // slot pixel inside the piece, quad position
    const int vpl = lane / C4B, vql = lane - vpl * C4B;
    auto u_off = [&](int k) -> unsigned {
        const int Xs = k * PPA + upl;                                            // pixel slot
        const int X = CA == 32 ? Xs : (Xs ^ ((Xs >> 1) & 1));                    // source pixel (+1)
        const int q = CA == 32 ? (uql ^ (4 * ((X >> 1) & 1))) : uql;             // source quad
        const int x = X - 1;
        return ((unsigned)x < (unsigned)a.Wu && q * 4 < ca_valid) ? (unsigned)(x * a.u_ld + q * 4) * 4u : OOB;
    };
    auto v_off = [&](int k) -> unsigned {
        const int Xs = k * PPB + vpl;
        const int X = CB == 32 ? Xs : (Xs ^ ((Xs >> 1) & 1));
        const int q = CB == 32 ? (vql ^ (4 * ((X >> 1) & 1))) : vql;
        return (X < a.Wv && q * 4 < cb_valid) ? (unsigned)(X * a.v_ld + q * 4) * 4u : OOB;
    };
    unsigned u_goff[MAXP], v_goff[MAXP];
#pragma unroll
    for (int j = 0; j < MAXP; ++j) { u_goff[j] = u_off(wave + NW * j); v_goff[j] = v_off(wave + NW * j); }
    auto dma_urow = [&](int b, int r, int slot) {            // input row r of image b -> ring slot
        float* dst0 = ubuf + slot * UPp * CA;
        int bl;
        const int sg = wgrad_seg_of(a, b, bl);
        const rv_rsrc_t rs_u = rv_make_rsrc(a.Useg[sg], a.u_bytes);       // (the resource covers the image's own segment)
        const unsigned base = (unsigned)r < (unsigned)a.Hu ? (unsigned)(((bl * a.Hu + r) * a.Wu) * a.u_ld + a0) * 4u : OOB;
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const int k = wave + NW * j;
            if (k * PPA >= UPp) break;
            rv_buf_lds16(rs_u, dst0 + k * PPA * CA, u_goff[j] + base);
        }
        for (int k = wave + NW * MAXP; k * PPA < UPp; k += NW) rv_buf_lds16(rs_u, dst0 + k * PPA * CA, u_off(k) + base);
    };
    auto dma_vrow = [&](int b, int y, int vslot) {           // dY row y of image b -> vbuf row slot (0..3)
        float* dst0 = vbuf + vslot * VPp * CB;
        int bl;
        const int sg = wgrad_seg_of(a, b, bl);
        const rv_rsrc_t rs_v = rv_make_rsrc(a.Vseg[sg], a.v_bytes);
        const unsigned base = (unsigned)(((bl * a.Hv + y) * a.Wv) * a.v_ld + b0) * 4u;
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const int k = wave + NW * j;
            if (k * PPB >= VPp) break;
            rv_buf_lds16(rs_v, dst0 + k * PPB * CB, v_goff[j] + base);
        }
        for (int k = wave + NW * MAXP; k * PPB < VPp; k += NW) rv_buf_lds16(rs_v, dst0 + k * PPB * CB, v_off(k) + base);
    };
    // tile row yp of an image needs input rows 2 yp - 1 .. 2 yp + 2; input row r lives in ring slot (r + 1) % 6
    auto uslot = [&](int r) -> int { return (r + 1 + NSLOT) % NSLOT; };
    auto stage_full = [&](int b, int yp, int vb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dma_urow(b, 2 * yp - 1 + e, uslot(2 * yp - 1 + e));
        dma_vrow(b, 2 * yp, vb * 2);
        dma_vrow(b, 2 * yp + 1, vb * 2 + 1);
    };
    auto stage_next = [&](int b, int yp, int vb) {           // the two input rows tile row yp - 1 did not need
        dma_urow(b, 2 * yp + 1, uslot(2 * yp + 1));
        dma_urow(b, 2 * yp + 2, uslot(2 * yp + 2));
        dma_vrow(b, 2 * yp, vb * 2);
        dma_vrow(b, 2 * yp + 1, vb * 2 + 1);
    };

    // ---- per-lane fragment offsets (floats) relative to the first pixel of a k-step (4 tiles = 8 pixels; k-steps start at tiles that
    // are multiples of 4, so the permutation bits of a pixel depend on (g, ec) only) ----
    int uoff[4][TAW], voff[2][TB];
#pragma unroll
    for (int ec = 0; ec < 4; ++ec)
#pragma unroll
        for (int ta = 0; ta < TAW; ++ta) uoff[ec][ta] = wgw_pos<CA>(2 * g + ec, (hp * TAW + ta) * 16 + i);
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) voff[dx][tb] = wgw_pos<CB>(2 * g + dx, tb * 16 + i);

    bool prefetched = false;
    for (int pair = pair0; pair < pair1; ++pair) {
        const int b = pair / H2, yp = pair - b * H2;
        const int vb = pair & 1;
        if (!prefetched) {
            __syncthreads();                                  // nobody still reads the ring
            stage_full(b, yp, vb);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                      // tile row (b, yp) is resident for every wave
        prefetched = (pair + 1 < pair1) && (yp + 1 < H2);
        if (prefetched) stage_next(b, yp + 1, vb ^ 1);        // DMA under the MFMAs below
        const float* urow[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) urow[e] = ubuf + uslot(2 * yp - 1 + e) * UPp * CA;
        const float* vrow0 = vbuf + (vb * 2) * VPp * CB;
        const float* vrow1 = vrow0 + VPp * CB;
        // (a source-level software pipeline of the k-steps -- raw operands of step s + 1 loaded under the MFMAs of step s, as in
        // wgrad_mfma_k -- was measured 2-6 % slower than the schedule the compiler finds for this plain loop)
        // Fragment pointers of this wave's FIRST k-step of the tile row; k-step s of the wave reads at the compile-time distance
        // s * USTEP / s * VSTEP floats behind them (an instruction immediate), so the up-to-KMAX steps of a row are unrolled and cost
        // no address arithmetic at all (the rolled loop spent 55 vector adds per 32 MFMAs on it).
        constexpr int USTEP = 2 * 4 * NXG * CA, VSTEP = 2 * 4 * NXG * CB, KMAX = 8;
        const float* up[4][4][TAW];
        const float* vp[4][TB];
#pragma unroll
        for (int er = 0; er < 4; ++er)
#pragma unroll
            for (int ec = 0; ec < 4; ++ec)
#pragma unroll
                for (int ta = 0; ta < TAW; ++ta) up[er][ec][ta] = urow[er] + 2 * (xg * 4) * CA + uoff[ec][ta];
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                vp[dx][tb] = vrow0 + 2 * (xg * 4) * CB + voff[dx][tb];
                vp[2 + dx][tb] = vrow1 + 2 * (xg * 4) * CB + voff[dx][tb];
            }
        auto kstep = [&](const int uo, const int vo) __attribute__((always_inline)) {       // k-step at float offsets (uo, vo) from the pointers
            float d[16][TAW], y[4][TB];
#pragma unroll
            for (int er = 0; er < 4; ++er)
#pragma unroll
                for (int ec = 0; ec < 4; ++ec)
#pragma unroll
                    for (int ta = 0; ta < TAW; ++ta) d[er * 4 + ec][ta] = up[er][ec][ta][uo];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int tb = 0; tb < TB; ++tb) y[e][tb] = vp[e][tb][vo];
            // P = B^T d B (rows, then columns)
#pragma unroll
            for (int ta = 0; ta < TAW; ++ta) {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const float e0 = d[cc][ta], e1 = d[4 + cc][ta], e2 = d[8 + cc][ta], e3 = d[12 + cc][ta];
                    d[cc][ta] = e0 - e2; d[4 + cc][ta] = e1 + e2; d[8 + cc][ta] = e2 - e1; d[12 + cc][ta] = e1 - e3;
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const float e0 = d[4 * rr][ta], e1 = d[4 * rr + 1][ta], e2 = d[4 * rr + 2][ta], e3 = d[4 * rr + 3][ta];
                    d[4 * rr][ta] = e0 - e2; d[4 * rr + 1][ta] = e1 + e2; d[4 * rr + 2][ta] = e2 - e1; d[4 * rr + 3][ta] = e1 - e3;
                }
            }
            // Q = A dY A^T, A = [1 0; 1 1; 1 -1; 0 -1]
            float q[16][TB];
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                const float y00 = y[0][tb], y01 = y[1][tb], y10 = y[2][tb], y11 = y[3][tb];
                const float t[4][2] = {{y00, y01}, {y00 + y10, y01 + y11}, {y00 - y10, y01 - y11}, {-y10, -y11}};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    q[4 * r][tb] = t[r][0]; q[4 * r + 1][tb] = t[r][0] + t[r][1]; q[4 * r + 2][tb] = t[r][0] - t[r][1]; q[4 * r + 3][tb] = -t[r][1];
                }
                if (do_bias) accbs[tb] += q[5][tb];          // Q[1][1] = the sum of the tile's four dY values
            }
#pragma unroll
            for (int xi = 0; xi < 16; ++xi)
#pragma unroll
                for (int ta = 0; ta < TAW; ++ta)
#pragma unroll
                    for (int tb = 0; tb < TB; ++tb)
                        acc[xi][ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[xi][ta], q[xi][tb], acc[xi][ta][tb], 0, 0, 0);
        };
        const int nk = (WT4 - xg * 4 + 4 * NXG - 1) / (4 * NXG);          // k-steps of this wave on this tile row
        sfor<0, KMAX>([&](auto sc) {
            constexpr int S = decltype(sc)::value;
            if (S < nk) kstep(S * USTEP, S * VSTEP);
        });
        for (int sk = KMAX; sk < nk; ++sk) kstep(sk * USTEP, sk * VSTEP);      // (rows wider than 8 k-steps per wave: rolled)
    }
    if (do_bias) {
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            float sb = accbs[tb];
            sb += __shfl_xor(sb, 16, 64);
            sb += __shfl_xor(sb, 32, 64);
            accb[tb] = (f32x4){sb, sb, sb, sb};
        }
    }
    if (ABL(a) & 32) { if (acc[0][0][0][0] == 123.456f) a.part[0] = accbs[0]; return; }     // (timing ablation: no fold / stores)
    // dW = G^T M G per (a, b) element, G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]: every lane holds all sixteen M[xi] of its elements, so
    // the 4x4 -> 3x3 transform is lane-local (linear: it commutes with the folds over x groups and workgroups that follow) and the
    // partial sums leave in the direct kernel's [9][Ca][Cb] layout -- 16/9 fewer bytes, the same reduction pass
    f32x4 tapv[9][TAW][TB];
#pragma unroll
    for (int ta = 0; ta < TAW; ++ta)
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            f32x4 t[3][4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const f32x4 h = 0.5f * (acc[4 + cc][ta][tb] + acc[8 + cc][ta][tb]);
                t[0][cc] = acc[cc][ta][tb] + h;
                t[1][cc] = 0.5f * (acc[4 + cc][ta][tb] - acc[8 + cc][ta][tb]);
                t[2][cc] = h + acc[12 + cc][ta][tb];
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const f32x4 h = 0.5f * (t[ky][1] + t[ky][2]);
                tapv[ky * 3 + 0][ta][tb] = t[ky][0] + h;
                tapv[ky * 3 + 1][ta][tb] = 0.5f * (t[ky][1] - t[ky][2]);
                tapv[ky * 3 + 2][ta][tb] = h + t[ky][3];
            }
        }
    // fold the x groups through LDS as a binary tree (as wgrad_mfma_k), one 16-byte LDS access per accumulator vector
    constexpr int NVEC = 9 * TAW * TB + TB;
    static_assert((NXG & (NXG - 1)) == 0, "x groups must be a power of two");
#pragma unroll
    for (int half = NXG / 2; half >= 1; half >>= 1) {
        f32x4* fold = reinterpret_cast<f32x4*>(smem) + (hp * (NXG / 2) + (xg & (half - 1))) * NVEC * 64 + lane;
        __syncthreads();
        if (xg >= half && xg < 2 * half) {
            int qn = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int ta = 0; ta < TAW; ++ta)
#pragma unroll
                    for (int tb = 0; tb < TB; ++tb) fold[(qn++) * 64] = tapv[t][ta][tb];
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) fold[(qn++) * 64] = accb[tb];
        }
        __syncthreads();
        if (xg < half) {
            int qn = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int ta = 0; ta < TAW; ++ta)
#pragma unroll
                    for (int tb = 0; tb < TB; ++tb) tapv[t][ta][tb] += fold[(qn++) * 64];
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) accb[tb] += fold[(qn++) * 64];
        }
    }
    if (xg != 0) return;
    // partial sums in the direct kernel's layout: [tap][a][b] (+ [b] bias); D[row = a_local = 4g + r][col = b_local = i]
    float* dst = a.part + (long)blockIdx.x * a.pstride;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ta = 0; ta < TAW; ++ta)
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                const int bb = b0 + tb * 16 + i;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int aa = a0 + (hp * TAW + ta) * 16 + 4 * g + r;
                    if (aa < a.Ca && bb < a.Cb) dst[((long)t * a.Ca + aa) * a.Cb + bb] = tapv[t][ta][tb][r];
                }
            }
    if (do_bias && g == 0) {
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            const int bb = b0 + tb * 16 + i;
            if (bb < a.Cb) dst[(long)9 * a.Ca * a.Cb + bb] = accb[tb][0];
        }
    }
}

// VALU version for tiny channel counts (Ca*Cb*taps <= 144).  One wave = one partial.  The wider channel dimension is
// split into quads across neighbouring lanes (LPP lanes per pixel), so a lane carries 36-72 accumulators instead of 144:
// several waves fit on a SIMD and hide the load latency, the dY row is read with coalesced 16-byte loads, and the final
// fold is a few xor-shuffles over the pixel slots.
template <int CA, int CB, int KH, int KW, int S, int P>
__global__ __launch_bounds__(256) void wgrad_small_k(WgradArgs a) {
    constexpr int TAPS = KH * KW;
    constexpr bool SPLIT_B = CB >= CA;                       // which channel dimension is spread over lanes
    constexpr int CS = SPLIT_B ? CB : CA;
    constexpr int Q = CS >= 4 ? 4 : CS;                      // channels of the split dimension per lane
    constexpr int LPP = CS / Q;                              // lanes per pixel (1, 2 or 4)
    constexpr int PPW = 64 / LPP;                            // pixels per wave trip
    constexpr int CAL = SPLIT_B ? CA : Q, CBL = SPLIT_B ? Q : CB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ps = lane / LPP, cq = lane - ps * LPP;
    const int a_off = SPLIT_B ? 0 : cq * Q, b_off = SPLIT_B ? cq * Q : 0;
    const int pw = blockIdx.x * 4 + wave;
    if (pw >= a.nparts) return;
    float acc[TAPS][CAL][CBL];
    float accb[CBL];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int x = 0; x < CAL; ++x)
#pragma unroll
            for (int y = 0; y < CBL; ++y) acc[t][x][y] = 0.f;
#pragma unroll
    for (int y = 0; y < CBL; ++y) accb[y] = 0.f;
    const unsigned npix = (unsigned)(a.B * a.Hv * a.Wv);
    const unsigned per = (npix + a.nparts - 1) / a.nparts;
    const unsigned p0 = (unsigned)pw * per, p1 = min(p0 + per, npix);
    const bool vvec = (CBL % 4 == 0) && (a.v_ld % 4 == 0) && ((((uintptr_t)a.V) & 15) == 0);
    const bool uvec = (CAL % 4 == 0) && (a.u_ld % 4 == 0) && ((((uintptr_t)a.U) & 15) == 0);
    for (unsigned p = p0 + ps; p < p1; p += PPW) {
        const int b = (int)fastdiv(p, a.fd_vplane);
        const unsigned rem = p - (unsigned)b * (unsigned)(a.Hv * a.Wv);
        const int y = (int)fastdiv(rem, a.fd_wv), x = (int)rem - y * a.Wv;
        float v[CBL];
        const float* vp = a.V + (long)p * a.v_ld + b_off;
        if (CBL % 4 == 0 && vvec) {
#pragma unroll
            for (int c = 0; c < CBL; c += 4) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(vp + c);
                v[c] = q[0]; v[c + 1] = q[1]; v[c + 2] = q[2]; v[c + 3] = q[3];
            }
        } else {
#pragma unroll
            for (int c = 0; c < CBL; ++c) v[c] = vp[c];
        }
#pragma unroll
        for (int c = 0; c < CBL; ++c) accb[c] += v[c];
#pragma unroll
        for (int ky = 0; ky < KH; ++ky)
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                // branch-free taps (out-of-image taps read pixel 0 and select zero): the loads of all taps of a pixel can be in
                // flight together -- this kernel is bound by the loads in flight per wave (1->16: 67 -> 43 us).  The forward
                // kernel conv_small_k measured slower this way and keeps its early-outs.
                int iy = y * S - P + ky, ix = x * S - P + kx;
                const bool inside = (unsigned)iy < (unsigned)a.Hu && (unsigned)ix < (unsigned)a.Wu;
                const float* up = a.U + (inside ? (((long)b * a.Hu + iy) * a.Wu + ix) * a.u_ld : 0L) + a_off;
                float u[CAL];
                if (CAL % 4 == 0 && uvec) {
#pragma unroll
                    for (int c = 0; c < CAL; c += 4) {
                        const f32x4 q = *reinterpret_cast<const f32x4*>(up + c);
                        u[c] = inside ? q[0] : 0.f; u[c + 1] = inside ? q[1] : 0.f; u[c + 2] = inside ? q[2] : 0.f; u[c + 3] = inside ? q[3] : 0.f;
                    }
                } else {
#pragma unroll
                    for (int ca = 0; ca < CAL; ++ca) u[ca] = inside ? up[ca] : 0.f;
                }
#pragma unroll
                for (int ca = 0; ca < CAL; ++ca)
#pragma unroll
                    for (int cb = 0; cb < CBL; ++cb)
                        acc[ky * KW + kx][ca][cb] = fmaf(u[ca], v[cb], acc[ky * KW + kx][ca][cb]);
            }
    }
    // fold the pixel slots (lanes with the same channel quad): xor over the lane bits above log2(LPP)
    auto fold = [&](float s_) {
#pragma unroll
        for (int o = 32; o >= LPP; o >>= 1) s_ += __shfl_xor(s_, o, 64);
        return s_;
    };
    float* dst = a.part + (long)pw * a.pstride;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int ca = 0; ca < CAL; ++ca)
#pragma unroll
            for (int cb = 0; cb < CBL; ++cb) {
                const float s_ = fold(acc[t][ca][cb]);
                if (ps == 0) dst[((long)t * CA + a_off + ca) * CB + b_off + cb] = s_;
            }
#pragma unroll
    for (int cb = 0; cb < CBL; ++cb) {
        const float s_ = fold(accb[cb]);
        if (ps == 0 && a.want_bias && (SPLIT_B || cq == 0)) dst[(long)TAPS * CA * CB + b_off + cb] = s_;
    }
}

// 3x3 weight gradient of the 8 -> 1 / 8 -> 2 layers with the same locality as conv_narrow_out_k: a WAVE owns a strip of 32 pixel
// columns (2 lanes per pixel, one channel quad of U each) and walks `rows` rows down it with a sliding 3-row window of U in
// registers -- every U quad is requested 3 times by one CU instead of 9 times from flat pixel runs that put a pixel's vertical
// neighbours on other XCDs (wgrad_small_k: 1.2 GB fetched per step for 0.6 GB of operands).  One partial per wave.
template <int CB>
__global__ __launch_bounds__(256) void wgrad_small_sw_k(WgradArgs a, int rows, int nstrip, int nband) {
    constexpr int CA = 8, LPP = 2, PXW = 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane % LPP, col = lane / LPP;
    const int pw = blockIdx.x * 4 + wave;
    if (pw >= a.nparts) return;
    int id = pw;
    const int strip = id % nstrip; id /= nstrip;
    const int band = id % nband;
    const int b = id / nband;
    const int x = strip * PXW + col;
    const bool colok = x < a.Wv;
    const int y0 = band * rows, y1 = min(y0 + rows, a.Hv);
    const int cx[3] = {min(max(x - 1, 0), a.Wu - 1), min(x, a.Wu - 1), min(x + 1, a.Wu - 1)};
    const bool okx[3] = {colok && x - 1 >= 0, colok, colok && x + 1 < a.Wu};
    const float* img = a.U + (long)b * a.Hu * a.Wu * a.u_ld + 4 * q;
    const float* vimg = a.V + (long)b * a.Hv * a.Wv * a.v_ld;
    float acc[9][4][CB], accb[CB];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[t][c][cb] = 0.f;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) accb[cb] = 0.f;
    auto load_raw = [&](f32x4 (&r)[3], int iy) {
        const float* rowp = img + (long)min(max(iy, 0), a.Hu - 1) * a.Wu * a.u_ld;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) r[kx] = *reinterpret_cast<const f32x4*>(rowp + (long)cx[kx] * a.u_ld);
    };
    auto mask_row = [&](f32x4 (&r)[3], int iy) {
        const bool oky = iy >= 0 && iy < a.Hu;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int c = 0; c < 4; ++c) r[kx][c] = (oky && okx[kx]) ? r[kx][c] : 0.f;
    };
    auto load_v = [&](float (&v)[CB], int y) {
        const float* vp = vimg + ((long)min(y, a.Hv - 1) * a.Wv + min(x, a.Wv - 1)) * a.v_ld;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) v[cb] = vp[cb];
    };
    f32x4 win[3][3], nxt[3];
    float v[CB], vn[CB];
    load_raw(win[0], y0 - 1); load_raw(win[1], y0); load_raw(win[2], y0 + 1);
    load_v(v, y0);
    mask_row(win[0], y0 - 1); mask_row(win[1], y0); mask_row(win[2], y0 + 1);
    for (int y = y0; y < y1; ++y) {
        load_raw(nxt, y + 2);
        load_v(vn, y + 1);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const float vv = colok ? v[cb] : 0.f;
            if (q == 0) accb[cb] += vv;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[ky * 3 + kx][c][cb] = fmaf(win[ky][kx][c], vv, acc[ky * 3 + kx][c][cb]);
        }
        mask_row(nxt, y + 2);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) { win[0][kx] = win[1][kx]; win[1][kx] = win[2][kx]; win[2][kx] = nxt[kx]; }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) v[cb] = vn[cb];
    }
    // fold the 32 pixel columns of the wave (lanes with the same channel quad): xor over the lane bits above log2(LPP)
    auto fold = [&](float s_) {
#pragma unroll
        for (int o = 32; o >= LPP; o >>= 1) s_ += __shfl_xor(s_, o, 64);
        return s_;
    };
    float* dst = a.part + (long)pw * a.pstride;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const float s_ = fold(acc[t][c][cb]);
                if (col == 0) dst[((long)t * CA + 4 * q + c) * CB + cb] = s_;
            }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        float s_ = accb[cb];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s_ += __shfl_xor(s_, o, 64);
        if (lane == 0 && a.want_bias) dst[(long)9 * CA * CB + cb] = s_;
    }
}

// 3x3 weight gradient of the one-channel-input layer (1 -> 16): 4 lanes per pixel, each owning a channel quad of V (= dY) and the
// 9 x 4 sums against the sliding 3-row window of the one-channel U; a wave owns a strip of 16 pixel columns.  Same locality
// argument as wgrad_small_sw_k; V is read with 16-byte loads, 1 KiB contiguous per wave instruction.
__global__ __launch_bounds__(256) void wgrad_cin1_k(WgradArgs a, int rows, int nstrip, int nband) {
    constexpr int CB = 16, LPP = 4, PXW = 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane % LPP, col = lane / LPP;
    const int pw = blockIdx.x * 4 + wave;
    if (pw >= a.nparts) return;
    int id = pw;
    const int strip = id % nstrip; id /= nstrip;
    const int band = id % nband;
    const int b = id / nband;
    const int x = strip * PXW + col;
    const bool colok = x < a.Wv;
    const int y0 = band * rows, y1 = min(y0 + rows, a.Hv);
    const int cx[3] = {min(max(x - 1, 0), a.Wu - 1), min(x, a.Wu - 1), min(x + 1, a.Wu - 1)};
    const bool okx[3] = {colok && x - 1 >= 0, colok, colok && x + 1 < a.Wu};
    const float* img = a.U + (long)b * a.Hu * a.Wu * a.u_ld;
    const float* vimg = a.V + (long)b * a.Hv * a.Wv * a.v_ld + 4 * q;
    f32x4 acc[9], accb = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_raw = [&](float (&r)[3], int iy) {
        const float* rowp = img + (long)min(max(iy, 0), a.Hu - 1) * a.Wu * a.u_ld;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) r[kx] = rowp[(long)cx[kx] * a.u_ld];
    };
    auto mask_row = [&](float (&r)[3], int iy) {
        const bool oky = iy >= 0 && iy < a.Hu;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) r[kx] = (oky && okx[kx]) ? r[kx] : 0.f;
    };
    auto load_v = [&](int y) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(vimg + ((long)min(y, a.Hv - 1) * a.Wv + min(x, a.Wv - 1)) * a.v_ld);
    };
    // operand rows travel PD rows ahead of their use (round 6: 1 -> 16 @640x229 41.4 -> 36.1 us; the sums and their order are untouched: bit-identical results)
    constexpr int PD = 3;
    float win[3][3], nxt[3], pre[PD][3];
    f32x4 vq[PD];
    load_raw(win[0], y0 - 1); load_raw(win[1], y0); load_raw(win[2], y0 + 1);
    f32x4 v = load_v(y0);
#pragma unroll
    for (int d = 0; d < PD; ++d) { load_raw(pre[d], y0 + 2 + d); vq[d] = load_v(y0 + 1 + d); }
    mask_row(win[0], y0 - 1); mask_row(win[1], y0); mask_row(win[2], y0 + 1);
    for (int y = y0; y < y1; ++y) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            nxt[kx] = pre[0][kx];
#pragma unroll
            for (int d = 0; d + 1 < PD; ++d) pre[d][kx] = pre[d + 1][kx];
        }
        const f32x4 vn = vq[0];
#pragma unroll
        for (int d = 0; d + 1 < PD; ++d) vq[d] = vq[d + 1];
        load_raw(pre[PD - 1], y + 2 + PD);
        vq[PD - 1] = load_v(y + 1 + PD);
        const f32x4 vv = colok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        accb += vv;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] += vv * win[ky][kx];
        mask_row(nxt, y + 2);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) { win[0][kx] = win[1][kx]; win[1][kx] = win[2][kx]; win[2][kx] = nxt[kx]; }
        v = vn;
    }
    auto fold = [&](float s_) {                       // over the 16 pixel columns of the wave (lanes with the same quad)
#pragma unroll
        for (int o = 32; o >= LPP; o >>= 1) s_ += __shfl_xor(s_, o, 64);
        return s_;
    };
    float* dst = a.part + (long)pw * a.pstride;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float s_ = fold(acc[t][c]);
            if (col == 0) dst[(long)t * CB + 4 * q + c] = s_;
        }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float s_ = fold(accb[c]);
        if (col == 0 && a.want_bias) dst[(long)9 * CB + 4 * q + c] = s_;
    }
}

// fold the per-wave partials and scatter into the PyTorch weight layout
struct WreduceArgs {
    const float* part; long pstride; int nparts;
    int taps, Ca, Cb;
    float* dw; long s_a, s_b; int flip;     // dw[a*s_a + b*s_b + tap']
    float* dbias;                            // nullable, [Cb]
    int accumulate;
};

// EL output elements x (256 / EL) partial lanes per workgroup.  The host picks EL so that the grid fills the chip:
// the small-channel layers have few outputs but hundreds of partials, and with 64 elements per workgroup their
// reduction was a handful of workgroups walking long dependent load chains (38 us for 144 outputs).
template <int EL, bool ATOMIC>
__device__ __forceinline__ void wgrad_reduce_body(const WreduceArgs& a, long block, float* sh) {
    constexpr int PL = 256 / EL;
    const long nel = (long)a.taps * a.Ca * a.Cb + (a.dbias ? a.Cb : 0);
    const int el = threadIdx.x % EL, pl = threadIdx.x / EL;
    const long e = block * EL + el;
    float s0 = 0.f, s1 = 0.f;
    if (e < nel) {
        const float* p = a.part + e;
        int k = pl;
        for (; k + PL < a.nparts; k += 2 * PL) {
            s0 += p[(long)k * a.pstride];
            s1 += p[(long)(k + PL) * a.pstride];
        }
        for (; k < a.nparts; k += PL) s0 += p[(long)k * a.pstride];
    }
    sh[pl * EL + el] = s0 + s1;
    __syncthreads();
#pragma unroll
    for (int h = PL / 2; h >= 1; h >>= 1) {               // fixed tree: deterministic
        if (pl < h) sh[pl * EL + el] += sh[(pl + h) * EL + el];
        __syncthreads();
    }
    if (pl != 0 || e >= nel) return;
    const float s = sh[el];
    const long nw = (long)a.taps * a.Ca * a.Cb;
    if (e < nw) {
        int bb = (int)(e % a.Cb);
        long t = e / a.Cb;
        int aa = (int)(t % a.Ca);
        int tap = (int)(t / a.Ca);
        int tt = a.flip ? (a.taps - 1 - tap) : tap;
        float* d = a.dw + (long)aa * a.s_a + (long)bb * a.s_b + tt;
        if (ATOMIC) atomicAdd(d, s);
        else *d = a.accumulate ? *d + s : s;
    } else {
        float* d = a.dbias + (e - nw);
        if (ATOMIC) atomicAdd(d, s);
        else *d = a.accumulate ? *d + s : s;
    }
}

// The same with 16-byte loads: a thread owns FOUR consecutive elements (EL / 4 threads cover the workgroup's elements, 1024 / EL
// partial lanes), so a partial-sum vector is fetched with a quarter of the load instructions and every thread has all its loads in
// flight at once.  Needs nel, pstride % 4 == 0 and a 16-byte aligned partial buffer (the host checks: WreduceEntry::vec4).
template <int EL, bool ATOMIC>
__device__ __forceinline__ void wgrad_reduce_body_v4(const WreduceArgs& a, long block, float* sh) {
    constexpr int VL = EL / 4, PL = 256 / VL;
    const long nel = (long)a.taps * a.Ca * a.Cb + (a.dbias ? a.Cb : 0);
    const int vl = threadIdx.x % VL, pl = threadIdx.x / VL;
    const long e = block * EL + vl * 4;
    f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (e < nel) {
        const float* p = a.part + e;
        int k = pl;
#pragma unroll 2
        for (; k + 3 * PL < a.nparts; k += 4 * PL) {
            s0 += *reinterpret_cast<const f32x4*>(p + (long)k * a.pstride);
            s1 += *reinterpret_cast<const f32x4*>(p + (long)(k + PL) * a.pstride);
            s2 += *reinterpret_cast<const f32x4*>(p + (long)(k + 2 * PL) * a.pstride);
            s3 += *reinterpret_cast<const f32x4*>(p + (long)(k + 3 * PL) * a.pstride);
        }
        for (; k < a.nparts; k += PL) s0 += *reinterpret_cast<const f32x4*>(p + (long)k * a.pstride);
    }
    f32x4* shv = reinterpret_cast<f32x4*>(sh);
    shv[pl * VL + vl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
#pragma unroll
    for (int h = PL / 2; h >= 1; h >>= 1) {               // fixed tree: deterministic
        if (pl < h) shv[pl * VL + vl] += shv[(pl + h) * VL + vl];
        __syncthreads();
    }
    if (pl != 0 || e >= nel) return;
    const f32x4 s = shv[vl];
    const long nw = (long)a.taps * a.Ca * a.Cb;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long eq = e + q;
        float* d;
        if (eq < nw) {
            const int bb = (int)(eq % a.Cb);
            const long t = eq / a.Cb;
            const int aa = (int)(t % a.Ca);
            const int tap = (int)(t / a.Ca);
            const int tt = a.flip ? (a.taps - 1 - tap) : tap;
            d = a.dw + (long)aa * a.s_a + (long)bb * a.s_b + tt;
        } else {
            d = a.dbias + (eq - nw);
        }
        if (ATOMIC) atomicAdd(d, s[q]);
        else *d = a.accumulate ? *d + s[q] : s[q];
    }
}

template <int EL, bool V4 = false>
__global__ __launch_bounds__(256) void wgrad_reduce_k(WreduceArgs a) {
    __shared__ __attribute__((aligned(16))) float sh[V4 ? 1024 : 256];
    if (V4) wgrad_reduce_body_v4<EL, false>(a, blockIdx.x, sh);
    else wgrad_reduce_body<EL, false>(a, blockIdx.x, sh);
}

// All deferred reductions of a backward pass in ONE launch (rv_wgrad_reduce_table): a workgroup finds its entry by
// bisection over the block prefix.  Several entries may target the same gradient (the same layer in several passes of the
// step), so the final add is an fp32 atomic.
struct WreduceEntry {
    WreduceArgs a;
    long block0;      // nblocks until rv_wgrad_table_finalize turns it into the exclusive prefix
    int el;           // elements per workgroup: 4, 16 or 64
    int vec4;         // el >= 16: 16-byte loads (wgrad_reduce_body_v4)
};
__global__ __launch_bounds__(256) void wgrad_reduce_table_k(const WreduceEntry* tab, int count, int plain) {
    __shared__ WreduceEntry ent;
    __shared__ __attribute__((aligned(16))) float sh[1024];
    if (threadIdx.x == 0) {
        int lo = 0, hi = count - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tab[mid].block0 <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        ent = tab[lo];
    }
    __syncthreads();
    const long block = (long)blockIdx.x - ent.block0;
#ifdef RV_ABLATION
    if (plain) {                                  // (timing ablation, wrong results: the final add as a plain store)
        ent.a.accumulate = 0;
        if (ent.vec4 && ent.el == 64) wgrad_reduce_body_v4<64, false>(ent.a, block, sh);
        else if (ent.vec4 && ent.el == 16) wgrad_reduce_body_v4<16, false>(ent.a, block, sh);
        else if (ent.el == 64) wgrad_reduce_body<64, false>(ent.a, block, sh);
        else if (ent.el == 16) wgrad_reduce_body<16, false>(ent.a, block, sh);
        else wgrad_reduce_body<4, false>(ent.a, block, sh);
        return;
    }
#endif
    if (ent.vec4 && ent.el == 1024) wgrad_reduce_body_v4<1024, true>(ent.a, block, sh);
    else if (ent.vec4 && ent.el == 64) wgrad_reduce_body_v4<64, true>(ent.a, block, sh);
    else if (ent.vec4 && ent.el == 16) wgrad_reduce_body_v4<16, true>(ent.a, block, sh);
    else if (ent.el == 64) wgrad_reduce_body<64, true>(ent.a, block, sh);
    else if (ent.el == 16) wgrad_reduce_body<16, true>(ent.a, block, sh);
    else wgrad_reduce_body<4, true>(ent.a, block, sh);
}

// ------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------
template <int KH, int KW, int S, int P, int R, bool SC>
static int launch_conv_rt(const ConvArgs& a, int NT, int MT, hipStream_t st, bool ksplit = false) {
    const long ntiles = (a.npix + 15) / 16;
    if (ksplit) {                              // K loop split over the four waves of a workgroup (deep layers; R == 4 only)
        if constexpr (R == 4 && !(KH == 3)) {
#define RV_CASEK(nt, mt)                                                                         \
    if (NT == nt && MT == mt) {                                                                  \
        dim3 grid(cdiv(ntiles, mt), a.ntile_n / nt);                                             \
        hipLaunchKernelGGL((conv_mfma_k<KH, KW, S, P, R, nt, mt, SC, 4>), grid, dim3(256), 0, st, a); \
        return RV_OK;                                                                            \
    }
            RV_CASEK(1, 1) RV_CASEK(1, 2) RV_CASEK(2, 1) RV_CASEK(2, 2) RV_CASEK(1, 4) RV_CASEK(4, 1)
#undef RV_CASEK
        }
        return RV_EUNSUPPORTED;
    }
#define RV_CASE(nt, mt)                                                                          \
    if (NT == nt && MT == mt) {                                                                  \
        dim3 grid(cdiv(ntiles, 4 * mt), a.ntile_n / nt);                                         \
        hipLaunchKernelGGL((conv_mfma_k<KH, KW, S, P, R, nt, mt, SC>), grid, dim3(256), 0, st, a); \
        return RV_OK;                                                                            \
    }
    RV_CASE(1, 1) RV_CASE(1, 2) RV_CASE(1, 4)
    RV_CASE(2, 1) RV_CASE(2, 2) RV_CASE(2, 4)
    RV_CASE(3, 1) RV_CASE(3, 2) RV_CASE(3, 4)
    RV_CASE(4, 1) RV_CASE(4, 2) RV_CASE(4, 4)
#undef RV_CASE
    return RV_EUNSUPPORTED;
}

// choose (NT, MT): as much register-level reuse as possible while keeping >= ~2048 waves in flight
static void choose_tiles(long ntiles, int ntile_n, int* NT, int* MT) {
    static const int nts[4] = {4, 3, 2, 1};
    static const int mts[3] = {4, 2, 1};
    int best_nt = 1, best_mt = 1;
    long best_score = -1;
    for (int a = 0; a < 4; ++a) {
        int nt = nts[a];
        if (ntile_n % nt) continue;
        for (int b = 0; b < 3; ++b) {
            int mt = mts[b];
            long waves = ((ntiles + mt - 1) / mt) * (ntile_n / nt);
            // reuse score = MFMAs per fragment load; penalise launches that cannot fill the chip
            long reuse = (long)(nt * mt * 100) / (nt + mt);
            long fill = waves >= 2048 ? 100 : (waves * 100) / 2048;
            long score = reuse * fill;
            if (score > best_score) { best_score = score; best_nt = nt; best_mt = mt; }
        }
    }
    *NT = best_nt; *MT = best_mt;
}

// LDS-pipelined 3x3 launch: pick (NT, MTW, TH) so that two units fit LDS and the persistent grid covers the chip
static size_t conv3x3_lds_bytes(int R, int NT, int TH, int W, int nchunk, int nbuf = 2) {
    return ((size_t)nbuf * (TH + 2) * (W + 2) * 4 * R + (size_t)(nchunk > 1 ? nbuf : 1) * 9 * NT * 64 * R) * sizeof(float);
}

template <int R, int NW, bool BF = false>
static int launch_conv3x3_lds_r(const ConvArgs& a, int NT, int MTW, int TH, int wgs_per_cu, hipStream_t st) {
    ConvLdsArgs aa;
    aa.c = a; aa.TH = TH; aa.nbands = cdiv(a.H, TH);
    aa.total_bands = a.B * aa.nbands;
    const int nsplit = a.ntile_n / NT;
    int wgs = (256 * wgs_per_cu) / nsplit;
    if (wgs < 1) wgs = 1;
    if (wgs > aa.total_bands) wgs = aa.total_bands;
    aa.bands_per_wg = cdiv(aa.total_bands, wgs);
    wgs = cdiv(aa.total_bands, aa.bands_per_wg);
    // RV_CONV_NBUF=3: a third unit buffer (prefetch distance 2) when it fits.  Measured neutral on MI355X (the
    // kernel is not DMA-latency bound), so two buffers -- less LDS -- stay the default.
    static const int nbuf_env = getenv("RV_CONV_NBUF") ? atoi(getenv("RV_CONV_NBUF")) : 0;
    aa.nbuf = 2;
    if (nbuf_env == 3 && conv3x3_lds_bytes(R, NT, TH, a.W, a.nchunk, 3) <= (size_t)(wgs_per_cu > 1 ? 76 : 150) * 1024) aa.nbuf = 3;
    aa.ablate = getenv("RV_ABLATE") ? atoi(getenv("RV_ABLATE")) : 0;
    static const int skew_env = getenv("RV_CONV_SKEW") ? atoi(getenv("RV_CONV_SKEW")) : 1;
    aa.skew = (aa.nbuf == 3 && NW >= 8) ? skew_env : 0;
    const size_t lds = conv3x3_lds_bytes(R, NT, TH, a.W, a.nchunk, aa.nbuf);
    static const int xcd_env = getenv("RV_CONV_XCD") ? atoi(getenv("RV_CONV_XCD")) : 1;
    aa.nsplit = nsplit; aa.xcd = xcd_env;
    dim3 grid(wgs * nsplit), blk(NW * 64);
#define RV_L3(nt, mt)                                                                              \
    if (NT == nt && MTW == mt) {                                                                  \
        auto kern = conv3x3_lds_k<R, nt, mt, NW, BF>;                                             \
        /* (launches come from the autograd thread as well as from the main thread: once_flag, not a plain static bool) */ \
        static std::once_flag attr_once;                                                          \
        std::call_once(attr_once, [&] {                                                           \
            /* the kernel also has 1 KiB of static LDS: dynamic + static must stay within the 160 KiB of a CU */ \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) \
                (void)hipGetLastError();                                                          \
        });                                                                                       \
        hipLaunchKernelGGL(kern, grid, blk, lds, st, aa);                                         \
        return RV_OK;                                                                             \
    }
    if constexpr (NW == 12) {
        // 12 waves = three per SIMD with 3 / 5 / 6 tiles each: 9 / 15 / 18 tiles per SIMD and band, the band sizes at which whole
        // rows of the 57- / 114- / 229-pixel-wide layers fill 99 % of the tile slots (a power-of-two tile count leaves 89 %)
        RV_L3(1, 3) RV_L3(2, 3) RV_L3(3, 3) RV_L3(4, 3)
        RV_L3(1, 5) RV_L3(2, 5) RV_L3(1, 6) RV_L3(2, 6)
        RV_L3(1, 1) RV_L3(2, 1) RV_L3(4, 1) RV_L3(1, 2) RV_L3(2, 2) RV_L3(4, 2) RV_L3(1, 4) RV_L3(2, 4)
    } else {
        RV_L3(1, 1) RV_L3(1, 2) RV_L3(1, 4) RV_L3(1, 8)
        RV_L3(2, 1) RV_L3(2, 2) RV_L3(2, 4) RV_L3(2, 8)
        RV_L3(3, 1) RV_L3(3, 2) RV_L3(3, 4)
        RV_L3(4, 1) RV_L3(4, 2) RV_L3(4, 4)
    }
#undef RV_L3
    return RV_EUNSUPPORTED;
}

static int launch_conv3x3_lds(const ConvArgs& a, int R, hipStream_t st, int force_nt = 0, int force_mt = 0, int nw = 4, int force_th = 0,
                              bool bf = false) {
    static const int mts_pow2[7] = {8, 4, 2, 1, 0, 0, 0}, mts_12[7] = {6, 5, 4, 3, 2, 1, 0};
    const int* mts = nw == 12 ? mts_12 : mts_pow2;
    // search (NT, MTW): prefer large tiles, but need >= ~1.5 units per workgroup slot on the chip
    int best_nt = 0, best_mt = 0, best_th = 0, best_wpc = 1;
    long best_score = -1;
    for (int nt = 4; nt >= 1; --nt) {
        if (a.ntile_n % nt) continue;
        if (force_nt && nt != force_nt) continue;
        for (int k = 0; k < 7; ++k) {
            const int mt = mts[k];
            if (!mt || (force_mt && mt != force_mt)) continue;
            if (nt * mt > 16 || (nw >= 12 && nt * mt > (nw == 16 ? 8 : 12)) || (nw == 12 && ((mt > 3 && nt > 2) || (nt == 3 && mt != 3)))) continue;      // accumulator budget (<= 128 registers per wave at 16 waves)
            int th = (mt * 16 * nw) / a.W;                                // nw waves x mt tiles x 16 pixels per band
            if (th > a.H) th = a.H;
            if (force_th) {                                               // fewer rows than the tile slots hold: band-count quantisation
                if (force_th > th) continue;
                th = force_th;
            }
            if (th < 1) continue;
            const size_t lds = conv3x3_lds_bytes(R, nt, th, a.W, a.nchunk);
            if (lds > 150 * 1024) continue;
            const int wpc = (lds <= 76 * 1024 && nw == 4) ? 2 : 1;         // workgroups per CU
            const long bands = (long)a.B * cdiv(a.H, th) * (a.ntile_n / nt);
            const long slots = 256L * wpc;
            // efficiency model: work per slot in whole bands (tail effect) x register-level reuse
            const long rounds = (bands + slots - 1) / slots;
            const long fill = (bands * 100) / (rounds * slots);            // 0..100
            const long reuse = (long)(nt * mt * 100) / (nt + mt);
            const long score = fill * (50 + reuse);
            if (score > best_score) { best_score = score; best_nt = nt; best_mt = mt; best_th = th; best_wpc = wpc; }
        }
    }
    if (!best_nt) return RV_EUNSUPPORTED;
    if (bf) {                                  // bf16 operands (16-channel chunks only; the caller checked R == 4)
        if (nw == 12) return launch_conv3x3_lds_r<4, 12, true>(a, best_nt, best_mt, best_th, best_wpc, st);
        if (nw == 16) return launch_conv3x3_lds_r<4, 16, true>(a, best_nt, best_mt, best_th, best_wpc, st);
        if (nw == 8) return launch_conv3x3_lds_r<4, 8, true>(a, best_nt, best_mt, best_th, best_wpc, st);
        return launch_conv3x3_lds_r<4, 4, true>(a, best_nt, best_mt, best_th, best_wpc, st);
    }
    if (nw == 12)
        return R == 4 ? launch_conv3x3_lds_r<4, 12>(a, best_nt, best_mt, best_th, best_wpc, st)
                      : launch_conv3x3_lds_r<2, 12>(a, best_nt, best_mt, best_th, best_wpc, st);
    if (nw == 16)
        return R == 4 ? launch_conv3x3_lds_r<4, 16>(a, best_nt, best_mt, best_th, best_wpc, st)
                      : launch_conv3x3_lds_r<2, 16>(a, best_nt, best_mt, best_th, best_wpc, st);
    if (nw == 8)
        return R == 4 ? launch_conv3x3_lds_r<4, 8>(a, best_nt, best_mt, best_th, best_wpc, st)
                      : launch_conv3x3_lds_r<2, 8>(a, best_nt, best_mt, best_th, best_wpc, st);
    return R == 4 ? launch_conv3x3_lds_r<4, 4>(a, best_nt, best_mt, best_th, best_wpc, st)
                  : launch_conv3x3_lds_r<2, 4>(a, best_nt, best_mt, best_th, best_wpc, st);
}

// Winograd launch (families 0x6NM: 8 waves, 0xANM / 0xCNM: 8 / 12 waves with the half-chunk patch): a band of TH (even) rows holds (TH/2) x ceil(W/2) tiles of 2x2 outputs,
// NW x MTW groups of 16 tiles per unit.  force_th = 0: as many rows as the tile slots hold.
static size_t conv3x3_wino_bytes(int NT, int TH, int W, int wbufs) {
    const int NP = (2 * ((W + 1) / 2) + 2 + 15) / 16;            // 1 KiB pieces (16 pixels x 16 channels) per staged row
    return (size_t)2 * (TH + 2) * NP * 1024 + (size_t)wbufs * 16 * NT * 1024;
}

// cap / wg_slots: LDS budget of a workgroup and workgroup slots on the chip -- 154 KiB and 256 (one workgroup per CU) for the shipped families;
// 78 KiB and 512 for the half-CU experiment family 0xE (round 6: two four-wave workgroups per CU, possibly of two different kernels)
template <int NW, bool HALF = false, int LAY = 1>
static int launch_conv3x3_wino(const ConvArgs& a0, int NT, int MTW, int force_th, hipStream_t st, size_t cap = 154 * 1024, int wg_slots = 256) {
    if (NT < 1 || a0.ntile_n % NT) return RV_EUNSUPPORTED;
    if (NW == 12 && a0.bn_z) return RV_EUNSUPPORTED;      // the fused BatchNorm-backward epilogue does not fit three waves per SIMD without scratch
    ConvLdsArgs aa;
    aa.c = a0;
    const long in_bytes = (((long)a0.B * a0.H * a0.W - 1) * a0.in_ld + a0.Cin) * 4;
    if (in_bytes >= 0x3f000000L) return RV_EUNSUPPORTED;      // the staging loads address the input view with 30-bit offsets (see the kernel)
    aa.in_bytes = (unsigned)in_bytes;
    const int WT = (a0.W + 1) / 2;
    aa.c.fd_pw = fastdiv_make((unsigned)WT);
    int trows = (NW * MTW * 16) / WT;
    if (trows < 1) return RV_EUNSUPPORTED;
    int TH = 2 * trows;
    if (TH > a0.H) TH = (a0.H + 1) & ~1;
    if (force_th) {
        if (force_th > TH || (force_th & 1)) return RV_EUNSUPPORTED;
        TH = force_th;
    }
    // weights: resident for the whole kernel when all chunks fit next to the two band buffers (no weight DMA after unit 0, 16 NT KiB less
    // L2 traffic per unit); else double-buffered per chunk like the band
    static const int wres_env = getenv("RV_WINO_WRES") ? atoi(getenv("RV_WINO_WRES")) : 1;
    aa.wres = 0;
    size_t lds = conv3x3_wino_bytes(NT, TH, a0.W, a0.nchunk > 1 ? 2 : 1);
    // "as many rows as the tile slots hold" (force_th == 0) also means: as many as the LDS holds (a staged row is a whole number of
    // 1 KiB pieces, so e.g. a 114-pixel row takes 8 KiB)
    while (!force_th && lds > cap && TH > 2) {
        TH -= 2;
        lds = conv3x3_wino_bytes(NT, TH, a0.W, a0.nchunk > 1 ? 2 : 1);
    }
    if (a0.nchunk > 1 && wres_env) {
        const size_t lds_res = conv3x3_wino_bytes(NT, TH, a0.W, a0.nchunk);
        if (lds_res <= cap) { aa.wres = 1; lds = lds_res; }
    }
    if (lds > cap) return RV_EUNSUPPORTED;
    aa.TH = TH; aa.nbands = cdiv(a0.H, TH);
    aa.total_bands = a0.B * aa.nbands;
    const int nsplit = a0.ntile_n / NT;
    int wgs = wg_slots / nsplit;
    if (wgs < 1) wgs = 1;
    if (wgs > aa.total_bands) wgs = aa.total_bands;
    aa.bands_per_wg = cdiv(aa.total_bands, wgs);
    wgs = cdiv(aa.total_bands, aa.bands_per_wg);
    aa.nbuf = 2; aa.skew = 0;
    aa.ablate = getenv("RV_ABLATE") ? atoi(getenv("RV_ABLATE")) : 0;
    static const int xcd_env = getenv("RV_CONV_XCD") ? atoi(getenv("RV_CONV_XCD")) : 1;
    aa.nsplit = nsplit; aa.xcd = xcd_env;
    dim3 grid(wgs * nsplit), blk(NW * 64);
#define RV_WN(nt, mt)                                                                              \
    if (NT == nt && MTW == mt) {                                                                  \
        static std::once_flag attr_once;                                                          \
        std::call_once(attr_once, [] {                                                            \
            if (hipFuncSetAttribute((const void*)conv3x3_wino_k<nt, mt, NW, HALF, LAY, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) \
                (void)hipGetLastError();                                                          \
            if (hipFuncSetAttribute((const void*)conv3x3_wino_k<nt, mt, NW, HALF, LAY, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) \
                (void)hipGetLastError();                                                          \
        });                                                                                       \
        if (a0.bn_z) hipLaunchKernelGGL((conv3x3_wino_k<nt, mt, NW, HALF, LAY, true>), grid, blk, lds, st, aa);  \
        else hipLaunchKernelGGL((conv3x3_wino_k<nt, mt, NW, HALF, LAY, false>), grid, blk, lds, st, aa);         \
        return RV_OK;                                                                             \
    }
    // (instantiated: the tiles that fit the register file without scratch -- 64 accumulator registers per (tile group, n-tile) pair)
    if constexpr (HALF && NW == 8) {
        RV_WN(2, 1) RV_WN(1, 1)
    } else {
        RV_WN(1, 1)
    }
#undef RV_WN
    return RV_EUNSUPPORTED;
}

static int frag_R(int kdim) { return (kdim % 16 == 0) ? 4 : ((kdim % 8 == 0) ? 2 : 0); }
// 3x3 weights with 16-channel chunks carry their Winograd transform behind the nine tap fragments (conv3x3_wino_k)
static bool wino_packed(int taps, int kdim) { return taps == 9 && kdim % 16 == 0; }

// ---- profiling aid: raw f32 MFMA issue rate (NACC independent accumulators per wave) -------------------------
template <int NACC>
__global__ __launch_bounds__(256) void mfma_peak_k(float* out, int iters, float seed) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a0 = seed + threadIdx.x, b0 = seed * 0.5f + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0 + r, b0 + i, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

extern "C" {

// profiling aid: nacc in {2,4,8}; returns 0.  FLOPs = blocks*4 waves*iters*4*nacc*2048
int rv_debug_mfma_peak(float* out, int blocks, int iters, int nacc, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (nacc == 2) hipLaunchKernelGGL(mfma_peak_k<2>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f);
    else if (nacc == 4) hipLaunchKernelGGL(mfma_peak_k<4>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f);
    else hipLaunchKernelGGL(mfma_peak_k<8>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f);
    RV_LAUNCH_CHECK("rv_debug_mfma_peak");
    return RV_OK;
}

// Size (floats) of the packed fragment buffer for a logical Wm[taps][kdim][ndim].
long rv_packed_weight_floats(int taps, int kdim, int ndim) {
    int R = frag_R(kdim);
    if (R == 0) return (long)taps * kdim * ndim;   // plain layout (small-channel kernels)
    int nchunk = kdim / (4 * R), ntile_n = (ndim + 15) / 16;
    return (long)(taps + (wino_packed(taps, kdim) ? 16 : 0)) * nchunk * ntile_n * 64 * R;
}

// Pack a PyTorch-layout weight into MFMA fragment order (or plain [tap][k][n] when the layer runs on
// the small-channel VALU kernels: kdim not a multiple of 8, or ndim <= 2).
//   value(tap,k,n) = w[k*s_k + n*s_n + (flip ? taps-1-tap : tap)]
//   scatter_cmid>0 (2x2/s2 scatter GEMM): n = tap4*cmid + c, value = w[k*s_k + c*s_n + tap4], taps must be 1.
static int pack_args_make(PackArgs& a, const float* w, float* out, int taps, int kdim, int ndim, long s_k, long s_n,
                          int flip, int scatter_cmid, int force_plain) {
    a.w = w; a.out = out; a.taps = taps; a.kdim = kdim; a.ndim = ndim;
    a.s_k = s_k; a.s_n = s_n; a.flip = flip; a.scatter_cmid = scatter_cmid; a.wino0 = 0;
    const int R = force_plain ? 0 : frag_R(kdim);
    if (R == 0) {
        RV_CHECK_ARG(scatter_cmid == 0, "rv_pack_weights: plain layout has no scatter form");
        a.R = 0; a.ntile_n = 0; a.nchunk = 0;
        a.total = (long)taps * kdim * ndim;
    } else {
        a.R = R; a.nchunk = kdim / (4 * R); a.ntile_n = (ndim + 15) / 16;
        a.total = (long)taps * a.nchunk * a.ntile_n * 64 * R;
        if (wino_packed(taps, kdim) && scatter_cmid == 0) {
            a.wino0 = a.total;
            a.total += (long)16 * a.nchunk * a.ntile_n * 64 * R;
        }
    }
    return RV_OK;
}

int rv_pack_weights(const float* w, float* out, int taps, int kdim, int ndim, long s_k, long s_n, int flip,
                    int scatter_cmid, int force_plain, void* stream) {
    PackArgs a;
    const int rc = pack_args_make(a, w, out, taps, kdim, ndim, s_k, s_n, flip, scatter_cmid, force_plain);
    if (rc != RV_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (a.R == 0) hipLaunchKernelGGL(pack_plain_k, dim3(cdiv(a.total, 256)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(pack_frag_k, dim3(cdiv(a.total, 256)), dim3(256), 0, st, a);
    RV_LAUNCH_CHECK("rv_pack_weights");
    return RV_OK;
}

// Batched form: the caller builds a table of rv_pack_table_entry_bytes()-sized entries IN HOST MEMORY with
// rv_pack_table_fill (entry i of `table`, same arguments as rv_pack_weights; entries must be filled in order
// 0, 1, 2, ... because each records the running workgroup offset), copies it to the device once, and then repacks
// every weight with one rv_pack_table_run launch per optimiser step.  rv_pack_table_fill returns the total number
// of workgroups up to and including entry i (pass the last value as total_blocks).
long rv_pack_table_entry_bytes(void) { return (long)sizeof(PackEntry); }

long rv_pack_table_fill(void* table, int i, const float* w, float* out, int taps, int kdim, int ndim, long s_k, long s_n,
                        int flip, int scatter_cmid, int force_plain) {
    PackEntry* tab = (PackEntry*)table;
    if (pack_args_make(tab[i].a, w, out, taps, kdim, ndim, s_k, s_n, flip, scatter_cmid, force_plain) != RV_OK) return -1;
    tab[i].block0 = i == 0 ? 0 : tab[i - 1].block0 + cdiv(tab[i - 1].a.total, 256);
    return tab[i].block0 + cdiv(tab[i].a.total, 256);
}

int rv_pack_table_run(const void* table_dev, int count, long total_blocks, void* stream) {
    RV_CHECK_ARG(count > 0 && total_blocks > 0 && total_blocks < (1L << 31), "rv_pack_table_run: empty or oversized table");
    hipLaunchKernelGGL(pack_table_k, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const PackEntry*)table_dev, count);
    RV_LAUNCH_CHECK("rv_pack_table_run");
    return RV_OK;
}

// mode: 0 = 3x3 s1 p1, 1 = 1x1, 2 = 2x2 s2 gather (down fwd / up dgrad), 3 = 2x2 s2 scatter (up fwd / down dgrad)
// algo: 0 = default, 1 = direct (LDS-free) kernel, 2 = LDS/DMA-pipelined kernel (mode 0 only)
// in  : [B,H,W,*] pixel stride in_ld, Cin channels read from the pointer
// out : [B,Ho,Wo,*] pixel stride out_ld, Cout channels written
// bn_sums != NULL: on return (stream order) bn_sums[0..Cout) += per-channel sum and bn_sums[Cout..2Cout) += sum of
// squares of the values written to `out` -- the BatchNorm2d batch statistics of the consumer (rv_bn_lrelu_fwd with
// sums_ready = 1).  The persistent 3x3 kernel produces them in its epilogue; every other kernel is followed by the
// BatchNorm statistics pass.
// bn_z != NULL (with bn_z_ld, bn_coef = the [5C] coefficients rv_bn_lrelu_fwd saved, bn_slope): `out` is the gradient
// wrt the OUTPUT of a BatchNorm+leaky-ReLU whose pre-normalisation input is bn_z, and bn_sums receives that layer's
// backward reduction instead (sum of dd and of dd * xhat, dd = out * lrelu'): rv_bn_lrelu_bwd(..., sums_ready = 1)
// then skips its reduction pass.
struct BnFuse { double* sums; const float* z; int z_ld; const float* coef; float slope; };
static int conv_fwd_impl(int mode, const float* in, int in_ld, int B, int H, int W, int Cin, float* out, int out_ld, int Ho,
                         int Wo, int Cout, const float* wpack, const float* bias, int accumulate, int algo, const BnFuse& bn,
                         bool* sums_done, hipStream_t st);

int rv_conv_fwd(int mode, const float* in, int in_ld, int B, int H, int W, int Cin, float* out, int out_ld, int Ho,
                int Wo, int Cout, const float* wpack, const float* bias, int accumulate, int algo, double* bn_sums,
                const float* bn_z, int bn_z_ld, const float* bn_coef, float bn_slope, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    bool sums_done = false;
    RV_CHECK_ARG(!bn_z || (bn_sums && bn_coef), "rv_conv_fwd: bn_z needs bn_sums and bn_coef");
    const BnFuse bn = {bn_sums, bn_z, bn_z_ld, bn_coef, bn_slope};
    const int rc = conv_fwd_impl(mode, in, in_ld, B, H, W, Cin, out, out_ld, Ho, Wo, Cout, wpack, bias, accumulate, algo,
                                 bn, &sums_done, st);
    if (rc != RV_OK || !bn_sums || sums_done) return rc;
    if (bn_z) return rv_internal_bn_bwd_stats(out, out_ld, bn_z, bn_z_ld, (long)B * Ho * Wo, Cout, bn_coef, bn_slope, bn_sums, st);
    return rv_internal_bn_stats(out, out_ld, (long)B * Ho * Wo, Cout, bn_sums, st);
}

static int conv_fwd_impl(int mode, const float* in, int in_ld, int B, int H, int W, int Cin, float* out, int out_ld, int Ho,
                         int Wo, int Cout, const float* wpack, const float* bias, int accumulate, int algo, const BnFuse& bn,
                         bool* sums_done, hipStream_t st) {
    double* bn_sums = bn.sums;
    RV_CHECK_ARG(mode >= 0 && mode <= 3, "rv_conv_fwd: bad mode %d", mode);
    if (mode == 0 || mode == 1) RV_CHECK_ARG(Ho == H && Wo == W, "rv_conv_fwd: same-size conv needs Ho==H, Wo==W");
    if (mode == 2) RV_CHECK_ARG(Ho == H / 2 && Wo == W / 2, "rv_conv_fwd: down conv needs Ho=H/2, Wo=W/2");
    if (mode == 3)
        RV_CHECK_ARG(Ho >= 2 * H && Ho <= 2 * H + 1 && Wo >= 2 * W && Wo <= 2 * W + 1,
                     "rv_conv_fwd: up conv needs 2H<=Ho<=2H+1 (got H=%d Ho=%d W=%d Wo=%d)", H, Ho, W, Wo);
    const int R = frag_R(Cin);
    const bool small = (R == 0) || (Cout <= 2 && mode != 3);
    if (small) {
        SmallArgs s;
        s.in = in; s.in_ld = in_ld; s.H = H; s.W = W; s.out = out; s.out_ld = out_ld; s.Ho = Ho; s.Wo = Wo;
        s.B = B; s.wplain = wpack; s.bias = bias; s.npix = (long)B * Ho * Wo; s.accumulate = accumulate;
        s.fd_plane = fastdiv_make((unsigned)(Ho * Wo)); s.fd_wo = fastdiv_make((unsigned)Wo);
        const int tpp = Cout / small_cpt(Cin, Cout, mode == 0 ? 9 : 1);
        RV_CHECK_ARG(s.npix * tpp < (1L << 31), "rv_conv_fwd: more than 2^31 outputs");
        s.bn_sums = (Cout >= 4 && !bn.z) ? bn_sums : nullptr;
        if (s.bn_sums) *sums_done = true;
        long nblk = cdiv(s.npix * tpp, 256);
        const long cap = s.bn_sums ? 1024 : 4096;   // grid stride (a multiple of every COUT/4); fewer workgroups = fewer atomics
        if (nblk > cap) nblk = cap;
        dim3 grid((unsigned)nblk), blk(256);
#define RV_SMALL(ci, co, kh, kw, ss, pp)                                                          \
    if (Cin == ci && Cout == co) {                                                               \
        hipLaunchKernelGGL((conv_small_k<ci, co, kh, kw, ss, pp>), grid, blk, 0, st, s);         \
        RV_LAUNCH_CHECK("conv_small");                                                           \
        return RV_OK;                                                                            \
    }
#define RV_NARROW(ci, co)                                                                         \
    if (mode == 0 && Cin == ci && Cout == co && (in_ld & 3) == 0 && ((((uintptr_t)in) & 15) == 0)) { \
        RV_CHECK_ARG(s.npix < (1L << 28), "rv_conv_fwd: more than 2^28 pixels");                \
        static const int nrows_env = getenv("RV_NARROW_ROWS") ? atoi(getenv("RV_NARROW_ROWS")) : 16;   \
        s.rows = nrows_env;                                                                      \
        const long nb = (long)B * cdiv(H, s.rows) * cdiv(W, 256 / (ci / 4));                     \
        hipLaunchKernelGGL((conv_narrow_out_k<ci, co>), dim3((unsigned)nb), blk, 0, st, s);      \
        RV_LAUNCH_CHECK("conv_narrow_out");                                                      \
        return RV_OK;                                                                            \
    }
        RV_NARROW(16, 1) RV_NARROW(8, 2) RV_NARROW(8, 1)
#undef RV_NARROW
        static const int cin1_env = getenv("RV_CONV_CIN1") ? atoi(getenv("RV_CONV_CIN1")) : 1;
        const bool cin12 = mode == 0 && ((Cin == 1 && (Cout == 16 || Cout == 8)) || (Cin == 2 && Cout == 8 && (in_ld & 1) == 0 && ((((uintptr_t)in) & 7) == 0)));
        const bool bz_ok = !bn.z || ((bn.z_ld & 3) == 0 && ((((uintptr_t)bn.z) & 15) == 0) && ((((uintptr_t)bn.coef) & 15) == 0));
        if (cin1_env && cin12 && bz_ok && (!bias || ((((uintptr_t)bias) & 15) == 0))) {
            s.rows = 16;
            s.bn_z = bn.z; s.bn_z_ld = bn.z_ld; s.bn_coef = bn.coef; s.bn_slope = bn.slope;
            if (bn.z) { s.bn_sums = bn_sums; *sums_done = true; }      // the backward reduction rides in this kernel's epilogue
            const long nb = (long)B * cdiv(H, s.rows) * cdiv(W, 1024 / Cout);
            if (Cin == 1 && Cout == 16) hipLaunchKernelGGL((conv_cin12_k<1, 16>), dim3((unsigned)nb), blk, 0, st, s);
            else if (Cin == 1) hipLaunchKernelGGL((conv_cin12_k<1, 8>), dim3((unsigned)nb), blk, 0, st, s);
            else hipLaunchKernelGGL((conv_cin12_k<2, 8>), dim3((unsigned)nb), blk, 0, st, s);
            RV_LAUNCH_CHECK("conv_cin12");
            return RV_OK;
        }
        if (mode == 0) {
            RV_SMALL(1, 16, 3, 3, 1, 1) RV_SMALL(16, 1, 3, 3, 1, 1)
            RV_SMALL(8, 2, 3, 3, 1, 1)  RV_SMALL(2, 8, 3, 3, 1, 1)
            RV_SMALL(8, 1, 3, 3, 1, 1)  RV_SMALL(1, 8, 3, 3, 1, 1)
            RV_SMALL(1, 48, 3, 3, 1, 1) RV_SMALL(48, 1, 3, 3, 1, 1)      // ConvStack layer 0 (model/onset_frame_VAT.py:328)
        } else if (mode == 1) {
            RV_SMALL(1, 16, 1, 1, 1, 0) RV_SMALL(16, 1, 1, 1, 1, 0)
        }
#undef RV_SMALL
        rv_set_error("rv_conv_fwd: no small-channel kernel for mode %d Cin=%d Cout=%d", mode, Cin, Cout);
        return RV_EUNSUPPORTED;
    }
    ConvArgs a;
    a.in = in; a.in_ld = in_ld; a.H = H; a.W = W; a.out = out; a.out_ld = out_ld; a.Ho = Ho; a.Wo = Wo;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.wpack = wpack; a.bias = bias; a.bn_sums = bn_sums;
    a.bn_z = bn.z; a.bn_z_ld = bn.z_ld; a.bn_coef = bn.coef; a.bn_slope = bn.slope;
    a.nchunk = Cin / (4 * R);
    a.accumulate = accumulate;
    a.vec_store = ((out_ld & 3) == 0) && ((((uintptr_t)out) & 15) == 0);
    RV_CHECK_ARG((in_ld % R) == 0 && ((((uintptr_t)in) & (4 * R - 1)) == 0), "rv_conv_fwd: input not %d-byte aligned", 4 * R);
    if (mode == 3) {
        RV_CHECK_ARG(Cout % 16 == 0, "rv_conv_fwd: scatter conv needs Cout %% 16 == 0");
        a.ntile_n = 4 * Cout / 16;
        a.Ph = H + (Ho - 2 * H); a.Pw = W + (Wo - 2 * W);
    } else {
        a.ntile_n = (Cout + 15) / 16;
        a.Ph = Ho; a.Pw = Wo;
    }
    a.npix = (long)B * a.Ph * a.Pw;
    a.fd_pw = fastdiv_make((unsigned)a.Pw); a.fd_plane = fastdiv_make((unsigned)(a.Ph * a.Pw)); a.fd_w = fastdiv_make((unsigned)W);
    RV_CHECK_ARG(a.npix < (1L << 31), "rv_conv_fwd: more than 2^31 pixels");
    // algo: 0 = library default, 1 = LDS-free direct kernel, 2 = LDS/DMA-pipelined kernel (3x3 only),
    // 0x100|NT<<4|MT = direct kernel with that register tile, 0x200|NT<<4|MTW = LDS kernel with that tile (4 waves),
    // 0x300|NT<<4|MTW = LDS kernel with 8 waves per workgroup (two per SIMD), 0x400|... = 16 waves (four per SIMD),
    // 0x700|NT<<4|MTW = 12 waves (three per SIMD) with MTW in {3, 5, 6}: the band sizes that fit whole 57/114/229-pixel rows.
    // TH<<12 on top of a forced LDS tile: rows per band (<= what the tile slots hold; 0 = as many as they hold).
    // Forced tiles that do not fit the shape return RV_EUNSUPPORTED (the host autotuner skips them).
    // RV_ALGO_BF16 (bit 20): bf16 operands on the persistent 3x3 kernel (16-channel chunks; ignored -- fp32 -- everywhere else)
    const bool bf = ((algo >> 20) & 1) && mode == 0 && R == 4;
    algo &= ~(1 << 20);
    const int fam = (algo >> 8) & 15, f_nt = (algo >> 4) & 15, f_mt = algo & 15, f_th = (algo >> 12) & 255;
    if (fam == 6 || fam == 10 || fam == 12) {   // Winograd F(2x2,3x3): 0x6NM = 8 waves, 0xANM = 8 waves + half-chunk patch, 0xCNM = 12 waves + half-chunk patch
        if (mode != 0 || R != 4) { rv_set_error("rv_conv_fwd: the Winograd kernel needs a 3x3 conv with Cin %% 16 == 0"); return RV_EUNSUPPORTED; }
        const int rcw = fam == 6 ? launch_conv3x3_wino<8>(a, f_nt, f_mt, f_th, st)
                                 : (fam == 10 ? launch_conv3x3_wino<8, true>(a, f_nt, f_mt, f_th, st) : launch_conv3x3_wino<12, true>(a, f_nt, f_mt, f_th, st));
        if (rcw != RV_OK) { rv_set_error("rv_conv_fwd: forced Winograd tile NT=%d MTW=%d TH=%d does not fit", f_nt, f_mt, f_th); return rcw; }
        RV_LAUNCH_CHECK("rv_conv_fwd(winograd)");
        *sums_done = true;
        return RV_OK;
    }
    if (fam == 14) {   // 0xENM (round 6 EXPERIMENT, never in the shipped table): four-wave half-CU workgroups -- <= 78 KiB of LDS, <= 256 registers, 512 slots
        if (mode != 0 || R != 4) { rv_set_error("rv_conv_fwd: the Winograd kernel needs a 3x3 conv with Cin %% 16 == 0"); return RV_EUNSUPPORTED; }
        const int rcw = launch_conv3x3_wino<4, true>(a, f_nt, f_mt, f_th, st, (size_t)78 * 1024, 512);
        if (rcw != RV_OK) { rv_set_error("rv_conv_fwd: forced half-CU Winograd tile NT=%d MTW=%d TH=%d does not fit", f_nt, f_mt, f_th); return rcw; }
        RV_LAUNCH_CHECK("rv_conv_fwd(winograd, half CU)");
        *sums_done = true;
        return RV_OK;
    }
    if (fam == 8 || fam == 9 || fam == 11 || fam == 13) {   // software-pipelined Winograd (conv_wino2.hip): 0x8NM / 0x9NM = 8 waves, full / half-chunk patch; 0xBNM = 4 waves; 0xDNM = 12 waves, half-chunk patch
        if (mode != 0 || R != 4) { rv_set_error("rv_conv_fwd: the Winograd kernel needs a 3x3 conv with Cin %% 16 == 0"); return RV_EUNSUPPORTED; }
        const int rcw = rv_launch_conv3x3_wino2(a, f_nt, f_mt, fam == 13 ? 12 : ((fam == 8 || fam == 9) ? 8 : 4), (fam == 9 || fam == 13) ? 1 : 0, f_th, st);
        if (rcw != RV_OK) { rv_set_error("rv_conv_fwd: forced pipelined Winograd tile NT=%d MTW=%d TH=%d does not fit", f_nt, f_mt, f_th); return rcw; }
        RV_LAUNCH_CHECK("rv_conv_fwd(winograd, pipelined)");
        *sums_done = true;
        return RV_OK;
    }
    if (mode == 0 && algo != 1 && fam != 1) {
        const bool forced = (fam >= 2 && fam <= 4) || fam == 7;
        int rc3 = forced ? launch_conv3x3_lds(a, R, st, f_nt, f_mt, fam == 7 ? 12 : (fam == 4 ? 16 : (fam == 3 ? 8 : 4)), f_th, bf)
                         : launch_conv3x3_lds(a, R, st, 0, 0, 4, 0, bf);
        if (rc3 == RV_OK) { RV_LAUNCH_CHECK("rv_conv_fwd(lds)"); *sums_done = true; return RV_OK; }
        if (forced) { rv_set_error("rv_conv_fwd: forced LDS tile NT=%d MTW=%d does not fit", f_nt, f_mt); return RV_EUNSUPPORTED; }
    }
    int NT, MT;
    choose_tiles((a.npix + 15) / 16, a.ntile_n, &NT, &MT);
    const bool ksplit = fam == 5;              // 0x5NM: direct kernel, K loop split over the four waves (1x1 / 2x2 modes, 16-channel chunks)
    if (ksplit) {
        if (mode == 0 || R != 4 || f_nt < 1 || a.ntile_n % f_nt || a.nchunk * (mode == 2 ? 4 : 1) < 4) {
            rv_set_error("rv_conv_fwd: forced split-K direct tile NT=%d MT=%d invalid here", f_nt, f_mt);
            return RV_EUNSUPPORTED;
        }
        NT = f_nt; MT = f_mt;
    }
    if (fam == 1) {
        if (f_nt < 1 || a.ntile_n % f_nt || !(f_mt == 1 || f_mt == 2 || f_mt == 4) || f_nt > 4) {
            rv_set_error("rv_conv_fwd: forced direct tile NT=%d MT=%d invalid", f_nt, f_mt);
            return RV_EUNSUPPORTED;
        }
        NT = f_nt; MT = f_mt;
    }
    int rc = RV_EUNSUPPORTED;
    if (R == 4) {
        if (mode == 0) rc = launch_conv_rt<3, 3, 1, 1, 4, false>(a, NT, MT, st);
        else if (mode == 1) rc = launch_conv_rt<1, 1, 1, 0, 4, false>(a, NT, MT, st, ksplit);
        else if (mode == 2) rc = launch_conv_rt<2, 2, 2, 0, 4, false>(a, NT, MT, st, ksplit);
        else rc = launch_conv_rt<1, 1, 1, 0, 4, true>(a, NT, MT, st, ksplit);
    } else {
        if (mode == 0) rc = launch_conv_rt<3, 3, 1, 1, 2, false>(a, NT, MT, st);
        else if (mode == 1) rc = launch_conv_rt<1, 1, 1, 0, 2, false>(a, NT, MT, st);
        else if (mode == 2) rc = launch_conv_rt<2, 2, 2, 0, 2, false>(a, NT, MT, st);
        else rc = launch_conv_rt<1, 1, 1, 0, 2, true>(a, NT, MT, st);
    }
    if (rc != RV_OK) { rv_set_error("rv_conv_fwd: no kernel instance for NT=%d MT=%d", NT, MT); return rc; }
    RV_LAUNCH_CHECK("rv_conv_fwd");
    return RV_OK;
}

// wgrad partitioning shared by the workspace query and the launch
struct WgradPlan { bool small; int TA, TB, nga, ngb, rows_per_wave, nparts, nw; bool wino; };

// per-shape launch partition of wgrad_mfma_k chosen by the host autotuner (rv_conv_wgrad_set_plan): waves per workgroup and
// workgroups on the chip.  Keyed like rv_conv_wgrad_workspace_bytes, whose result depends on it.
struct WgradTune { int taps, B, Hv, Ca, Cb, nw, wgs; };
static WgradTune g_wgrad_tune[256];
static int g_wgrad_ntune = 0;
static const WgradTune* wgrad_tuned(int taps, int B, int Hv, int Ca, int Cb) {
    for (int i = 0; i < g_wgrad_ntune; ++i) {
        const WgradTune& t = g_wgrad_tune[i];
        if (t.taps == taps && t.B == B && t.Hv == Hv && t.Ca == Ca && t.Cb == Cb) return &t;
    }
    return nullptr;
}

static WgradPlan wgrad_plan(int taps, int B, int Hv, int Ca, int Cb) {
    WgradPlan p;
    p.nw = 0;
    p.wino = false;
    const int nrows = B * Hv;
    p.small = (Ca * Cb * taps <= 144) && (Ca < 8 || Cb < 8);
    if (p.small) {
        p.TA = p.TB = p.nga = p.ngb = 1; p.rows_per_wave = 0;
        { static int np = getenv("RV_WGS_NPARTS") ? atoi(getenv("RV_WGS_NPARTS")) : 4096; p.nparts = nrows < np ? nrows : np; }
        return p;
    }
    // channel-group shape (TA x TB tiles of 16): the one that pads the channel counts least (48 = 3 x 16, not 2 x 32);
    // ties go to the larger group (more register reuse per LDS read)
    long best = -1;
    p.TA = p.TB = 1;
    for (int ta = 2; ta >= 1; --ta)
        for (int tb = 2; tb >= 1; --tb) {
            const long padded = (long)cdiv(Ca, ta * 16) * ta * cdiv(Cb, tb * 16) * tb;
            if (best < 0 || padded < best) { best = padded; p.TA = ta; p.TB = tb; }
        }
    p.nga = cdiv(Ca, p.TA * 16); p.ngb = cdiv(Cb, p.TB * 16);
    static int want_total = 0;
    if (!want_total) { const char* e = getenv("RV_WGRAD_WGS"); want_total = e ? atoi(e) : 256; }
    int total = want_total;
    p.wino = false;
    if (const WgradTune* t = wgrad_tuned(taps, B, Hv, Ca, Cb)) {
        if (t->wgs > 0) total = t->wgs;
        p.nw = t->nw & 15;
        p.wino = (t->nw & 16) != 0 && taps == 9 && (Hv & 1) == 0;      // Winograd F(3x3, 2x2) form: tile rows = row pairs
    }
    int want = total / (p.nga * p.ngb);        // one resident workgroup per CU (register-limited): exactly one block wave, no tail
    if (want < 4) want = 4;
    const int units = p.wino ? nrows / 2 : nrows;     // what a workgroup walks: rows, or row pairs
    if (want > units) want = units;
    p.rows_per_wave = cdiv(units, want);     // rows (row pairs) per WORKGROUP for the LDS-staged kernels
    p.nparts = cdiv(units, p.rows_per_wave);
    return p;
}

// Workspace (bytes) rv_conv_wgrad needs for the given problem.
// a one-channel input against a wide dY (ConvStack layer 0, 1 -> 48) runs as Cb/16 launches of the (1, 16) VALU kernel on
// 16-channel slices of dY
static inline bool wgrad_sliced(int taps, int Ca, int Cb) { return taps == 9 && Ca == 1 && Cb > 16 && Cb % 16 == 0; }

// Autotuner hook: the partition rv_conv_wgrad uses for this shape from now on (nw: 4 or 8 waves per workgroup, 0 = default;
// wgs: workgroups on the chip, 0 = default 256).  Changes what rv_conv_wgrad_workspace_bytes returns for the shape.
int rv_conv_wgrad_set_plan(int taps, int B, int Hv, int Ca, int Cb, int nw, int wgs) {
    RV_CHECK_ARG(nw == 0 || nw == 4 || nw == 8 || nw == 24, "rv_conv_wgrad_set_plan: nw must be 0, 4, 8 or 24 (8 waves, Winograd form)");
    RV_CHECK_ARG(wgs == 0 || (wgs >= 32 && wgs <= 4096), "rv_conv_wgrad_set_plan: wgs out of range");
    WgradTune* t = const_cast<WgradTune*>(wgrad_tuned(taps, B, Hv, Ca, Cb));
    if (!t) {
        RV_CHECK_ARG(g_wgrad_ntune < 256, "rv_conv_wgrad_set_plan: table full");
        t = &g_wgrad_tune[g_wgrad_ntune++];
    }
    *t = WgradTune{taps, B, Hv, Ca, Cb, nw, wgs};
    return RV_OK;
}

long rv_conv_wgrad_workspace_bytes(int taps, int B, int Hv, int Ca, int Cb) {
    if (wgrad_sliced(taps, Ca, Cb)) Cb = 16;
    WgradPlan p = wgrad_plan(taps, B, Hv, Ca, Cb);
    return (long)p.nparts * ((long)taps * Ca * Cb + Cb) * 4;
}

// G[tap][a][b] = sum_p U[f(p,tap)][a] * V[p][b], db[b] = sum_p V[p][b]; results scattered to
//   dw[a*s_a + b*s_b + (flip ? taps-1-tap : tap)],  dbias[b]
// mode: 0 = 3x3 s1 p1 (U = conv input, V = dY), 1 = 1x1, 2 = 2x2 s2 (U gathered at 2p+tap)
static int conv_wgrad_impl(int mode, const float* U, int u_ld, int Hu, int Wu, int Ca, const float* V, int v_ld, int Hv, int Wv,
                           int Cb, int B, float* dw, long s_a, long s_b, int flip, float* dbias, int accumulate, void* workspace,
                           long workspace_bytes, void* stream, WreduceEntry* defer, int nseg = 1, const float* const* Useg = nullptr,
                           const float* const* Vseg = nullptr);

int rv_conv_wgrad(int mode, const float* U, int u_ld, int Hu, int Wu, int Ca, const float* V, int v_ld, int Hv, int Wv,
                  int Cb, int B, float* dw, long s_a, long s_b, int flip, float* dbias, int accumulate, void* workspace,
                  long workspace_bytes, void* stream) {
    return conv_wgrad_impl(mode, U, u_ld, Hu, Wu, Ca, V, v_ld, Hv, Wv, Cb, B, dw, s_a, s_b, flip, dbias, accumulate, workspace,
                           workspace_bytes, stream, nullptr);
}

// Deferred form: launches only the partial-sum kernel and writes the description of the pending reduction (which ADDS into
// dw / dbias) to *entry_host (rv_wgrad_table_entry_bytes() bytes of host memory).  The workspace must stay untouched until
// rv_wgrad_reduce_table has run.  Returns the number of workgroups the reduction needs (> 0) or a negative status.
long rv_conv_wgrad_deferred(int mode, const float* U, int u_ld, int Hu, int Wu, int Ca, const float* V, int v_ld, int Hv, int Wv,
                            int Cb, int B, float* dw, long s_a, long s_b, int flip, float* dbias, void* workspace,
                            long workspace_bytes, void* entry_host, void* stream) {
    if (!entry_host) { rv_set_error("rv_conv_wgrad_deferred: null entry"); return RV_EINVAL; }
    WreduceEntry* e = (WreduceEntry*)entry_host;
    const int rc = conv_wgrad_impl(mode, U, u_ld, Hu, Wu, Ca, V, v_ld, Hv, Wv, Cb, B, dw, s_a, s_b, flip, dbias, 1, workspace,
                                   workspace_bytes, stream, e);
    return rc != RV_OK ? (long)rc : e->block0;
}

// Segmented forms: the reduction runs over nseg (1..4) runs of Bseg images, run s at U[s] / V[s] (identical geometry and strides) -- the
// (input, dY) pairs of the same layer from several backward passes in ONE launch (plan / workspace: those of B = nseg * Bseg images).
// Only the MFMA kernels take segments (wgrad_mfma_k, wgrad_wino_k); RV_EUNSUPPORTED for the small-channel layers: launch those per pass.
int rv_conv_wgrad_seg(int mode, int nseg, const float* const* U, const float* const* V, int u_ld, int Hu, int Wu, int Ca, int v_ld, int Hv,
                      int Wv, int Cb, int Bseg, float* dw, long s_a, long s_b, int flip, float* dbias, int accumulate, void* workspace,
                      long workspace_bytes, void* stream) {
    RV_CHECK_ARG(nseg >= 1 && nseg <= 4 && U && V && Bseg > 0, "rv_conv_wgrad_seg: 1..4 segments");
    return conv_wgrad_impl(mode, U[0], u_ld, Hu, Wu, Ca, V[0], v_ld, Hv, Wv, Cb, nseg * Bseg, dw, s_a, s_b, flip, dbias, accumulate,
                           workspace, workspace_bytes, stream, nullptr, nseg, U, V);
}
long rv_conv_wgrad_deferred_seg(int mode, int nseg, const float* const* U, const float* const* V, int u_ld, int Hu, int Wu, int Ca, int v_ld,
                                int Hv, int Wv, int Cb, int Bseg, float* dw, long s_a, long s_b, int flip, float* dbias, void* workspace,
                                long workspace_bytes, void* entry_host, void* stream) {
    if (!entry_host) { rv_set_error("rv_conv_wgrad_deferred_seg: null entry"); return RV_EINVAL; }
    if (!(nseg >= 1 && nseg <= 4 && U && V && Bseg > 0)) { rv_set_error("rv_conv_wgrad_deferred_seg: 1..4 segments"); return RV_EINVAL; }
    WreduceEntry* e = (WreduceEntry*)entry_host;
    const int rc = conv_wgrad_impl(mode, U[0], u_ld, Hu, Wu, Ca, V[0], v_ld, Hv, Wv, Cb, nseg * Bseg, dw, s_a, s_b, flip, dbias, 1,
                                   workspace, workspace_bytes, stream, e, nseg, U, V);
    return rc != RV_OK ? (long)rc : e->block0;
}

long rv_wgrad_table_entry_bytes(void) { return (long)sizeof(WreduceEntry); }

// in place: per-entry workgroup counts -> exclusive prefix; returns the total number of workgroups
long rv_wgrad_table_finalize(void* table_host, int count) {
    WreduceEntry* tab = (WreduceEntry*)table_host;
    long total = 0;
    for (int i = 0; i < count; ++i) { const long n = tab[i].block0; tab[i].block0 = total; total += n; }
    return total;
}

// table_dev: DEVICE copy of a finalized table
int rv_wgrad_reduce_table(const void* table_dev, int count, long total_blocks, void* stream) {
    if (count <= 0 || total_blocks <= 0) return RV_OK;
    RV_CHECK_ARG(table_dev, "rv_wgrad_reduce_table: null table");
    static const int plain = getenv("RV_ABL_WREDUCE_PLAIN") ? atoi(getenv("RV_ABL_WREDUCE_PLAIN")) : 0;      // (ablation build only)
    hipLaunchKernelGGL(wgrad_reduce_table_k, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const WreduceEntry*)table_dev, count, plain);
    RV_LAUNCH_CHECK("rv_wgrad_reduce_table");
    return RV_OK;
}

static int conv_wgrad_impl(int mode, const float* U, int u_ld, int Hu, int Wu, int Ca, const float* V, int v_ld, int Hv, int Wv,
                           int Cb, int B, float* dw, long s_a, long s_b, int flip, float* dbias, int accumulate, void* workspace,
                           long workspace_bytes, void* stream, WreduceEntry* defer, int nseg, const float* const* Useg,
                           const float* const* Vseg) {
    // mode bit 8 (RV_WGRAD_BF16): bf16 operands on the matrix pipe for the 3x3 MFMA kernel (opt-in experiment; ignored elsewhere)
    const bool want_bf = (mode & 0x100) != 0;
    mode &= 0xff;
    hipStream_t st = (hipStream_t)stream;
    RV_CHECK_ARG(mode >= 0 && mode <= 2, "rv_conv_wgrad: bad mode %d", mode);
    const int taps = mode == 0 ? 9 : (mode == 1 ? 1 : 4);
    if (nseg > 1 && (wgrad_sliced(taps, Ca, Cb) || B % nseg)) {
        rv_set_error("rv_conv_wgrad_seg: no segmented form for this layer (taps %d, %d -> %d channels)", taps, Ca, Cb);
        return RV_EUNSUPPORTED;
    }
    if (wgrad_sliced(taps, Ca, Cb)) {
        if (defer) { rv_set_error("rv_conv_wgrad_deferred: the sliced 1 -> %d channel case has no deferred form", Cb); return RV_EUNSUPPORTED; }
        for (int c0 = 0; c0 < Cb; c0 += 16) {
            const int rc = rv_conv_wgrad(mode, U, u_ld, Hu, Wu, Ca, V + c0, v_ld, Hv, Wv, 16, B, dw + c0 * s_b, s_a, s_b, flip,
                                         dbias ? dbias + c0 : nullptr, accumulate, workspace, workspace_bytes, stream);
            if (rc != RV_OK) return rc;
        }
        return RV_OK;
    }
    WgradArgs a;
    a.U = U; a.u_ld = u_ld; a.Hu = Hu; a.Wu = Wu; a.Ca = Ca; a.V = V; a.v_ld = v_ld; a.Hv = Hv; a.Wv = Wv; a.Cb = Cb;
    a.B = B; a.want_bias = dbias != nullptr;
    a.nseg = nseg; a.Bseg = B / nseg;
    for (int i = 0; i < 4; ++i) { a.Useg[i] = nseg > 1 ? Useg[i < nseg ? i : 0] : U; a.Vseg[i] = nseg > 1 ? Vseg[i < nseg ? i : 0] : V; }
    a.ablate = getenv("RV_ABLATE") ? atoi(getenv("RV_ABLATE")) : 0;
    a.dbg = getenv("RV_DBG_PTR") ? (unsigned long long*)strtoull(getenv("RV_DBG_PTR"), nullptr, 10) : nullptr;
    a.pstride = (long)taps * Ca * Cb + Cb;
    a.fd_vplane = fastdiv_make((unsigned)(Hv * Wv)); a.fd_wv = fastdiv_make((unsigned)Wv);
    RV_CHECK_ARG((long)B * Hv * Wv < (1L << 31), "rv_conv_wgrad: more than 2^31 pixels");
    WgradPlan plan = wgrad_plan(taps, B, Hv, Ca, Cb);
    size_t wino_lds = 0;
    if (plan.wino) {
        // the Winograd form needs six input rows and four dY rows in LDS at once (whole 1 KiB DMA pieces) and the direct kernel's geometry
        const int WT4 = (((Wv + 1) / 2) + 3) & ~3, ppa = 64 / (plan.TA * 4), ppb = 64 / (plan.TB * 4);
        const int UPp = cdiv(2 * WT4 + 2, ppa) * ppa, VPp = cdiv(2 * WT4, ppb) * ppb;
        wino_lds = ((size_t)6 * UPp * plan.TA * 16 + (size_t)4 * VPp * plan.TB * 16) * sizeof(float);
        const int nh_ = (plan.TA == 2 && plan.TB == 2) ? 2 : 1, nxg_ = 8 / nh_;
        const size_t fold = (size_t)nh_ * (nxg_ / 2) * (9 * (plan.TA / nh_) * plan.TB + plan.TB) * 4 * 64 * sizeof(float);
        if (wino_lds < fold) wino_lds = fold;
        const long ub = (((long)a.Bseg * Hu * Wu - 1) * u_ld + Ca) * 4, vb = (((long)a.Bseg * Hv * Wv - 1) * v_ld + Cb) * 4;   // per segment
        a.u_bytes = (unsigned)ub; a.v_bytes = (unsigned)vb;   // (the staging loads address the views with 30-bit offsets: see the kernel)
        if (want_bf || mode != 0 || Hu != Hv || Wu != Wv || wino_lds > 160 * 1024 || ub >= 0x3f000000L || vb >= 0x3f000000L) {
            plan.wino = false;                             // this launch runs the direct form on the same partition, in ROWS
            plan.rows_per_wave *= 2;                       // (the workspace was sized for this many partial sums: same layout)
            plan.nparts = cdiv((long)B * Hv, plan.rows_per_wave);
        }
    }
    RV_CHECK_ARG(workspace_bytes >= (long)plan.nparts * a.pstride * 4, "rv_conv_wgrad: workspace too small");
    a.part = (float*)workspace;
    a.nparts = plan.nparts; a.rows_per_wave = plan.rows_per_wave; a.ngb = plan.ngb;
    if (plan.small && nseg > 1) {
        rv_set_error("rv_conv_wgrad_seg: the small-channel kernels take one segment");
        return RV_EUNSUPPORTED;
    }
    if (plan.small) {
        dim3 grid(cdiv(a.nparts, 4)), blk(256);
#define RV_WS(ca, cb, kh, kw, ss, pp)                                                             \
    if (Ca == ca && Cb == cb) {                                                                  \
        hipLaunchKernelGGL((wgrad_small_k<ca, cb, kh, kw, ss, pp>), grid, blk, 0, st, a);        \
        goto reduce;                                                                             \
    }
        if (mode == 0 && Ca == 8 && (Cb == 1 || Cb == 2) && (u_ld & 3) == 0 && ((((uintptr_t)U) & 15) == 0) && Hu == Hv && Wu == Wv) {
            // sliding-window kernel: one partial per wave = (image, band of `rows` rows, strip of 32 columns)
            static const int sw_env = getenv("RV_WGRAD_SW") ? atoi(getenv("RV_WGRAD_SW")) : 1;
            if (sw_env) {
                const int nstrip = cdiv(Wv, 32);
                int rows = 16;
                while (rows < Hv && (long)B * cdiv(Hv, rows) * nstrip > plan.nparts) rows *= 2;   // the workspace holds plan.nparts partials
                const int nband = cdiv(Hv, rows);
                if ((long)B * nband * nstrip <= plan.nparts) {          // (tiny images: the flat kernel below)
                    a.nparts = B * nband * nstrip;
                    dim3 g2(cdiv(a.nparts, 4));
                    if (Cb == 2) hipLaunchKernelGGL((wgrad_small_sw_k<2>), g2, blk, 0, st, a, rows, nstrip, nband);
                    else hipLaunchKernelGGL((wgrad_small_sw_k<1>), g2, blk, 0, st, a, rows, nstrip, nband);
                    goto reduce;
                }
            }
        }
        if (mode == 0 && Ca == 1 && Cb == 16 && (v_ld & 3) == 0 && ((((uintptr_t)V) & 15) == 0) && Hu == Hv && Wu == Wv) {
            static const int c1_env = getenv("RV_WGRAD_SW") ? atoi(getenv("RV_WGRAD_SW")) : 1;
            if (c1_env) {
                const int nstrip = cdiv(Wv, 16);
                int rows = 32;
                while (rows < Hv && (long)B * cdiv(Hv, rows) * nstrip > plan.nparts) rows *= 2;
                const int nband = cdiv(Hv, rows);
                if ((long)B * nband * nstrip <= plan.nparts) {
                    a.nparts = B * nband * nstrip;
                    hipLaunchKernelGGL(wgrad_cin1_k, dim3(cdiv(a.nparts, 4)), blk, 0, st, a, rows, nstrip, nband);
                    goto reduce;
                }
            }
        }
        if (mode == 0) { RV_WS(1, 16, 3, 3, 1, 1) RV_WS(8, 2, 3, 3, 1, 1) RV_WS(8, 1, 3, 3, 1, 1) }
        else if (mode == 1) { RV_WS(1, 16, 1, 1, 1, 0) }
#undef RV_WS
        rv_set_error("rv_conv_wgrad: no small-channel kernel for mode %d Ca=%d Cb=%d", mode, Ca, Cb);
        return RV_EUNSUPPORTED;
    } else {
        const int TA = plan.TA, TB = plan.TB;
        dim3 grid(a.nparts, plan.nga * plan.ngb), blk(256);
        const int KH = mode == 0 ? 3 : (mode == 1 ? 1 : 2), SS = mode == 2 ? 2 : 1;
        bool bf = want_bf && mode == 0;
        int Wv4 = bf ? ((Wv + 15) & ~15) : ((Wv + 3) & ~3);
        int UP = SS * (Wv4 - 1) + KH;
        const int nslot = SS == 1 ? KH + 1 : 2 * KH;
        size_t lds = ((size_t)nslot * UP * TA * 16 + (size_t)2 * Wv4 * TB * 16) * sizeof(float);
        if (bf && lds > 156 * 1024) {                      // the 16-pixel padding does not fit: fp32 kernel
            bf = false; Wv4 = (Wv + 3) & ~3; UP = SS * (Wv4 - 1) + KH;
            lds = ((size_t)nslot * UP * TA * 16 + (size_t)2 * Wv4 * TB * 16) * sizeof(float);
        }
        static int nw_env = 0;
        if (!nw_env) { const char* e = getenv("RV_WGRAD_NW"); nw_env = (e && atoi(e) == 4) ? 4 : 8; }
        const int nw = plan.nw ? plan.nw : nw_env;
        // fold tree: (x groups / 2) slots per tile half, each [taps * TAW * TB + TB] accumulators x 64 lanes
        const int nh_ = (nw == 8 && TA == 2 && TB == 2) ? 2 : 1, nxg_ = nw / nh_;
        const size_t fold = (size_t)nh_ * (nxg_ / 2) * ((mode == 0 ? 9 : (mode == 1 ? 1 : 4)) * (TA / nh_) * TB + TB) * 4 * 64 * sizeof(float);
        if (lds < fold) lds = fold;
        RV_CHECK_ARG(lds <= 160 * 1024, "rv_conv_wgrad: row of %d pixels x %d channels does not fit LDS", Wv, TA * 16);
#define RV_WG1B(ta, tb)                                                                           \
    do {                                                                                         \
        if (nw == 8) {                                                                           \
            auto kern = wgrad_mfma_k<3, 3, 1, 1, ta, tb, 8, true>;                               \
            if (lds > 64 * 1024)                                                                 \
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, a);                               \
        } else {                                                                                 \
            auto kern = wgrad_mfma_k<3, 3, 1, 1, ta, tb, 4, true>;                               \
            if (lds > 64 * 1024)                                                                 \
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL(kern, grid, blk, lds, st, a);                                     \
        }                                                                                        \
    } while (0)
#define RV_WG1(kh, kw, ss, pp, ta, tb)                                                            \
    do {                                                                                         \
        if (nw == 8) {                                                                           \
            auto kern = wgrad_mfma_k<kh, kw, ss, pp, ta, tb, 8>;                                 \
            if (lds > 64 * 1024)                                                                 \
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, a);                               \
        } else {                                                                                 \
            auto kern = wgrad_mfma_k<kh, kw, ss, pp, ta, tb, 4>;                                 \
            if (lds > 64 * 1024)                                                                 \
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL(kern, grid, blk, lds, st, a);                                     \
        }                                                                                        \
    } while (0)
#define RV_WG(kh, kw, ss, pp)                                                                     \
    do {                                                                                         \
        if (TA == 1 && TB == 1) RV_WG1(kh, kw, ss, pp, 1, 1);                                    \
        else if (TA == 1 && TB == 2) RV_WG1(kh, kw, ss, pp, 1, 2);                               \
        else if (TA == 2 && TB == 1) RV_WG1(kh, kw, ss, pp, 2, 1);                               \
        else RV_WG1(kh, kw, ss, pp, 2, 2);                                                       \
    } while (0)
        if (plan.wino) {
#define RV_WW(ta, tb)                                                                             \
    do {                                                                                         \
        auto kern = wgrad_wino_k<ta, tb, 8>;                                                     \
        if (wino_lds > 64 * 1024)                                                                \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino_lds); \
        hipLaunchKernelGGL(kern, grid, dim3(512), wino_lds, st, a);                              \
    } while (0)
            if (TA == 1 && TB == 1) RV_WW(1, 1);
            else if (TA == 1 && TB == 2) RV_WW(1, 2);
            else if (TA == 2 && TB == 1) RV_WW(2, 1);
            else RV_WW(2, 2);
#undef RV_WW
        } else if (bf) {
            if (TA == 1 && TB == 1) RV_WG1B(1, 1);
            else if (TA == 1 && TB == 2) RV_WG1B(1, 2);
            else if (TA == 2 && TB == 1) RV_WG1B(2, 1);
            else RV_WG1B(2, 2);
        } else if (mode == 0) RV_WG(3, 3, 1, 1);
        else if (mode == 1) RV_WG(1, 1, 1, 0);
        else RV_WG(2, 2, 2, 0);
#undef RV_WG
#undef RV_WG1
#undef RV_WG1B
    }
reduce:
    RV_LAUNCH_CHECK("rv_conv_wgrad");
#ifdef RV_ABLATION
    static const int skip_reduce = getenv("RV_ABL_SKIP_WREDUCE") ? atoi(getenv("RV_ABL_SKIP_WREDUCE")) : 0;   // timing ablation (wrong results)
#else
    const int skip_reduce = 0;
#endif
    if (!skip_reduce) {
        WreduceArgs r;
        r.part = a.part; r.pstride = a.pstride; r.nparts = a.nparts; r.taps = taps; r.Ca = Ca; r.Cb = Cb;
        r.dw = dw; r.s_a = s_a; r.s_b = s_b; r.flip = flip; r.dbias = dbias; r.accumulate = accumulate;
        long nel = (long)taps * Ca * Cb + (dbias ? Cb : 0);
        const int el = (cdiv(nel, 64) >= 256 || a.nparts <= 8) ? 64 : ((cdiv(nel, 16) >= 256 || a.nparts <= 32) ? 16 : 4);
        static const int v4_env = getenv("RV_WREDUCE_V4") ? atoi(getenv("RV_WREDUCE_V4")) : 1;
        const int vec4 = v4_env && el >= 16 && (nel & 3) == 0 && (a.pstride & 3) == 0 && (((uintptr_t)a.part) & 15) == 0;
        // deferred (table) form of a large gradient: 1024 elements per workgroup, every thread walks ALL partials of its four elements
        // -- 4 KiB contiguous per partial and workgroup instead of 256-byte pieces at partial stride; the table's other entries fill the chip
        static const int wide_env = getenv("RV_WREDUCE_WIDE") ? atoi(getenv("RV_WREDUCE_WIDE")) : 1;
        if (defer && vec4 && wide_env && nel >= 8192) {
            defer->a = r; defer->block0 = cdiv(nel, 1024); defer->el = 1024; defer->vec4 = 1;
            return RV_OK;
        }
        if (defer) {
            defer->a = r; defer->block0 = cdiv(nel, el); defer->el = el; defer->vec4 = vec4;
            return RV_OK;
        }
        if (vec4 && el == 64) hipLaunchKernelGGL((wgrad_reduce_k<64, true>), dim3(cdiv(nel, 64)), dim3(256), 0, st, r);
        else if (vec4 && el == 16) hipLaunchKernelGGL((wgrad_reduce_k<16, true>), dim3(cdiv(nel, 16)), dim3(256), 0, st, r);
        else if (el == 64) hipLaunchKernelGGL(wgrad_reduce_k<64>, dim3(cdiv(nel, 64)), dim3(256), 0, st, r);
        else if (el == 16) hipLaunchKernelGGL(wgrad_reduce_k<16>, dim3(cdiv(nel, 16)), dim3(256), 0, st, r);
        else hipLaunchKernelGGL(wgrad_reduce_k<4>, dim3(cdiv(nel, 4)), dim3(256), 0, st, r);
        RV_LAUNCH_CHECK("rv_conv_wgrad(reduce)");
    }
    return RV_OK;
}

}  // extern "C"
