// Fused 31-frame local multi-head attention (reference MutliHeadAttention1D.forward,
// model/UNet_onset.py:56-91) for gfx950 -- no [B, L, F, 31] unfold is ever materialised.
//
//   energy[b,t,g,w] = sum_f q[b,t,g,f] * (k[b,t+w-15,g,f] + rel[g*dh+f, w])   (k = 0 outside [0,L): the
//                     reference zero-pads x and W_k has no bias, so padded slots score q.rel -- unmasked)
//   att = softmax_w(energy)          (no 1/sqrt(d) scaling)
//   out[b,t,g,f]    = sum_w att[b,t,g,w] * v[b,t+w-15,g,f]
//
// q, k, v (and dq, dk, dv): rows of G*dh floats with an explicit row stride `ld` (they may be column slices of one
// fused [B*L, 3F] projection buffer); out, dout: [B, L, F] contiguous; relT: [31, F] (the parameter [F, 31]
// transposed -- the host packs it once per optimiser step); att, de: [B, L, G, 31].
//
// One workgroup (5 waves) = (batch b, 16-frame tile, head g).  Both contractions run on v_mfma_f32_16x16x4_f32:
//   scores  S[16 x 80] = X[16 x dh] . Y^T, Y = [ 46-row key window (padded to 48) ; rel^T (31 rows padded to 32) ]:
//           one 16x16 tile per wave, operands read from LDS with ds_read_b128 (a lane's 4 floats feed 4 MFMAs);
//           energy[t][w] = S[t][t+w] + S[t][48+w]
//   apply   O[16 x dh] = A[16 x K] . Y[K x dh] with A the BANDED attention matrix A[t][t+w] = att[t][w] (K = 48),
//           for dq A = [ band(de) | de ] against [ key window ; rel^T ] (K = 80); output tiles round-robin over waves
// The head dimension is padded to a multiple of 16 in LDS (229 -> 240, zero filled).  ~0.5 GFLOP per forward, so the
// kernels are bound by staging the windows (each row is read 46/16 times, from L2).
#include "common.h"
#include <stdlib.h>
#include <mutex>

// Timing ablations of attn_fwd_k (tools/attn_ablate.sh; wrong results by design; 0 in every product build):
//   1 no staging (q tile, key / value windows, rel^T) | 2 no score MFMAs | 4 no softmax | 8 no banded apply | 16 return at entry
#ifndef RV_ATTN_ABL
#define RV_ATTN_ABL 0
#endif
#define AT_W 31
#define AT_P 15
#define AT_TT 16
#define AT_WIN (AT_TT + 2 * AT_P)   // 46 rows
#define AT_WINP 48                  // ... padded to three MFMA tiles
#define AT_NW 5                     // waves per workgroup (5 score tiles)
#define AT_NTHR (AT_NW * 64)
#define AT_SLD 80                   // raw score row: 48 window columns + 32 rel columns
#define AT_A2LD 52                  // banded matrix row (48 + pad, 16-byte multiple)
#define AT_A3LD 84                  // [band | dense] row (80 + pad)
#define AT_MAXT 4                   // output tiles per wave (dh <= 256)

struct AttnArgs {
    const float* q; const float* k; const float* v; long ld;      // ld: row stride of q, k, v
    const float* rel;
    float* out; float* att;
    const float* dout; const float* de_in;
    float* dq; float* dk; float* dv; long dld;                    // dld: row stride of dq, dk, dv
    float* de;
    int B, L, G, dh, dhp;
    int v4, dv4;                     // 16-byte DMA legal for q/k/v rows / for dout and rel^T rows (stride F)
    int seq_kv;                      // attn_bwd_kv_k: one window buffer used twice (see the kernel)
    int rel_regs;                    // attn_fwd_k / attn_bwd_q_k (wide heads): rel^T fragments straight from global memory into registers
                                     // instead of a 32-row LDS block, so that two workgroups fit a CU
};

// dst[r][f] (r < ntotal, f < dhp) = src[(row0 + r) * ld + col0 + f] for r < nvalid, 0 <= row0 + r < L, f < dh; else 0.
// Rows travel by LDS-DMA (16 bytes per lane when `v4`: rows 16-byte aligned, else 4): all of a wave's rows are in
// flight at once and nothing passes through VGPRs; the caller waits (stage_wait) before the barrier.
// Rows outside [0, L) (the zero padding of the key / value windows) and the tail rows r >= nvalid are fetched from a buffer of zeros in
// global memory instead of being zero-filled by stores: per row the code is one scalar range test, one select of the source pointer and the
// DMA instructions -- no exec-mask branch around a store loop (the same scheme as the conv kernels' rv_zero_piece, round 4).
__device__ __attribute__((aligned(16))) const float rv_attn_zero[256] = {0.f};

__device__ __forceinline__ int stage_rows(float* dst, int ldk, const float* src, long ld, int col0, int dh, int dhp,
                                          int row0, int nvalid, int ntotal, int L, bool v4) {
    int issued = 0;                                                  // DMA instructions of this wave (wave-uniform)
    // Lean on purpose: the kernels are bound by the instructions spent here, not by bytes.  Everything per lane is
    // hoisted (byte offset inside a row, lane mask); per row there is a pointer bump, a range test and the DMA.
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // v4: the whole quads of a row travel 16 bytes per lane -- the source needs 4-byte alignment only (tools/probes/glds16_unaligned.hip), so this
    // also serves dh = 229 (57 quads + one 4-byte tail instruction instead of four 4-byte instructions per row); else 4 bytes per lane
    const int per = v4 ? 256 : 64;                                   // floats one DMA instruction moves
    const int nfull = v4 ? (dh & ~3) : dh;                           // floats covered by the main instructions
    const int ninst = (nfull + per - 1) / per;
    const int rem = dh - nfull;                                      // v4: 0..3 floats left for one 4-byte instruction
    const int lf = v4 ? lane * 4 : lane;                             // this lane's first float inside an instruction
    const float* s = src + ((long)(row0 + wave) * ld + col0) + lf;   // only dereferenced for rows inside [0, L)
    const float* z = rv_attn_zero + lf;                              // (one instruction's worth of zeros: every k reads the same piece)
    float* drow = dst + wave * ldk;
    const long sstep = (long)AT_NW * ld;
    const int dstep = AT_NW * ldk;
    const int npad = dhp - dh;
    for (int r = wave; r < ntotal; r += AT_NW, s += sstep, drow += dstep) {
        const int t = row0 + r;
        const bool real = r < nvalid && t >= 0 && t < L;
        if (v4) {
            for (int k = 0; k < ninst; ++k)
                if (k * 256 + lf < nfull) glds16(real ? s + k * 256 : z, drow + k * 256);
            if (rem) {
                if (lane < rem) glds4(real ? s - lf + nfull + lane : rv_attn_zero + lane, drow + nfull);
                ++issued;
            }
        } else {
            for (int k = 0; k < ninst; ++k)
                if (k * 64 + lf < dh) glds4(real ? s + k * 64 : z, drow + k * 64);
        }
        issued += ninst;
        if (lane < npad) drow[dh + lane] = 0.f;                      // npad < 16
    }
    return issued;
}
__device__ __forceinline__ void stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// one 16x16 score tile: S[t][y] = sum_f X[t][f] * Y[y][f]; the lane ends up with S[4*g4 + r][i], r = 0..3
__device__ __forceinline__ f32x4 score_tile(const float* X, const float* Y, int ldk, int nchunk, int i, int g4) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* xp = X + i * ldk + 4 * g4;
    const float* yp = Y + i * ldk + 4 * g4;
    for (int c = 0; c < nchunk; ++c) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(xp + 16 * c);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(yp + 16 * c);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r], b4[r], acc, 0, 0, 0);
    }
    return acc;
}

// the same with the Y operand in registers: yb[c] = Y[i][16 c + 4 g4 .. + 3]
__device__ __forceinline__ f32x4 score_tile_regs(const float* X, const f32x4 (&yb)[16], int ldk, int nchunk, int i, int g4) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* xp = X + i * ldk + 4 * g4;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (c < nchunk) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(xp + 16 * c);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r], yb[c][r], acc, 0, 0, 0);
        }
    }
    return acc;
}

// O[t][f] = sum_k A[t][k] * Y[k][f] over NCH chunks of 16 k; wave `wave` owns the output tiles wave, wave+5, ...
// Rows t0 + t of dst (row stride dld, first column col0) receive the result for f < dh.
// rel^T rows as the B operand of apply_tiles, from global memory: yk[q][c][r] = relT[16 c + 4 g4 + r][col0 + 16 nt + i] (0 past row 30 / column dh)
struct RelFrag { f32x4 y[AT_MAXT][2]; };
__device__ __forceinline__ void rel_frag_load(RelFrag& rf, const float* relT, int F, int col0, int dh, int ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int q = 0; q < AT_MAXT; ++q) {
        const int f = 16 * (wave + AT_NW * q) + i;
        const bool fok = wave + AT_NW * q < ntiles && f < dh;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int w = 16 * c + 4 * g4 + r;
                rf.y[q][c][r] = (fok && w < AT_W) ? relT[(long)w * F + col0 + f] : 0.f;
            }
    }
}

// NCH chunks of 16 k from LDS; with `rf` two more chunks (k = 16 NCH .. 16 NCH + 31 of A) whose Y rows are rel^T fragments in registers
template <int NCH>
__device__ __forceinline__ void apply_tiles(const float* A, int lda, const float* Y, int ldk, int ntiles, float* dst,
                                            long dld, int col0, int dh, int nrows_valid, const RelFrag* rf = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g4 = lane >> 4;
    f32x4 acc[AT_MAXT];
#pragma unroll
    for (int q = 0; q < AT_MAXT; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (rf) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(A + i * lda + 16 * (NCH + c) + 4 * g4);
#pragma unroll
            for (int q = 0; q < AT_MAXT; ++q) {
                if (wave + AT_NW * q < ntiles) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r], rf->y[q][c][r], acc[q], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(A + i * lda + 16 * c + 4 * g4);
        const float* yk = Y + (16 * c + 4 * g4) * ldk + i;
#pragma unroll
        for (int q = 0; q < AT_MAXT; ++q) {
            const int nt = wave + AT_NW * q;
            if (nt < ntiles) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r], yk[r * ldk + 16 * nt], acc[q], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < AT_MAXT; ++q) {
        const int nt = wave + AT_NW * q;
        const int f = 16 * nt + i;
        if (nt >= ntiles || f >= dh) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = 4 * g4 + r;
            if (t < nrows_valid) dst[(long)t * dld + col0 + f] = acc[q][r];
        }
    }
}

__device__ __forceinline__ float grp16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float grp16_max(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(AT_NTHR) void attn_fwd_k(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int dh = a.dh, dhp = a.dhp, ldk = dhp + 4, nchunk = dhp >> 4, F = a.G * dh;
    float* Qs = smem;                           // [16][ldk]
    float* Kx = Qs + AT_TT * ldk;               // [80][ldk]  key window | rel^T ; value window over rows 0..47 later  ([48][ldk] with rel_regs)
    float* Sr = Kx + (a.rel_regs ? AT_WINP : 80) * ldk;   // [16][80]
    float* A2 = Sr + AT_TT * AT_SLD;            // [16][52]
    const int ntile = (a.L + AT_TT - 1) / AT_TT;
    const int bx = xcd_remap(blockIdx.x, gridDim.x);     // an XCD (one L2) owns a run of neighbouring tiles: window overlap hits
    const int b = bx / ntile, t0 = (bx - b * ntile) * AT_TT;
    const int g = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g4 = lane >> 4;
    const long boff = (long)b * a.L * a.ld;
    if (RV_ATTN_ABL & 16) return;
    if (!(RV_ATTN_ABL & 1)) {
    stage_rows(Qs, ldk, a.q + boff, a.ld, g * dh, dh, dhp, t0, AT_TT, AT_TT, a.L, a.v4);
    stage_rows(Kx, ldk, a.k + boff, a.ld, g * dh, dh, dhp, t0 - AT_P, AT_WIN, AT_WINP, a.L, a.v4);
    }
    f32x4 yb[16];                               // rel_regs: waves 3, 4 hold their rel^T score operand (row 16 (wave - 3) + i) in registers
    if (a.rel_regs) {
        if (wave >= 3) {
            const int w = 16 * (wave - 3) + i;
            const float* rp = a.rel + (long)min(w, AT_W - 1) * F + g * dh + 4 * g4;
#pragma unroll
            for (int c = 0; c < 16; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) yb[c][r] = (c < nchunk && w < AT_W && 16 * c + 4 * g4 + r < dh) ? rp[16 * c + r] : 0.f;
        }
    } else if (!(RV_ATTN_ABL & 1)) {
        stage_rows(Kx + AT_WINP * ldk, ldk, a.rel, F, g * dh, dh, dhp, 0, AT_W, 32, AT_W, a.dv4);
    }
    for (int idx = tid; idx < AT_TT * AT_A2LD; idx += AT_NTHR) A2[idx] = 0.f;
    stage_wait();
    __syncthreads();
    if (!(RV_ATTN_ABL & 2)) {
        const f32x4 s = (a.rel_regs && wave >= 3) ? score_tile_regs(Qs, yb, ldk, nchunk, i, g4)
                                                  : score_tile(Qs, Kx + 16 * wave * ldk, ldk, nchunk, i, g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) Sr[(4 * g4 + r) * AT_SLD + 16 * wave + i] = s[r];
    }
    __syncthreads();                            // scores complete, key window dead
    if (!(RV_ATTN_ABL & 1)) stage_rows(Kx, ldk, a.v + boff, a.ld, g * dh, dh, dhp, t0 - AT_P, AT_WIN, AT_WINP, a.L, a.v4);
    if (tid < 256 && !(RV_ATTN_ABL & 4)) {      // softmax: 16 lanes per frame, two window slots per lane
        const int t = tid >> 4, l = tid & 15;
        const float* sr = Sr + t * AT_SLD;
        const float e0 = sr[t + l] + sr[AT_WINP + l];
        const float e1 = l < 15 ? sr[t + l + 16] + sr[AT_WINP + l + 16] : -INFINITY;
        const float m = grp16_max(fmaxf(e0, e1));
        float p0 = expf(e0 - m), p1 = l < 15 ? expf(e1 - m) : 0.f;
        const float inv = 1.f / grp16_sum(p0 + p1);
        p0 *= inv; p1 *= inv;
        A2[t * AT_A2LD + t + l] = p0;
        if (l < 15) A2[t * AT_A2LD + t + l + 16] = p1;
        if (a.att && t0 + t < a.L) {
            float* ar = a.att + (((long)b * a.L + t0 + t) * a.G + g) * AT_W;
            ar[l] = p0;
            if (l < 15) ar[l + 16] = p1;
        }
    }
    stage_wait();
    __syncthreads();
    if (!(RV_ATTN_ABL & 8)) apply_tiles<3>(A2, AT_A2LD, Kx, ldk, nchunk, a.out + ((long)b * a.L + t0) * F, F, g * dh, dh, a.L - t0);
}

// backward, query side: de (softmax backward) and dq for a 16-frame tile
__global__ __launch_bounds__(AT_NTHR) void attn_bwd_q_k(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int dh = a.dh, dhp = a.dhp, ldk = dhp + 4, nchunk = dhp >> 4, F = a.G * dh;
    float* Ds = smem;                           // [16][ldk]  dout tile
    float* Kx = Ds + AT_TT * ldk;               // [80][ldk]  value window, then key window | rel^T   ([48][ldk] with rel_regs)
    float* Sr = Kx + (a.rel_regs ? AT_WINP : 80) * ldk;   // [16][80]   dout . v^T (48 columns used)
    float* A3 = Sr + AT_TT * AT_SLD;            // [16][84]   [ band(de) | de ]
    const int ntile = (a.L + AT_TT - 1) / AT_TT;
    const int bx = xcd_remap(blockIdx.x, gridDim.x);     // an XCD (one L2) owns a run of neighbouring tiles: window overlap hits
    const int b = bx / ntile, t0 = (bx - b * ntile) * AT_TT;
    const int g = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g4 = lane >> 4;
    const long boff = (long)b * a.L * a.ld;
    stage_rows(Ds, ldk, a.dout + (long)b * a.L * F, F, g * dh, dh, dhp, t0, AT_TT, AT_TT, a.L, a.dv4);
    stage_rows(Kx, ldk, a.v + boff, a.ld, g * dh, dh, dhp, t0 - AT_P, AT_WIN, AT_WINP, a.L, a.v4);
    RelFrag rf;
    if (a.rel_regs) rel_frag_load(rf, a.rel, F, g * dh, dh, nchunk);      // (consumed by the last phase: the loads fly under everything else)
    else stage_rows(Kx + AT_WINP * ldk, ldk, a.rel, F, g * dh, dh, dhp, 0, AT_W, 32, AT_W, a.dv4);
    for (int idx = tid; idx < AT_TT * AT_A3LD; idx += AT_NTHR) A3[idx] = 0.f;
    stage_wait();
    __syncthreads();
    if (wave < 3) {
        const f32x4 s = score_tile(Ds, Kx + 16 * wave * ldk, ldk, nchunk, i, g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) Sr[(4 * g4 + r) * AT_SLD + 16 * wave + i] = s[r];
    }
    __syncthreads();                            // value window dead
    stage_rows(Kx, ldk, a.k + boff, a.ld, g * dh, dh, dhp, t0 - AT_P, AT_WIN, AT_WINP, a.L, a.v4);
    if (tid < 256) {
        const int t = tid >> 4, l = tid & 15;
        const bool rv = t0 + t < a.L;
        const long arow = (((long)b * a.L + t0 + t) * a.G + g) * AT_W;
        const float p0 = rv ? a.att[arow + l] : 0.f;
        const float p1 = (rv && l < 15) ? a.att[arow + l + 16] : 0.f;
        const float* sr = Sr + t * AT_SLD;
        const float d0 = sr[t + l], d1 = l < 15 ? sr[t + l + 16] : 0.f;
        const float dot = grp16_sum(p0 * d0 + p1 * d1);
        const float e0 = p0 * (d0 - dot), e1 = p1 * (d1 - dot);
        float* ar = A3 + t * AT_A3LD;
        ar[t + l] = e0;
        ar[AT_WINP + l] = e0;
        if (l < 15) { ar[t + l + 16] = e1; ar[AT_WINP + l + 16] = e1; }
        if (rv) {
            a.de[arow + l] = e0;
            if (l < 15) a.de[arow + l + 16] = e1;
        }
    }
    stage_wait();
    __syncthreads();
    if (a.rel_regs) apply_tiles<3>(A3, AT_A3LD, Kx, ldk, nchunk, a.dq + ((long)b * a.L + t0) * a.dld, a.dld, g * dh, dh, a.L - t0, &rf);
    else apply_tiles<5>(A3, AT_A3LD, Kx, ldk, nchunk, a.dq + ((long)b * a.L + t0) * a.dld, a.dld, g * dh, dh, a.L - t0);
}

// backward, key/value side for a 16-frame tile of window rows s:
//   dv[s] = sum_w att[s-w+15][w] * dout[s-w+15],   dk[s] = sum_w de[s-w+15][w] * q[s-w+15]
// i.e. the transposed band A4[s][u] = coef[s0-15+u][s - u + 30] (u = 0..45) against the 46-row window of dout / q.
__global__ __launch_bounds__(AT_NTHR) void attn_bwd_kv_k(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int dh = a.dh, dhp = a.dhp, ldk = dhp + 4, nchunk = dhp >> 4, F = a.G * dh;
    // a.seq_kv (wide heads): ONE row-window buffer, used for the dout window (dv) and then for the q window (dk) -- 66 KB
    // instead of 113 KB of LDS at dh = 229, so that TWO workgroups fit a CU and one's DMA wait hides under the other's MFMAs
    // (a 5-wave workgroup alone leaves the CU idle through every staging phase: 82 -> 58 us).  Narrow heads (dh = 128) fit
    // twice either way and keep both windows in flight at once.
    float* Yd = smem;                           // [48][ldk]  dout window (then q window)
    float* Yq = a.seq_kv ? Yd : Yd + AT_WINP * ldk;   // [48][ldk]  q window
    float* Ca = Yq + AT_WINP * ldk;             // [48][32]   att window
    float* Ce = Ca + AT_WINP * 32;              // [48][32]   de window
    float* A4 = Ce + AT_WINP * 32;              // [16][52]   transposed band of att
    float* A5 = A4 + AT_TT * AT_A2LD;           // [16][52]   transposed band of de
    const int ntile = (a.L + AT_TT - 1) / AT_TT;
    const int bx = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bx / ntile, s0 = (bx - b * ntile) * AT_TT;
    const int g = blockIdx.y;
    const int tid = threadIdx.x;
    // everything this tile needs is in flight at once: both row windows and both coefficient windows (a row of the
    // latter is 31 contiguous floats)
    stage_rows(Yd, ldk, a.dout + (long)b * a.L * F, F, g * dh, dh, dhp, s0 - AT_P, AT_WIN, AT_WINP, a.L, a.dv4);
    if (!a.seq_kv) stage_rows(Yq, ldk, a.q + (long)b * a.L * a.ld, a.ld, g * dh, dh, dhp, s0 - AT_P, AT_WIN, AT_WINP, a.L, a.v4);
    const long cld = (long)a.G * AT_W;
    stage_rows(Ca, 32, a.att + (long)b * a.L * cld, cld, g * AT_W, AT_W, 32, s0 - AT_P, AT_WIN, AT_WINP, a.L, false);
    stage_rows(Ce, 32, a.de_in + (long)b * a.L * cld, cld, g * AT_W, AT_W, 32, s0 - AT_P, AT_WIN, AT_WINP, a.L, false);
    stage_wait();
    __syncthreads();
    for (int idx = tid; idx < AT_TT * AT_A2LD; idx += AT_NTHR) {
        const int sl = idx / AT_A2LD, u = idx - sl * AT_A2LD;
        const int w = sl - u + 2 * AT_P;        // t = s - w + 15  <=>  w = s - t + 15 = sl - u + 30
        const bool in = u < AT_WIN && w >= 0 && w < AT_W;
        A4[idx] = in ? Ca[u * 32 + w] : 0.f;
        A5[idx] = in ? Ce[u * 32 + w] : 0.f;
    }
    __syncthreads();
    const long orow = ((long)b * a.L + s0) * a.dld;
    apply_tiles<3>(A4, AT_A2LD, Yd, ldk, nchunk, a.dv + orow, a.dld, g * dh, dh, a.L - s0);
    if (a.seq_kv) {
        __syncthreads();                        // dout window dead
        stage_rows(Yq, ldk, a.q + (long)b * a.L * a.ld, a.ld, g * dh, dh, dhp, s0 - AT_P, AT_WIN, AT_WINP, a.L, a.v4);
        stage_wait();
        __syncthreads();
    }
    apply_tiles<3>(A5, AT_A2LD, Yq, ldk, nchunk, a.dk + orow, a.dld, g * dh, dh, a.L - s0);
}

// 16-byte DMA lanes are legal for every fp32 row (the source needs 4-byte alignment only; a row tail of dh % 4 floats goes 4 bytes per lane).
// RV_ATTN_V4_ALIGNED=1 restores the round-3 rule (16-byte aligned rows of a multiple of four floats) for A/B runs.
static bool attn_v4(int dh, long ld, const void* p) {
    static const int aligned_only = getenv("RV_ATTN_V4_ALIGNED") ? atoi(getenv("RV_ATTN_V4_ALIGNED")) : 0;
    if (!aligned_only) return true;
    return (dh % 4) == 0 && (ld % 4) == 0 && (((uintptr_t)p) & 15) == 0;
}

static int attn_setup(AttnArgs& a, int dh, const char* who) {
    RV_CHECK_ARG(dh >= 1 && dh <= 256, "%s: head dim %d unsupported", who, dh);
    a.dh = dh;
    a.dhp = (dh + 15) & ~15;
    static std::once_flag attr_once;              // (forward and backward launches come from different threads)
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute((const void*)attn_fwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_bwd_q_k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_bwd_kv_k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    });
    return RV_OK;
}

extern "C" {

int rv_local_attn_fwd(const float* q, const float* k, const float* v, long ld, const float* relT, float* out, float* att, int B,
                      int L, int G, int dh, void* stream) {
    AttnArgs a = {};
    const int rc = attn_setup(a, dh, "rv_local_attn_fwd");
    if (rc != RV_OK) return rc;
    RV_CHECK_ARG(ld >= (long)G * dh, "rv_local_attn_fwd: row stride %ld < G*dh", ld);
    a.q = q; a.k = k; a.v = v; a.ld = ld; a.rel = relT; a.out = out; a.att = att; a.B = B; a.L = L; a.G = G;
    a.v4 = attn_v4(dh, ld, q) && attn_v4(dh, ld, k) && attn_v4(dh, ld, v);
    a.dv4 = attn_v4(dh, (long)G * dh, relT);
    const int ntile = (L + AT_TT - 1) / AT_TT, ldk = a.dhp + 4;
    size_t lds = ((size_t)(AT_TT + 80) * ldk + AT_TT * AT_SLD + AT_TT * AT_A2LD) * sizeof(float);
    static const int relregs_env = getenv("RV_ATTN_REL_REGS") ? atoi(getenv("RV_ATTN_REL_REGS")) : 1;
    static const size_t occ_kb = getenv("RV_ATTN_OCC_KB") ? (size_t)atoi(getenv("RV_ATTN_OCC_KB")) : 78;      // (experiment: 50 -> three workgroups per CU at dh = 128)
    a.rel_regs = relregs_env && lds > occ_kb * 1024;       // the rel^T block would leave room for only one workgroup per CU
    if (a.rel_regs) lds -= (size_t)32 * ldk * sizeof(float);
    hipLaunchKernelGGL(attn_fwd_k, dim3(B * ntile, G), dim3(AT_NTHR), lds, (hipStream_t)stream, a);
    RV_LAUNCH_CHECK("rv_local_attn_fwd");
    return RV_OK;
}

// Inputs: dout and the forward's q, k, v, att.  Outputs: dq, dk, dv (row stride dld) and de [B,L,G,31] (the energy
// gradient; drel = sum_{b,t} de (x) q is a plain GEMM the host issues through rv_gemm).
int rv_local_attn_bwd(const float* dout, const float* q, const float* k, const float* v, long ld, const float* relT,
                      const float* att, float* dq, float* dk, float* dv, long dld, float* de, int B, int L, int G, int dh,
                      void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AttnArgs a = {};
    const int rc = attn_setup(a, dh, "rv_local_attn_bwd");
    if (rc != RV_OK) return rc;
    RV_CHECK_ARG(ld >= (long)G * dh && dld >= (long)G * dh, "rv_local_attn_bwd: row stride < G*dh");
    a.q = q; a.k = k; a.v = v; a.ld = ld; a.rel = relT; a.att = (float*)att; a.dout = dout;
    a.dq = dq; a.dk = dk; a.dv = dv; a.dld = dld; a.de = de; a.de_in = de; a.B = B; a.L = L; a.G = G;
    a.v4 = attn_v4(dh, ld, q) && attn_v4(dh, ld, k) && attn_v4(dh, ld, v);
    a.dv4 = attn_v4(dh, (long)G * dh, relT) && attn_v4(dh, (long)G * dh, dout);
    const int ntile = (L + AT_TT - 1) / AT_TT, ldk = a.dhp + 4;
    size_t lds1 = ((size_t)(AT_TT + 80) * ldk + AT_TT * AT_SLD + AT_TT * AT_A3LD) * sizeof(float);
    static const int relregs_env = getenv("RV_ATTN_REL_REGS") ? atoi(getenv("RV_ATTN_REL_REGS")) : 1;
    static const size_t occ_kb = getenv("RV_ATTN_OCC_KB") ? (size_t)atoi(getenv("RV_ATTN_OCC_KB")) : 78;
    a.rel_regs = relregs_env && lds1 > occ_kb * 1024;
    if (a.rel_regs) lds1 -= (size_t)32 * ldk * sizeof(float);
    hipLaunchKernelGGL(attn_bwd_q_k, dim3(B * ntile, G), dim3(AT_NTHR), lds1, st, a);
    RV_LAUNCH_CHECK("rv_local_attn_bwd(q)");
    const size_t lds_two = ((size_t)2 * AT_WINP * ldk + 2 * AT_WINP * 32 + 2 * AT_TT * AT_A2LD) * sizeof(float);
    a.seq_kv = lds_two > occ_kb * 1024;            // two windows at once would leave room for only one workgroup per CU
    const size_t lds2 = lds_two - (a.seq_kv ? (size_t)AT_WINP * ldk * sizeof(float) : 0);
    hipLaunchKernelGGL(attn_bwd_kv_k, dim3(B * ntile, G), dim3(AT_NTHR), lds2, st, a);
    RV_LAUNCH_CHECK("rv_local_attn_bwd(kv)");
    return RV_OK;
}

}  // extern "C"
