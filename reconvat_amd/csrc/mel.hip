// Fused log-Mel front-end for gfx950: framing + reflect pad + Hann + 2048-point FFT in LDS + |.|^2 +
// sparse Slaney mel filterbank + log(. + 1e-5), written time-major [B, T, 229], then a per-clip
// min-max normalisation pass.
//
// Reference: nnAudio MelSpectrogram as STFT-by-conv1d (model/Spectrogram.py:187-231, :443-461: two
// 1025x2048 conv1d kernels = 5.4 GFLOP/segment, then a 99%-zero 229x1025 matmul), log and
// Normalization('imagewise') (model/UNet_onset.py:419-423, model/utils.py:94-100).  Here the DFT is a
// real-input radix-4 FFT (1024 complex points per frame, 0.015 GFLOP/segment) and the mel product touches only the 2025
// non-zeros, so the stage is bound by reading the audio once (1.3 MB/segment) and writing 0.59 MB/segment.
#include "common.h"

#define NFFT 2048
#define LOGN 11
#define NFREQ 1025

struct MelArgs {
    const float* audio; long audio_stride; int nsamp;   // [B, nsamp]
    const float* window;     // [2048]
    const float* twiddle;    // [1024][2] = cos, -sin of 2*pi*k/2048
    const int* mel_start;    // [n_mels]
    const int* mel_len;      // [n_mels]
    const float* mel_w;      // [n_mels][mel_ld]
    int mel_ld, n_mels;
    float* out;              // [B, T, n_mels]
    unsigned* minmax;        // [B][2] order-preserving uint encodings (min, max)
    int T, hop, do_log;
};

__device__ __forceinline__ unsigned f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// One workgroup = FPW consecutive frames of one clip.  The frames overlap by 3/4 (hop 512, window 2048), so their
// 2048 + (FPW-1)*512 samples are fetched from global ONCE into LDS (reflect padding applied there), together with the Hann
// window and the 1024 roots of unity.  Per frame the 2048 real samples are packed as 1024 complex points z[n] = x[2n] + i x[2n+1],
// transformed by a radix-4 decimation-in-time FFT (5 stages x 256 butterflies, one per thread; input stored in base-4
// digit-reversed order) and split into the 1025 bins of the real transform:
//     X[k] = E[k] + W_2048^k O[k],  E = (Z[k] + conj Z[1024-k]) / 2,  O = (Z[k] - conj Z[1024-k]) / (2i).
// That is half the butterflies of a complex 2048-point transform in half the stages of a radix-2 one (the previous kernel:
// 11 radix-2 stages over 2048 complex points, twiddles re-read from global in every stage, one frame per workgroup so that
// every sample was fetched four times).
#define FPW 4
#define NC (NFFT / 2)            // complex points
// LDS index swizzle of the 1024-point work array: in stages 0..2 the butterflies of a 32-lane pass touch points whose
// low five index bits do not cover all 32 values (strides 4, 16, 64), i.e. 8-, 8- and 4-way bank conflicts on every access.
// XOR-ing index bits 5 and 6 into the low bits (bit 5 -> 0b00101, bit 6 -> 0b11010) makes the 32 addresses of a pass distinct
// modulo 32 in every stage (it is a bijection on the low five bits for each of the five access patterns).
__device__ __forceinline__ int zsw(int i) { return i ^ (((i >> 5) & 1) * 5) ^ (((i >> 6) & 1) * 26); }

__device__ __forceinline__ unsigned digitrev4_10(unsigned n) {
    unsigned r = __brev(n) >> 22;                     // 10-bit bit reversal ...
    return ((r & 0x155u) << 1) | ((r >> 1) & 0x155u); // ... with the two bits of every base-4 digit swapped back
}

// Everything that is the same for every frame lives in REGISTERS of the thread that uses it -- its eight window values, the
// twelve twiddles of its butterflies (stage s > 0: W_{4L}^{m q}, q = tid mod L), the four split twiddles of its bins and the
// (<= MELW) filter taps of its mel band -- so LDS holds only the audio chunk, two 1024-point work arrays and two power
// spectra (39 KB: four workgroups per CU) and the inner loops read no tables.
#define MELW 32
__global__ __launch_bounds__(256) void mel_frame_k(MelArgs a) {
    constexpr int CHUNK = NFFT + (FPW - 1) * 512;
    __shared__ __attribute__((aligned(16))) float audio[CHUNK];
    __shared__ f32x2 z[2][NC];
    __shared__ float pw[2][NFREQ + MELW + 3];
    __shared__ float red[2][4];
    const int b = blockIdx.y, t0 = blockIdx.x * FPW, tid = threadIdx.x;
    const float* x = a.audio + (long)b * a.audio_stride;
    const int start = t0 * a.hop - NFFT / 2;
    const int nframes = min(FPW, a.T - t0);
    const int need = NFFT + (nframes - 1) * a.hop;
    for (int n = tid; n < need; n += 256) {
        int i = start + n;
        if (i < 0) i = -i;
        if (i >= a.nsamp) i = 2 * (a.nsamp - 1) - i;
        audio[n] = x[i];
    }
    auto root = [&](int idx) -> f32x2 {                // W_2048^idx, idx < 2048 (the table holds half a turn)
        const int k = idx & (NC - 1);
        const float sg = (idx & NC) ? -1.f : 1.f;
        return (f32x2){sg * a.twiddle[2 * k], sg * a.twiddle[2 * k + 1]};
    };
    f32x2 wn[4];                                       // window of points 2n, 2n+1 for n = tid + 256 j
#pragma unroll
    for (int j = 0; j < 4; ++j) wn[j] = *reinterpret_cast<const f32x2*>(a.window + 2 * (tid + 256 * j));
    f32x2 wst[4][3];                                   // butterfly twiddles of stages 1..4
#pragma unroll
    for (int s = 1; s < 5; ++s) {
        const int L = 1 << (2 * s), q = tid & (L - 1);
#pragma unroll
        for (int m = 1; m <= 3; ++m) wst[s - 1][m - 1] = root(m * q * (512 >> (2 * s)));
    }
    f32x2 wsp[4];                                      // split twiddles W_2048^k, k = tid + 256 j
#pragma unroll
    for (int j = 0; j < 4; ++j) wsp[j] = root(tid + 256 * j);
    float mw[MELW];                                    // this thread's mel band (tid < n_mels)
    int ms0 = 0, mlen = 0;
    // the filter rows are zero-padded to the row stride (== MELW, checked by the host): load all of it, unconditionally --
    // a per-tap "load or zero" select makes the compiler branch around every load and wait for each one in turn
    if (tid < a.n_mels) ms0 = a.mel_start[tid];
    const float* wrow = a.mel_w + (long)min(tid, a.n_mels - 1) * MELW;
#pragma unroll
    for (int i = 0; i < MELW; i += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + i);
        mw[i] = v[0]; mw[i + 1] = v[1]; mw[i + 2] = v[2]; mw[i + 3] = v[3];
    }
    (void)mlen;
    float mn = INFINITY, mx = -INFINITY;
    for (int k = tid; k < MELW; k += 256) { pw[0][NFREQ + k] = 0.f; pw[1][NFREQ + k] = 0.f; }   // taps past a band's end read zeros
    // TWO frames per pass: their butterflies are independent, so every barrier interval carries twice the LDS / VALU work
    // of one frame (the kernel is latency-bound: ~10 dependent LDS operations between barriers)
    for (int f = 0; f < nframes; f += 2) {
        __syncthreads();                               // audio chunk / previous pass's readers
        const int fB = min(f + 1, nframes - 1);        // odd tail: the second slot recomputes the last frame, its store is skipped
        const float* fr[2] = {audio + f * a.hop, audio + fB * a.hop};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = tid + 256 * j;
            const int dst = zsw((int)digitrev4_10((unsigned)n));
#pragma unroll
            for (int h = 0; h < 2; ++h) z[h][dst] = *reinterpret_cast<const f32x2*>(fr[h] + 2 * n) * wn[j];
        }
        __syncthreads();
        // radix-4 DIT stages: sub-transforms of length L -> 4L; butterfly tid: position q in the sub-transform, group grp
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int L = 1 << (2 * s);
            const int q = tid & (L - 1), grp = tid >> (2 * s);
            const int base = grp * 4 * L + q;
            const int i0 = zsw(base), i1 = zsw(base + L), i2 = zsw(base + 2 * L), i3 = zsw(base + 3 * L);
            f32x2 a0[2], a1[2], a2[2], a3[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) { a0[h] = z[h][i0]; a1[h] = z[h][i1]; a2[h] = z[h][i2]; a3[h] = z[h][i3]; }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (s > 0) {
                    const f32x2 w1 = wst[s - 1][0], w2 = wst[s - 1][1], w3 = wst[s - 1][2];
                    a1[h] = (f32x2){a1[h][0] * w1[0] - a1[h][1] * w1[1], a1[h][0] * w1[1] + a1[h][1] * w1[0]};
                    a2[h] = (f32x2){a2[h][0] * w2[0] - a2[h][1] * w2[1], a2[h][0] * w2[1] + a2[h][1] * w2[0]};
                    a3[h] = (f32x2){a3[h][0] * w3[0] - a3[h][1] * w3[1], a3[h][0] * w3[1] + a3[h][1] * w3[0]};
                }
                const f32x2 u0 = a0[h] + a2[h], u1 = a0[h] - a2[h], u2 = a1[h] + a3[h];
                const f32x2 d = a1[h] - a3[h];
                const f32x2 u3 = (f32x2){d[1], -d[0]};  // -i (a1 - a3)
                z[h][i0] = u0 + u2;                    // in place: these four points belong to this butterfly alone
                z[h][i1] = u1 + u3;
                z[h][i2] = u0 - u2;
                z[h][i3] = u1 - u3;
            }
            __syncthreads();
        }
        // real-transform split + power spectrum (the reference takes sqrt then squares again)
#pragma unroll
        for (int j = 0; j <= 4; ++j) {
            const int k = tid + 256 * j;
            if (j == 4 && tid != 0) break;             // bin 1024: one thread
            const int ik = zsw(k & (NC - 1)), ic = zsw((NC - k) & (NC - 1));
            const f32x2 w = j < 4 ? wsp[j & 3] : (f32x2){-1.f, 0.f};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2 zk = z[h][ik], zc = z[h][ic];
                const float er = 0.5f * (zk[0] + zc[0]), ei = 0.5f * (zk[1] - zc[1]);     // E = (Zk + conj Zc) / 2
                const float dr = zk[0] - zc[0], di = zk[1] + zc[1];                      // D = Zk - conj Zc
                const float orr = 0.5f * di, oi = -0.5f * dr;                            // O = D / (2i)
                const float xr = er + w[0] * orr - w[1] * oi, xi = ei + w[0] * oi + w[1] * orr;
                const float m = sqrtf(xr * xr + xi * xi);
                pw[h][k] = m * m;
            }
        }
        __syncthreads();
        if (tid < a.n_mels) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < MELW; ++i) acc = fmaf(mw[i], pw[h][ms0 + i], acc);
                const float v = a.do_log ? logf(acc + 1e-5f) : acc;
                if (h == 0 || f + 1 < nframes) {
                    a.out[((long)b * a.T + t0 + f + h) * a.n_mels + tid] = v;
                    mn = fminf(mn, v); mx = fmaxf(mx, v);
                }
            }
        }
    }
    mn = wave_min(mn); mx = wave_max(mx);
    if ((tid & 63) == 0) { red[0][tid >> 6] = mn; red[1][tid >> 6] = mx; }
    __syncthreads();
    if (tid == 0) {
        mn = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
        mx = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
        atomicMin(&a.minmax[2 * b], f2ord(mn));
        atomicMax(&a.minmax[2 * b + 1], f2ord(mx));
    }
}

__global__ void mel_init_minmax_k(unsigned* mm, int B) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) { mm[2 * i] = 0xffffffffu; mm[2 * i + 1] = 0u; }
}

__global__ __launch_bounds__(256) void mel_normalise_k(float* out, const unsigned* mm, long per_clip, int B) {
    const int b = blockIdx.y;
    const float mn = ord2f(mm[2 * b]), mx = ord2f(mm[2 * b + 1]);
    const float den = mx - mn;             // no epsilon, as model/utils.py:100
    float* o = out + (long)b * per_clip;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_clip; i += (long)gridDim.x * blockDim.x)
        o[i] = (o[i] - mn) / den;
}

extern "C" {

// audio [B, nsamp] (row stride audio_stride floats) -> out [B, T, n_mels] with T = 1 + nsamp/hop frames
// (center=True, reflect pad 1024).  do_log: log(mel + 1e-5); normalise: per-clip min-max ("imagewise").
// workspace: 2*B uint32.
int rv_melspec_lognorm_fwd(const float* audio, long audio_stride, int B, int nsamp, const float* window,
                           const float* twiddle, const int* mel_start, const int* mel_len, const float* mel_w, int mel_ld,
                           int n_mels, int hop, int do_log, int normalise, float* out, int T, void* workspace, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    RV_CHECK_ARG(nsamp > NFFT / 2, "rv_melspec_lognorm_fwd: signal shorter than the reflect padding");
    RV_CHECK_ARG(T == 1 + nsamp / hop, "rv_melspec_lognorm_fwd: T=%d but 1 + nsamp/hop = %d", T, 1 + nsamp / hop);
    RV_CHECK_ARG(mel_ld == MELW && ((((uintptr_t)mel_w) & 15) == 0),
                 "rv_melspec_lognorm_fwd: mel filter rows must be zero-padded to exactly %d taps, 16-byte aligned (row stride %d)", MELW, mel_ld);
    RV_CHECK_ARG(n_mels <= 256 && mel_ld >= 1, "rv_melspec_lognorm_fwd: at most 256 mel bands (one thread each), got %d", n_mels);
    RV_CHECK_ARG(hop == 512, "rv_melspec_lognorm_fwd: the frame-sharing kernel is built for hop 512 (got %d)", hop);
    MelArgs a;
    a.audio = audio; a.audio_stride = audio_stride; a.nsamp = nsamp; a.window = window; a.twiddle = twiddle;
    a.mel_start = mel_start; a.mel_len = mel_len; a.mel_w = mel_w; a.mel_ld = mel_ld; a.n_mels = n_mels;
    a.out = out; a.minmax = (unsigned*)workspace; a.T = T; a.hop = hop; a.do_log = do_log;
    hipLaunchKernelGGL(mel_init_minmax_k, dim3(cdiv(B, 64)), dim3(64), 0, st, a.minmax, B);
    hipLaunchKernelGGL(mel_frame_k, dim3(cdiv(T, FPW), B), dim3(256), 0, st, a);
    RV_LAUNCH_CHECK("rv_melspec_lognorm_fwd(frames)");
    if (normalise) {
        long per = (long)T * n_mels;
        hipLaunchKernelGGL(mel_normalise_k, dim3((int)((per + 1023) / 1024), B), dim3(256), 0, st, out, a.minmax, per, B);
        RV_LAUNCH_CHECK("rv_melspec_lognorm_fwd(normalise)");
    }
    return RV_OK;
}

}  // extern "C"
