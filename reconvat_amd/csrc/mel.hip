// Fused log-Mel front-end for gfx950: framing + reflect pad + Hann + 2048-point FFT in LDS + |.|^2 +
// sparse Slaney mel filterbank + log(. + 1e-5), written time-major [B, T, 229], then a per-clip
// min-max normalisation pass.
//
// Reference: nnAudio MelSpectrogram as STFT-by-conv1d (model/Spectrogram.py:187-231, :443-461: two
// 1025x2048 conv1d kernels = 5.4 GFLOP/segment, then a 99%-zero 229x1025 matmul), log and
// Normalization('imagewise') (model/UNet_onset.py:419-423, model/utils.py:94-100).  Here the DFT is a
// radix-2 FFT (0.03 GFLOP/segment) and the mel product touches only the 2025 non-zeros, so the stage
// is bound by reading the audio once (1.3 MB/segment) and writing 0.59 MB/segment.
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

__global__ __launch_bounds__(256) void mel_frame_k(MelArgs a) {
    __shared__ float re[NFFT];
    __shared__ float im[NFFT];
    __shared__ float red[2][4];
    const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    const float* x = a.audio + (long)b * a.audio_stride;
    const int start = t * a.hop - NFFT / 2;
    // windowed frame, stored bit-reversed for the in-place DIT FFT
    for (int n = tid; n < NFFT; n += 256) {
        int i = start + n;
        if (i < 0) i = -i;
        if (i >= a.nsamp) i = 2 * (a.nsamp - 1) - i;
        float v = x[i] * a.window[n];
        unsigned r = __brev((unsigned)n) >> (32 - LOGN);
        re[r] = v;
        im[r] = 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int s = 1; s <= LOGN; ++s) {
        const int half = 1 << (s - 1);
        const int tstep = NFFT >> s;           // twiddle index stride: W_2048^(pos * 2048/m)
        for (int jj = tid; jj < NFFT / 2; jj += 256) {
            int grp = jj >> (s - 1), pos = jj & (half - 1);
            int i0 = (grp << s) + pos, i1 = i0 + half;
            float wr = a.twiddle[2 * (pos * tstep)], wi = a.twiddle[2 * (pos * tstep) + 1];
            float xr = re[i1], xi = im[i1];
            float tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
            float ur = re[i0], ui = im[i0];
            re[i0] = ur + tr; im[i0] = ui + ti;
            re[i1] = ur - tr; im[i1] = ui - ti;
        }
        __syncthreads();
    }
    // power spectrum in place (bins 0..1024); the reference takes sqrt then squares again
    for (int k = tid; k < NFREQ; k += 256) {
        float m = sqrtf(re[k] * re[k] + im[k] * im[k]);
        re[k] = m * m;
    }
    __syncthreads();
    float mn = INFINITY, mx = -INFINITY;
    for (int m = tid; m < a.n_mels; m += 256) {
        const int s0 = a.mel_start[m], len = a.mel_len[m];
        const float* w = a.mel_w + (long)m * a.mel_ld;
        float acc = 0.f;
        for (int i = 0; i < len; ++i) acc = fmaf(w[i], re[s0 + i], acc);
        float v = a.do_log ? logf(acc + 1e-5f) : acc;
        a.out[((long)b * a.T + t) * a.n_mels + m] = v;
        mn = fminf(mn, v); mx = fmaxf(mx, v);
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
    MelArgs a;
    a.audio = audio; a.audio_stride = audio_stride; a.nsamp = nsamp; a.window = window; a.twiddle = twiddle;
    a.mel_start = mel_start; a.mel_len = mel_len; a.mel_w = mel_w; a.mel_ld = mel_ld; a.n_mels = n_mels;
    a.out = out; a.minmax = (unsigned*)workspace; a.T = T; a.hop = hop; a.do_log = do_log;
    hipLaunchKernelGGL(mel_init_minmax_k, dim3(cdiv(B, 64)), dim3(64), 0, st, a.minmax, B);
    hipLaunchKernelGGL(mel_frame_k, dim3(T, B), dim3(256), 0, st, a);
    RV_LAUNCH_CHECK("rv_melspec_lognorm_fwd(frames)");
    if (normalise) {
        long per = (long)T * n_mels;
        hipLaunchKernelGGL(mel_normalise_k, dim3((int)((per + 1023) / 1024), B), dim3(256), 0, st, out, a.minmax, per, B);
        RV_LAUNCH_CHECK("rv_melspec_lognorm_fwd(normalise)");
    }
    return RV_OK;
}

}  // extern "C"
