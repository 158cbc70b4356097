// Device-side segment cropper for the data feed: the reference's item rule
// PianoRollAudioDataset.__getitem__ (model/dataset.py:35-69) applied to a whole batch on the GPU.
//
// The corpus lives in HBM as the reference keeps it on the host: int16 audio and uint8 label / velocity rolls of all
// tracks, concatenated.  The host only draws the crop positions (the reference's RandomState rule) and hands over
// two small index arrays; one launch per tensor family then produces the float batch the training step consumes:
//   audio[b][i]      = float(corpus_audio[audio_begin[b] + i]) / 32768                       (:62, exact in fp32)
//   onset/offset/frame[b][s][k] = label == 3 / == 1 / > 1                                    (:63-65)
//   velocity[b][s][k] = float(vel) / 128                                                     (:66)
// Pure byte/integer streaming (HBM-bound, 2 B -> 4 B and 2 B -> 16 B expansion), bit-exact by construction;
// 16-byte loads and stores whenever the crop start is 16-byte aligned (track starts are padded by the host).
#include "common.h"

typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef unsigned char u8x16 __attribute__((ext_vector_type(16)));

struct CropArgs {
    const short* audio; const unsigned char* label; const unsigned char* velocity;
    const long* audio_begin; const long* label_begin;      // [B] element offsets into the corpus buffers
    long seq_len, nlab;                                     // samples per item, label bytes per item (n_steps * n_keys)
    float* out_audio; float* onset; float* offset; float* frame; float* out_velocity;
};

__global__ __launch_bounds__(256) void crop_audio_k(CropArgs a) {
    const int b = blockIdx.y;
    const short* src = a.audio + a.audio_begin[b];
    float* dst = a.out_audio + (long)b * a.seq_len;
    const float k = 1.0f / 32768.0f;                        // power of two: x * k == x / 32768 exactly
    const bool vec = ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
    const long stride = (long)gridDim.x * blockDim.x;
    if (vec) {
        const long n8 = a.seq_len >> 3;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
            const i16x8 v = *reinterpret_cast<const i16x8*>(src + i * 8);
            f32x4 lo = (f32x4){(float)v[0] * k, (float)v[1] * k, (float)v[2] * k, (float)v[3] * k};
            f32x4 hi = (f32x4){(float)v[4] * k, (float)v[5] * k, (float)v[6] * k, (float)v[7] * k};
            *reinterpret_cast<f32x4*>(dst + i * 8) = lo;
            *reinterpret_cast<f32x4*>(dst + i * 8 + 4) = hi;
        }
        for (long i = (n8 << 3) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.seq_len; i += stride) dst[i] = (float)src[i] * k;
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.seq_len; i += stride) dst[i] = (float)src[i] * k;
    }
}

__global__ __launch_bounds__(256) void crop_label_k(CropArgs a) {
    const int b = blockIdx.y;
    const unsigned char* lab = a.label + a.label_begin[b];
    const unsigned char* vel = a.velocity ? a.velocity + a.label_begin[b] : nullptr;
    const long o = (long)b * a.nlab;
    const float kv = 1.0f / 128.0f;
    const long stride = (long)gridDim.x * blockDim.x;
    // four labels per thread when everything is 4-byte (inputs) / 16-byte (outputs) aligned: one dword load, float4 stores
    const bool vec = ((((uintptr_t)lab) | (vel ? (uintptr_t)vel : 0)) & 3) == 0 && ((o | a.nlab) & 3) == 0 &&
                     (((uintptr_t)a.onset | (uintptr_t)a.frame | (uintptr_t)a.offset | (uintptr_t)a.out_velocity) & 15) == 0;
    long done = 0;
    if (vec) {
        const long n4 = a.nlab >> 2;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            const unsigned l4 = *reinterpret_cast<const unsigned*>(lab + i * 4);
            f32x4 on, of, fr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned l = (l4 >> (8 * q)) & 255u;
                on[q] = l == 3 ? 1.f : 0.f;
                of[q] = l == 1 ? 1.f : 0.f;
                fr[q] = l > 1 ? 1.f : 0.f;
            }
            *reinterpret_cast<f32x4*>(a.onset + o + i * 4) = on;
            if (a.offset) *reinterpret_cast<f32x4*>(a.offset + o + i * 4) = of;
            *reinterpret_cast<f32x4*>(a.frame + o + i * 4) = fr;
            if (vel && a.out_velocity) {
                const unsigned v4 = *reinterpret_cast<const unsigned*>(vel + i * 4);
                *reinterpret_cast<f32x4*>(a.out_velocity + o + i * 4) =
                    (f32x4){(float)(v4 & 255u) * kv, (float)((v4 >> 8) & 255u) * kv, (float)((v4 >> 16) & 255u) * kv,
                            (float)(v4 >> 24) * kv};
            }
        }
        done = n4 << 2;
    }
    for (long i = done + (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.nlab; i += stride) {
        const unsigned char l = lab[i];
        a.onset[o + i] = l == 3 ? 1.f : 0.f;
        if (a.offset) a.offset[o + i] = l == 1 ? 1.f : 0.f;
        a.frame[o + i] = l > 1 ? 1.f : 0.f;
        if (vel && a.out_velocity) a.out_velocity[o + i] = (float)vel[i] * kv;
    }
}

extern "C" {

// audio: int16 corpus, label / velocity: uint8 corpora (velocity, offset, out_velocity nullable); audio_begin /
// label_begin: DEVICE arrays of B element offsets (label_begin in bytes = step_begin * n_keys + track offset).
// out_audio [B, seq_len]; onset / offset / frame / out_velocity [B, n_steps, n_keys] float32.
int rv_crop_segments(const short* audio, const unsigned char* label, const unsigned char* velocity, const long* audio_begin,
                     const long* label_begin, int B, long seq_len, int n_steps, int n_keys, float* out_audio, float* onset,
                     float* offset, float* frame, float* out_velocity, void* stream) {
    RV_CHECK_ARG(B >= 1 && seq_len >= 1 && n_steps >= 1 && n_keys >= 1, "rv_crop_segments: empty batch");
    RV_CHECK_ARG(audio && label && audio_begin && label_begin && out_audio && onset && frame, "rv_crop_segments: null pointer");
    hipStream_t st = (hipStream_t)stream;
    CropArgs a;
    a.audio = audio; a.label = label; a.velocity = velocity; a.audio_begin = audio_begin; a.label_begin = label_begin;
    a.seq_len = seq_len; a.nlab = (long)n_steps * n_keys;
    a.out_audio = out_audio; a.onset = onset; a.offset = offset; a.frame = frame; a.out_velocity = out_velocity;
    long bx = cdiv(seq_len / 8 + 1, 256);
    if (bx > 64) bx = 64;                                   // 64 x B workgroups stream a 327 680-sample batch
    hipLaunchKernelGGL(crop_audio_k, dim3((unsigned)bx, B), dim3(256), 0, st, a);
    RV_LAUNCH_CHECK("rv_crop_segments(audio)");
    long lx = cdiv(a.nlab, 256);
    if (lx > 64) lx = 64;
    hipLaunchKernelGGL(crop_label_k, dim3((unsigned)lx, B), dim3(256), 0, st, a);
    RV_LAUNCH_CHECK("rv_crop_segments(labels)");
    return RV_OK;
}

}  // extern "C"
