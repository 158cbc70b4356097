// Declarations shared by the convolution translation units (conv.hip, conv_wino2.hip): launch arguments, LDS / LDS-DMA helpers,
// the Winograd LDS image permutation.  gfx950 only.
#pragma once
#include "common.h"
#include <stdlib.h>
#include <type_traits>

struct ConvArgs {
    const float* in;  int in_ld;  int H, W;       // input spatial dims
    float* out;       int out_ld; int Ho, Wo;     // output spatial dims
    int B, Cin, Cout;
    const float* wpack; const float* bias;
    int nchunk, ntile_n;
    int Pw, Ph;       // pixel grid the GEMM's N dimension runs over (= Ho,Wo for gather; H+eh,W+ew for scatter)
    long npix;        // B*Ph*Pw
    int vec_store;    // out pointer/ld allow 16-byte stores
    int accumulate;   // out += result
    FastDiv fd_pw, fd_plane, fd_w;   // divide by Pw, by Ph*Pw and by the input width W
    double* bn_sums;  // optional [2*Cout] fp64: += per-channel sum / sum of squares of the values written to `out`
    // bn_z != NULL turns bn_sums into the BACKWARD reduction of the BatchNorm+leaky-ReLU whose output this conv's
    // result is the gradient of:  dd = out * lrelu'(z*scale + shift);  sums += (dd, dd * (z - mean) * invstd)
    const float* bn_z; int bn_z_ld; const float* bn_coef; float bn_slope;
};

template <int R> struct VecR;
template <> struct VecR<4> { typedef f32x4 T; };
template <> struct VecR<2> { typedef f32x2 T; };

#ifdef RV_ABLATION
#define ABL(aa) ((aa).ablate)
#else
#define ABL(aa) 0
#endif
struct ConvLdsArgs {
    ConvArgs c;
    int TH, nbands, total_bands, bands_per_wg;
    int nbuf;          // LDS unit buffers (2 or 3): prefetch distance nbuf-1
    int skew;          // nbuf == 3: half of the waves stage after their multiplies
    int ablate;        // ABLATION (timing experiments only)
    int nsplit, xcd;   // n-splits per band group; XCD-aware placement on/off
    int wres;          // conv3x3_wino_k: the weights of ALL chunks stay in LDS for the whole kernel (staged once, with unit 0)
    unsigned in_bytes; // conv3x3_wino_k: bytes of the input view from c.in (buffer-resource range of the staging loads; < 0x3f000000)
};


// Buffer-resource LDS-DMA: 16 bytes per lane from `base + voff` into the wave's LDS slot (lane i -> lds + 16 i); a lane whose offset is
// outside [0, bytes) gets ZEROS written (tools/probes/buffer_lds_oob.hip).  The builtins exist in the device pass only: the host pass
// (which instantiates kernel templates to emit their launch stubs) sees placeholders.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t rv_rsrc_t;
__device__ __forceinline__ rv_rsrc_t rv_make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void rv_buf_lds16(rv_rsrc_t rs, void* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
}
#else
struct rv_rsrc_t { int unused; };
__device__ inline rv_rsrc_t rv_make_rsrc(const void*, unsigned) { return rv_rsrc_t{0}; }
__device__ inline void rv_buf_lds16(rv_rsrc_t, void*, unsigned) {}
#endif

// wait until at most n of this wave's VMEM operations are outstanding (n wave-uniform; clamping down is safe)
__device__ __forceinline__ void wait_vmcnt_le(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}

// hand-issued LDS reads (see the tap loop of conv3x3_lds_k): the compiler does not track them, the caller waits
__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
}
__device__ __forceinline__ void lds_read(f32x4& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
}
__device__ __forceinline__ void lds_read(f32x2& v, unsigned addr) {
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
}
__device__ __forceinline__ void lds_read(float& v, unsigned addr) {
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
}
// compile-time loop (the body sees its index as a constant expression: instruction immediates, register-array indices)
template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(static_cast<F&&>(f));
    }
}

template <int OFF> __device__ __forceinline__ void lds_read_o(f32x4& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_read_o(f32x2& v, unsigned addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}

// wait until at most N of this wave's LDS operations are outstanding (the counter is 4 bits wide: N is clamped to 15, which
// only makes the wait stricter)
template <int N> __device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N > 15 ? 15 : N) : "memory");
}

template <int LAY> __device__ __forceinline__ int wino_slot(int p, int q) {
    return LAY == 1 ? (p & 1) * 32 + (q >> 1) * 16 + (p >> 1) * 2 + (q & 1) : (p & 1) * 32 + (p >> 1) * 4 + q;
}
// ... and its inverse for the DMA side: (pixel of the piece, quad) that lane i fetches
template <int LAY> __device__ __forceinline__ void wino_lane(int i, int& p, int& q) {
    if (LAY == 1) { p = ((i >> 1) & 7) * 2 + (i >> 5); q = ((i >> 4) & 1) * 2 + (i & 1); }
    else { p = ((i >> 2) & 7) * 2 + (i >> 5); q = i & 3; }
}
// byte offset, inside a row, of quad g of the pixel pair index u = X >> 1 (even pixel; the odd one is 512 bytes further)
template <int LAY> __device__ __forceinline__ int wino_pair_off(int u, int g) {
    return (u >> 3) * 1024 + (LAY == 1 ? (g >> 1) * 256 + (u & 7) * 32 + (g & 1) * 16 : (u & 7) * 64 + g * 16);
}

// packed f32 add / subtract on channel pairs (v_pk_add_f32, the subtraction as a neg modifier)
__device__ __forceinline__ f32x2 pk_add(const f32x2 a, const f32x2 b) { return a + b; }
__device__ __forceinline__ f32x2 pk_sub(const f32x2 a, const f32x2 b) { return a - b; }
__device__ __forceinline__ f32x4 pk_add(const f32x4 a, const f32x4 b) {
    const f32x2 lo = (f32x2){a[0], a[1]} + (f32x2){b[0], b[1]}, hi = (f32x2){a[2], a[3]} + (f32x2){b[2], b[3]};
    return (f32x4){lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ f32x4 pk_sub(const f32x4 a, const f32x4 b) {      // (left to the compiler, a <4 x float> fsub is scalarised)
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"((f32x2){a[0], a[1]}), "v"((f32x2){b[0], b[1]}));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"((f32x2){a[2], a[3]}), "v"((f32x2){b[2], b[3]}));
    return (f32x4){lo[0], lo[1], hi[0], hi[1]};
}

// conv_wino2.hip: software-pipelined Winograd F(2x2,3x3) kernel (algo families 0x8NM / 0x9NM / 0xBNM / 0xDNM of rv_conv_fwd)
int rv_launch_conv3x3_wino2(const ConvArgs& a, int NT, int MTW, int nw, int half, int force_th, hipStream_t st);
