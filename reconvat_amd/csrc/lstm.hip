// Bidirectional LSTM recurrence and the ConvStack pooling/dropout pieces of the Onsets&Frames baseline, gfx950.
//
// Reference anchors:
//   nn.LSTM(input, H, batch_first=True, bidirectional=True) ..... model/onset_frame_VAT.py:614 (sequence_model)
//   Onset_Stack.forward_LSTM / Combine_Stack.forward_LSTM ........ model/onset_frame_VAT.py:370-381,401-410
//   nn.MaxPool2d((1, 2)) + nn.Dropout(0.25) ...................... model/onset_frame_VAT.py:336-343
//   nn.Dropout(0.5) behind the ConvStack's Linear ................ model/onset_frame_VAT.py:346-348
//
// The input projections x W_ih^T + b_ih + b_hh of all time steps are one MFMA GEMM (rv_gemm) done by the caller; the
// kernels here run only the sequential part.  One persistent launch covers both directions and all T steps:
//   * direction d, workgroup j owns hidden units [16 j, 16 j + 16) as 16-row MFMA tiles; its 4*KS waves (16 for H = 384)
//     each keep one tile x one K slice of W_hh in VGPRs for the whole sequence (24 registers per lane), laid out as the A
//     operand of v_mfma_f32_16x16x4_f32;
//   * per step the workgroup stages h_{t-1} (B x H) into LDS, issues 24 MFMAs per wave (batch on the N side), adds the
//     partial tiles through LDS, applies the gate non-linearities in the accumulator layout (a lane ends up with the four
//     gates of one (unit, batch) cell), writes h_t into the output tensor -- which is also the exchange buffer for the
//     next step -- and publishes a per-workgroup step counter;
//   * workgroups of one direction synchronise through those counters (release at agent scope; relaxed polling and ONE
//     acquire fence per step); nothing else is shared.  All 2*H/16 workgroups must be co-resident, which 256 CUs
//     guarantee for H <= 1024; a workgroup that waits too long gives up and raises the error word (ops.lstm_check).
// The backward kernel is the same machine run over W_hh^T with the K = 4H reduction split over the workgroup's waves.
#include "common.h"

#define LSTM_SPIN_LIMIT (1 << 22)
#ifndef RV_LSTM_ABL
#define RV_LSTM_ABL 0       // timing ablations (wrong results): 1 no cross-workgroup wait, 2 no h staging loads, 4 no xg/gates traffic
#endif
#ifndef RV_LSTM_STAGE
#define RV_LSTM_STAGE 1   // forward: stage h_{t-1} through LDS (measured faster than per-lane fragment loads from L2)
#endif
#ifndef RV_LSTM_XCD
#define RV_LSTM_XCD 0     // 1: exchange h_t / dpre_t inside one XCD's L2 with the data as its own ready flag (see below).
                          // Functional (all operator tests pass), measured 4.8 instead of 5.3 us per forward step but 10.5
                          // instead of 6.6 us per backward step (16 waves polling dwords), and it pins every launch to XCDs
                          // 0 and 1, so concurrent launches of the multi-stream step do not fit (32 CUs per XCD) and time
                          // out.  Kept as an experiment knob; the counter protocol is the shipped path.
#endif
#define LSTM_SENTINEL 0xFFFFFFFFu      // "not written yet" bit pattern of the exchange tensors (a NaN no kernel here produces)
#ifndef RV_LSTM_KS
#define RV_LSTM_KS 4      // K slices per tile for H = 384: 16 waves per workgroup, 24 MFMAs per wave and step
#endif


struct LstmArgs {
    const float* xg;        // fwd: [B,T,2,4H] gate pre-activations from the input GEMM; bwd: unused
    const float* whh[2];    // [4H, H] per direction (gate order i, f, g, o)
    float* out;             // fwd: [B,T,2H] hidden states
    float* gates;           // [B,T,2,4,H] activated gates (saved by fwd, read by bwd)
    float* cs;              // [B,T,2,H] cell states
    const float* dout;      // bwd: [B,T,2H]
    float* dxg;             // bwd: [B,T,2,4H] gradient wrt xg
    int* flags;             // [2][H/16] step counters + [1] per-launch error word, zeroed by the host wrapper
    int* sticky;            // persistent per-device error word (never cleared here): a time-out is atomicOr'ed into it directly
    int B, T;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// XCD-local exchange (RV_LSTM_XCD).  The dispatcher places workgroup b of a 1-D grid on XCD b % 8 (tools/xcc_probe.hip reads
// XCC_ID: no exception over grids of 48..2048 workgroups), so the kernels run direction d on the workgroups with b % 8 == d
// and retire the others at once: all workgroups that exchange data then share ONE L2, which is the coherence point --
// plain dword stores (the vector L1 is write-through) and L1-bypassing dword loads (relaxed agent-scope atomics, sc1) that
// hit the freshly written lines in that L2.  The exchange tensor is pre-filled with a sentinel and every word is written
// exactly once, so the data is its own ready flag: a reader polls the words it needs until none is the sentinel.  No
// counters, no fences, no L2 write-back / invalidate on the per-step critical path; if the placement assumption ever
// failed, readers would time out (flags[last] set, checked by the host) -- never compute on stale values.
__device__ __forceinline__ void xstore(float* p, float v) {
#if RV_LSTM_XCD
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    *p = v;
#endif
}
__device__ __forceinline__ float xload(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// wait until every workgroup of this direction has published step >= s; wave 0 polls, one counter per lane
template <int NWG>
__device__ __forceinline__ void wait_step(int* flag, int* err, int* sticky, int s, int tid) {
    if ((RV_LSTM_ABL & 1) == 0 && tid < NWG) {
        // poll relaxed (an acquire load would invalidate this XCD's L2 on every iteration, under the kernels of the other
        // streams too); ONE acquire fence once every counter has arrived.  A time-out raises the per-launch error word AND
        // the persistent per-device word (atomics: several streams may run recurrences at once); every poller looks at the
        // per-launch word every 256 polls, so once any workgroup has given up the remaining steps of the launch cost a few
        // hundred polls each instead of LSTM_SPIN_LIMIT (the launch is dead: its results are flagged invalid, the optimiser
        // kernel skips the update -- rv_adam_step's `skip`).
        int spins = 0;
        while (__hip_atomic_load(&flag[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < s) {
            ++spins;
            if ((spins & 255) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            if (spins > LSTM_SPIN_LIMIT) { atomicOr(err, 1); if (sticky) atomicOr(sticky, 1); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if ((RV_LSTM_ABL & 1) == 0 && tid < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // the polling wave drops stale L1 / L2 lines
    __syncthreads();
}

// Forward.  Workgroup (d, j): hidden units [16j, 16j+16) of direction d as four 16-row MFMA tiles (row <-> (unit, gate), so
// that a lane of the accumulator holds the four gates of one (unit, batch) cell).  4*KS waves: wave w works on tile w & 3 and
// on the K slice (w >> 2) of the H-long reduction, with its part of W_hh in registers for the whole sequence; its B
// fragments (h_{t-1}, batch on the N side) are loaded straight from L2, the KS partial tiles are added through LDS.
template <int H, int KS>
__global__ __launch_bounds__(256 * KS) void lstm_fwd_k(LstmArgs a) {
    constexpr int KW = H / KS, NC = KW / 16, NWG = H / 16;
    static_assert(KW % 16 == 0 && NWG <= 64, "K slice must be whole 16-chunks; one polling lane per workgroup");
#if RV_LSTM_XCD
    const int d = blockIdx.x & 7, j = blockIdx.x >> 3;      // direction d lives on XCD d
    if (d >= 2) return;
#else
    const int d = blockIdx.x / NWG, j = blockIdx.x - d * NWG;
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    const int tile = wave & 3, kpart = wave >> 2;
    const int B = a.B, T = a.T;
    __shared__ f32x4 part[KS > 1 ? KS - 1 : 1][4][64];
#if RV_LSTM_STAGE || RV_LSTM_XCD
    __shared__ __attribute__((aligned(16))) float hs[16][H + 4];
#endif
    int* flag = a.flags + d * NWG;
    int* err = a.flags + 2 * NWG;
    int spins = 0;
    bool dead = false;          // set (block-uniformly) after a time-out: stop polling, results are flagged invalid
    (void)flag; (void)spins; (void)dead;

    const int ubase = j * 16 + tile * 4;
    f32x4 wreg[NC];
    {
        const float* wrow = a.whh[d] + ((long)(li & 3) * H + ubase + (li >> 2)) * H + kpart * KW + 4 * g;
#pragma unroll
        for (int c = 0; c < NC; ++c) wreg[c] = *reinterpret_cast<const f32x4*>(wrow + 16 * c);
    }
    const int unit = ubase + g, b = li;
    const bool cell = kpart == 0 && b < B;
    const int brow = li < B ? li : 0;      // lanes beyond the batch duplicate row 0 (their columns are unused)
    float cstate = 0.f;
    for (int s = 0; s < T; ++s) {
        const int t = d ? T - 1 - s : s;
        const long cellbase = (((long)b * T + t) * 2 + d) * 4 * H + unit;
        float pre[4] = {0.f, 0.f, 0.f, 0.f};
        if (cell && (RV_LSTM_ABL & 4) == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) pre[i] = a.xg[cellbase + (long)i * H];
        }
        if (s > 0) {
            const int tp = d ? t + 1 : t - 1;
            f32x4 hb[NC];
#if RV_LSTM_XCD
            // poll h_{t-1} (B x H) into LDS until no word is the sentinel (all threads load, block-wide vote)
            while (true) {
                bool ok = true;
                for (int idx = tid; idx < B * (H / 4); idx += 256 * KS) {
                    const int bb = idx / (H / 4), k4 = idx - bb * (H / 4);
                    const float* src = a.out + ((long)bb * T + tp) * 2 * H + d * H + 4 * k4;
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { v[i] = xload(src + i); ok &= (__float_as_uint(v[i]) != LSTM_SENTINEL); }
                    *reinterpret_cast<f32x4*>(&hs[bb][4 * k4]) = v;
                }
                if (__syncthreads_and(ok || dead)) break;
                if (++spins > LSTM_SPIN_LIMIT) { dead = true; if (tid == 0) { atomicOr(err, 1); if (a.sticky) atomicOr(a.sticky, 1); } }     // block-uniform
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) hb[c] = *reinterpret_cast<const f32x4*>(&hs[brow][kpart * KW + 16 * c + 4 * g]);
#elif RV_LSTM_STAGE
            wait_step<NWG>(flag, err, a.sticky, s, tid);
            // h_{t-1} (B x H) once per workgroup through LDS: one 16-byte load per thread instead of NC per lane
            for (int idx = tid; (RV_LSTM_ABL & 2) == 0 && idx < B * (H / 4); idx += 256 * KS) {
                const int bb = idx / (H / 4), k4 = idx - bb * (H / 4);
                *reinterpret_cast<f32x4*>(&hs[bb][4 * k4]) =
                    *reinterpret_cast<const f32x4*>(a.out + ((long)bb * T + tp) * 2 * H + d * H + 4 * k4);
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < NC; ++c) hb[c] = *reinterpret_cast<const f32x4*>(&hs[brow][kpart * KW + 16 * c + 4 * g]);
#else
            wait_step<NWG>(flag, err, a.sticky, s, tid);
            const float* hp = a.out + ((long)brow * T + tp) * 2 * H + d * H + kpart * KW + 4 * g;
#pragma unroll
            for (int c = 0; c < NC; ++c) hb[c] = *reinterpret_cast<const f32x4*>(hp + 16 * c);
#endif
            f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[c][q], hb[c][q], acc[q & 1], 0, 0, 0);
            f32x4 sum = acc[0] + acc[1];
            if constexpr (KS > 1) {
                if (kpart > 0) part[kpart - 1][tile][lane] = sum;
                __syncthreads();
                if (kpart == 0) {
#pragma unroll
                    for (int k = 1; k < KS; ++k) sum += part[k - 1][tile][lane];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) pre[i] += sum[i];
        }
        if (cell) {
            const float gi = sigmoidf_(pre[0]), gf = sigmoidf_(pre[1]), gg = tanhf(pre[2]), go = sigmoidf_(pre[3]);
            cstate = fmaf(gf, cstate, gi * gg);
            const float h = go * tanhf(cstate);
            xstore(&a.out[((long)b * T + t) * 2 * H + d * H + unit], h);
            if (a.gates && (RV_LSTM_ABL & 4) == 0) {
                a.gates[cellbase] = gi; a.gates[cellbase + H] = gf; a.gates[cellbase + 2 * H] = gg; a.gates[cellbase + 3 * H] = go;
                a.cs[(((long)b * T + t) * 2 + d) * H + unit] = cstate;
            }
        }
#if !RV_LSTM_XCD
        __syncthreads();      // every wave's h_t stores are issued and complete (workgroup-scope release) ...
        if (tid == 0) __hip_atomic_store(&flag[j], s + 1, (RV_LSTM_ABL & 1) ? __ATOMIC_RELAXED : __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // ... and published
#endif
    }
}

// Backward through time.  Step s handles t = (d ? s : T-1-s); the recurrent term of dh_t is W_hh^T dpre_{t'} with t' the
// step handled just before (read from dxg, the exchange buffer).  Workgroup (d, j): units [16j, 16j+16) = the 16 rows of
// ONE tile; its 4*KS waves split the K = 4H reduction (W_hh^T slices in registers, B fragments straight from L2); the
// partial tiles are added through LDS and threads 0..255 (unit, batch) do the cell arithmetic.
template <int H, int KS>
__global__ __launch_bounds__(256 * KS) void lstm_bwd_k(LstmArgs a) {
    constexpr int NW = 4 * KS, KW = 4 * H / NW, NC = KW / 16, NWG = H / 16;
    static_assert(KW % 16 == 0, "K slice must be whole 16-chunks");
#if RV_LSTM_XCD
    const int d = blockIdx.x & 7, j = blockIdx.x >> 3;      // direction d lives on XCD d
    if (d >= 2) return;
#else
    const int d = blockIdx.x / NWG, j = blockIdx.x - d * NWG;
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    const int B = a.B, T = a.T;
    __shared__ float part[2][NW][16][16];       // double-buffered by step parity (no barrier between read and next write)
    int* flag = a.flags + d * NWG;
    int* err = a.flags + 2 * NWG;
    int spins = 0;
    bool dead = false;          // wave-uniform: set after a time-out
    (void)flag; (void)spins; (void)dead;

    // A operand: row li <-> unit 16j + li; k <-> W_hh row wave*KW + 16c + 4g + {0..3}
    float wreg[NC][4];
    {
        const float* wcol = a.whh[d] + ((long)wave * KW + 4 * g) * H + j * 16 + li;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) wreg[c][q] = wcol[(long)(16 * c + q) * H];
    }
    const int unit = j * 16 + ((tid >> 4) & 15), b = tid & 15;
    const bool cell = tid < 256 && b < B;
    const int brow = li < B ? li : 0;
    float dc_carry = 0.f;
    for (int s = 0; s < T; ++s) {
        const int t = d ? s : T - 1 - s;
        const int tfp = d ? t + 1 : t - 1;           // the step the forward pass ran before t (c_{prev})
        const long cellbase = (((long)b * T + t) * 2 + d) * 4 * H + unit;
        float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, ct = 0.f, cp = 0.f, dh = 0.f;
        if (cell) {
            gi = a.gates[cellbase]; gf = a.gates[cellbase + H]; gg = a.gates[cellbase + 2 * H]; go = a.gates[cellbase + 3 * H];
            ct = a.cs[(((long)b * T + t) * 2 + d) * H + unit];
            cp = (tfp >= 0 && tfp < T) ? a.cs[(((long)b * T + tfp) * 2 + d) * H + unit] : 0.f;
            dh = a.dout[((long)b * T + t) * 2 * H + d * H + unit];
        }
        if (s > 0) {
            const int tn = d ? t - 1 : t + 1;        // handled in the previous iteration
            const float* dp = a.dxg + (((long)brow * T + tn) * 2 + d) * 4 * H + wave * KW + 4 * g;
            f32x4 db[NC];
#if RV_LSTM_XCD
            while (true) {                            // every wave polls its own K slice of dpre_{t'}
                bool ok = true;
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { db[c][q] = xload(dp + 16 * c + q); ok &= (__float_as_uint(db[c][q]) != LSTM_SENTINEL); }
                if (__all(ok) || dead) break;
                if (++spins > LSTM_SPIN_LIMIT) { dead = true; if (lane == 0) { atomicOr(err, 1); if (a.sticky) atomicOr(a.sticky, 1); } }
            }
#else
            wait_step<NWG>(flag, err, a.sticky, s, tid);
#pragma unroll
            for (int c = 0; c < NC; ++c) db[c] = *reinterpret_cast<const f32x4*>(dp + 16 * c);
#endif
            f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[c][q], db[c][q], acc[q & 1], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) part[s & 1][wave][4 * g + i][li] = acc[0][i] + acc[1][i];
            __syncthreads();
            if (cell) {
                float r = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) r += part[s & 1][w][(tid >> 4) & 15][b];
                dh += r;
            }
        }
        if (cell) {
            const float th = tanhf(ct);
            const float d_o = dh * th;
            const float dc = fmaf(dh * go, 1.f - th * th, dc_carry);
            const float d_i = dc * gg, d_g = dc * gi, d_f = dc * cp;
            dc_carry = dc * gf;
            xstore(&a.dxg[cellbase], d_i * gi * (1.f - gi));
            xstore(&a.dxg[cellbase + H], d_f * gf * (1.f - gf));
            xstore(&a.dxg[cellbase + 2 * H], d_g * (1.f - gg * gg));
            xstore(&a.dxg[cellbase + 3 * H], d_o * go * (1.f - go));
        }
#if !RV_LSTM_XCD
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&flag[j], s + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
}

// the step counters are zeroed by a kernel, not hipMemsetAsync: a memset node captured into a hipGraph was observed NOT to
// be re-applied (or not in order) on later replays -- the counters and the error word then held stale data
__global__ void lstm_zero_flags_k(int* flags, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = 0;
}

extern "C" long rv_lstm_flag_bytes(int H) { return (long)(2 * (H / 16) + 1) * sizeof(int); }

static int lstm_check(int B, int T, int H) {
    RV_CHECK_ARG(B >= 1 && B <= 16, "rv_lstm: batch %d not in 1..16 (one 16-wide MFMA column tile)", B);
    RV_CHECK_ARG(T >= 1, "rv_lstm: T=%d", T);
    RV_CHECK_ARG(H == 384 || H == 32, "rv_lstm: hidden size %d not instantiated (384, 32)", H);
    return RV_OK;
}

extern "C" int rv_lstm_fwd(const float* xg, const float* whh_fwd, const float* whh_rev, float* out, float* gates, float* cs,
                           int* flags, int* sticky_err, int B, int T, int H, hipStream_t st) {
    if (int rc = lstm_check(B, T, H)) return rc;
    RV_CHECK_ARG(xg && whh_fwd && whh_rev && out && flags, "rv_lstm_fwd: null pointer");
    RV_CHECK_ARG((gates == nullptr) == (cs == nullptr), "rv_lstm_fwd: gates and cs are saved together");
    LstmArgs a = {};
    a.xg = xg; a.whh[0] = whh_fwd; a.whh[1] = whh_rev; a.out = out; a.gates = gates; a.cs = cs; a.flags = flags; a.sticky = sticky_err; a.B = B; a.T = T;
    hipLaunchKernelGGL(lstm_zero_flags_k, dim3(1), dim3(256), 0, st, flags, (int)(rv_lstm_flag_bytes(H) / sizeof(int)));
#if RV_LSTM_XCD
    if (hipMemsetAsync(out, 0xFF, (size_t)B * T * 2 * H * sizeof(float), st) != hipSuccess) { rv_set_error("rv_lstm_fwd: memset failed"); return RV_ELAUNCH; }
    dim3 grid(8 * (H / 16));      // workgroup b -> XCD b % 8; only b % 8 < 2 (one XCD per direction) do work
#else
    dim3 grid(2 * (H / 16));
#endif
    if (H == 384) hipLaunchKernelGGL((lstm_fwd_k<384, RV_LSTM_KS>), grid, dim3(256 * RV_LSTM_KS), 0, st, a);
    else hipLaunchKernelGGL((lstm_fwd_k<32, 2>), grid, dim3(512), 0, st, a);
    RV_LAUNCH_CHECK("lstm_fwd");
    return RV_OK;
}

extern "C" int rv_lstm_bwd(const float* dout, const float* whh_fwd, const float* whh_rev, const float* gates, const float* cs,
                           float* dxg, int* flags, int* sticky_err, int B, int T, int H, hipStream_t st) {
    if (int rc = lstm_check(B, T, H)) return rc;
    RV_CHECK_ARG(dout && whh_fwd && whh_rev && gates && cs && dxg && flags, "rv_lstm_bwd: null pointer");
    LstmArgs a = {};
    a.dout = dout; a.whh[0] = whh_fwd; a.whh[1] = whh_rev; a.gates = const_cast<float*>(gates); a.cs = const_cast<float*>(cs);
    a.dxg = dxg; a.flags = flags; a.sticky = sticky_err; a.B = B; a.T = T;
    hipLaunchKernelGGL(lstm_zero_flags_k, dim3(1), dim3(256), 0, st, flags, (int)(rv_lstm_flag_bytes(H) / sizeof(int)));
#if RV_LSTM_XCD
    if (hipMemsetAsync(dxg, 0xFF, (size_t)B * T * 8 * H * sizeof(float), st) != hipSuccess) { rv_set_error("rv_lstm_bwd: memset failed"); return RV_ELAUNCH; }
    dim3 grid(8 * (H / 16));
#else
    dim3 grid(2 * (H / 16));
#endif
    if (H == 384) hipLaunchKernelGGL((lstm_bwd_k<384, RV_LSTM_KS>), grid, dim3(256 * RV_LSTM_KS), 0, st, a);
    else hipLaunchKernelGGL((lstm_bwd_k<32, 2>), grid, dim3(512), 0, st, a);
    RV_LAUNCH_CHECK("lstm_bwd");
    return RV_OK;
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d((1,2)) over the frequency axis of an NHWC tensor fused with the Dropout that follows it, and a plain
// Dropout.  keep-mask from a counter hash of (seed, *epoch, element index): the backward pass recomputes nothing, it reads
// the one-byte code the forward wrote (bit 0: the odd column won the max, bit 1: kept).  `epoch` (nullable) is a DEVICE
// counter the training step bumps once per iteration, so a launch replayed from a hipGraph still draws a new mask.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned hash32(unsigned x, unsigned seed) {
    x ^= seed; x *= 0x9E3779B1u; x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool keep_draw(long idx, unsigned seed, float p) {
    const unsigned h = hash32((unsigned)idx, seed ^ (unsigned)(idx >> 32) * 0x27D4EB2Fu);
    return (h >> 8) * (1.0f / 16777216.0f) >= p;
}

// one thread = four consecutive channels of one output pixel (C % 4 == 0): 16-byte loads / stores, one packed code word
__global__ __launch_bounds__(256) void pool_drop_fwd_k(const float* x, float* y, unsigned char* code, long rows, int W, int Wo, int C,
                                                       float p, float scale, unsigned seed, const long* epoch) {
    if (epoch) seed ^= (unsigned)(*epoch) * 0x9E3779B1u;
    const int C4 = C >> 2;
    const long n = rows * Wo * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const long rw = i / C4;
        const int wo = (int)(rw % Wo);
        const long r = rw / Wo;
        const float* src = x + ((r * W + 2 * wo) * (long)C + 4 * c4);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + C);
        const long o = rw * C + 4 * c4;
        f32x4 out;
        unsigned packed = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool odd = v1[q] > v0[q];
            const bool keep = p <= 0.f || keep_draw(o + q, seed, p);
            out[q] = keep ? (odd ? v1[q] : v0[q]) * scale : 0.f;
            packed |= (unsigned)((odd ? 1 : 0) | (keep ? 2 : 0)) << (8 * q);
        }
        *reinterpret_cast<f32x4*>(y + o) = out;
        *reinterpret_cast<unsigned*>(code + o) = packed;
    }
}

__global__ __launch_bounds__(256) void pool_drop_fwd_scalar_k(const float* x, float* y, unsigned char* code, long rows, int W, int Wo,
                                                              int C, float p, float scale, unsigned seed, const long* epoch) {
    if (epoch) seed ^= (unsigned)(*epoch) * 0x9E3779B1u;
    const long n = rows * Wo * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long rw = i / C;
        const int wo = (int)(rw % Wo);
        const long r = rw / Wo;
        const float* src = x + ((r * W + 2 * wo) * (long)C + c);
        const float v0 = src[0], v1 = src[C];
        const bool odd = v1 > v0;
        const bool keep = p <= 0.f || keep_draw(i, seed, p);
        y[i] = keep ? (odd ? v1 : v0) * scale : 0.f;
        code[i] = (unsigned char)((odd ? 1 : 0) | (keep ? 2 : 0));
    }
}

// one thread = four consecutive channels of one OUTPUT pixel: writes both input columns (and zeroes the odd tail column)
__global__ __launch_bounds__(256) void pool_drop_bwd_k(const float* dy, const unsigned char* code, float* dx, long rows, int W, int Wo,
                                                       int C, float scale) {
    const int C4 = C >> 2;
    const long n = rows * Wo * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const long rw = i / C4;
        const int wo = (int)(rw % Wo);
        const long r = rw / Wo;
        const long o = rw * C + 4 * c4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(dy + o);
        const unsigned packed = *reinterpret_cast<const unsigned*>(code + o);
        f32x4 d0, d1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned cd = (packed >> (8 * q)) & 0xFF;
            const float v = (cd & 2) ? g[q] * scale : 0.f;
            d0[q] = (cd & 1) ? 0.f : v;
            d1[q] = (cd & 1) ? v : 0.f;
        }
        float* dst = dx + ((r * W + 2 * wo) * (long)C + 4 * c4);
        *reinterpret_cast<f32x4*>(dst) = d0;
        *reinterpret_cast<f32x4*>(dst + C) = d1;
        if ((W & 1) && wo == Wo - 1) *reinterpret_cast<f32x4*>(dst + 2 * C) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

__global__ __launch_bounds__(256) void pool_drop_bwd_scalar_k(const float* dy, const unsigned char* code, float* dx, long rows, int W,
                                                              int Wo, int C, float scale) {
    const long n = rows * W * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long rw = i / C;
        const int w = (int)(rw % W);
        const long r = rw / W;
        const int wo = w >> 1;
        float v = 0.f;
        if (wo < Wo) {
            const long o = (r * Wo + wo) * (long)C + c;
            const unsigned char cd = code[o];
            if ((cd & 2) && (cd & 1) == (w & 1)) v = dy[o] * scale;
        }
        dx[i] = v;
    }
}

__global__ __launch_bounds__(256) void dropout_k(const float* x, float* y, unsigned char* code, const unsigned char* code_in, long n,
                                                 float p, float scale, unsigned seed, const long* epoch) {
    if (epoch) seed ^= (unsigned)(*epoch) * 0x9E3779B1u;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const bool keep = code_in ? (code_in[i] != 0) : (p <= 0.f || keep_draw(i, seed, p));
        y[i] = keep ? x[i] * scale : 0.f;
        if (code) code[i] = keep ? 1 : 0;
    }
}

static inline int ew_grid(long n) { long g = (n + 255) / 256; return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

// x [rows, W, C] -> y [rows, W/2, C]; p: drop probability of the Dropout behind the pool (0 = none)
extern "C" int rv_maxpool_w2_dropout_fwd(const float* x, float* y, unsigned char* code, long rows, int W, int C, float p, unsigned seed,
                                         const long* epoch, hipStream_t st) {
    RV_CHECK_ARG(x && y && code && rows >= 0 && W >= 2 && C >= 1 && p >= 0.f && p < 1.f, "rv_maxpool_w2_dropout_fwd: bad arguments");
    const int Wo = W / 2;
    if (rows == 0) return RV_OK;
    const bool vec = (C & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0 && (((uintptr_t)code) & 3) == 0;
    if (vec) hipLaunchKernelGGL(pool_drop_fwd_k, dim3(ew_grid(rows * Wo * (C / 4))), dim3(256), 0, st, x, y, code, rows, W, Wo, C, p, 1.0f / (1.0f - p), seed, epoch);
    else hipLaunchKernelGGL(pool_drop_fwd_scalar_k, dim3(ew_grid(rows * Wo * C)), dim3(256), 0, st, x, y, code, rows, W, Wo, C, p, 1.0f / (1.0f - p), seed, epoch);
    RV_LAUNCH_CHECK("pool_drop_fwd");
    return RV_OK;
}

extern "C" int rv_maxpool_w2_dropout_bwd(const float* dy, const unsigned char* code, float* dx, long rows, int W, int C, float p,
                                         hipStream_t st) {
    RV_CHECK_ARG(dy && dx && code && rows >= 0 && W >= 2 && C >= 1 && p >= 0.f && p < 1.f, "rv_maxpool_w2_dropout_bwd: bad arguments");
    if (rows == 0) return RV_OK;
    const bool vec = (C & 3) == 0 && ((((uintptr_t)dy) | ((uintptr_t)dx)) & 15) == 0 && (((uintptr_t)code) & 3) == 0;
    if (vec) hipLaunchKernelGGL(pool_drop_bwd_k, dim3(ew_grid(rows * (W / 2) * (C / 4))), dim3(256), 0, st, dy, code, dx, rows, W, W / 2, C, 1.0f / (1.0f - p));
    else hipLaunchKernelGGL(pool_drop_bwd_scalar_k, dim3(ew_grid(rows * W * C)), dim3(256), 0, st, dy, code, dx, rows, W, W / 2, C, 1.0f / (1.0f - p));
    RV_LAUNCH_CHECK("pool_drop_bwd");
    return RV_OK;
}

// forward: code_out receives the keep mask; backward: pass the saved mask as code_in (seed is then ignored) and dy as x
extern "C" int rv_dropout(const float* x, float* y, unsigned char* code_out, const unsigned char* code_in, long n, float p, unsigned seed,
                          const long* epoch, hipStream_t st) {
    RV_CHECK_ARG(x && y && n >= 0 && p >= 0.f && p < 1.f, "rv_dropout: bad arguments");
    if (n == 0) return RV_OK;
    hipLaunchKernelGGL(dropout_k, dim3(ew_grid(n)), dim3(256), 0, st, x, y, code_out, code_in, n, p, 1.0f / (1.0f - p), seed, epoch);
    RV_LAUNCH_CHECK("dropout");
    return RV_OK;
}
