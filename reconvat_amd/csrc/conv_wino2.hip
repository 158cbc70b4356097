// Winograd F(2x2, 3x3) convolution, software-pipelined (round 5): algo families 0x8NM / 0x9NM (8 waves, full / half-chunk patch) and
// 0xBNM / 0xDNM (4 waves = one per SIMD, 512 registers) of rv_conv_fwd.  Replaces every 3x3 Conv2d / ConvTranspose2d forward and
// input gradient of the U-Nets (reference model/UNet_onset.py:186-224) that conv3x3_wino_k (conv.hip) serves; same math, same LDS image,
// same staging, same epilogue -- a different SCHEDULE.
//
// What round 4's kernel left on the table (profiles/r04_pmc_winograd.txt: matrix pipe busy 0.365 / 0.419): inside a unit every wave ran
// [patch ds_reads -> wait -> 64 packed transform adds -> 64 MFMAs] strictly in that order, and since the unit barrier lines all waves of a
// workgroup up, both waves of a SIMD read and transformed at the same time with the matrix pipe idle, then queued for it together.
// Here a wave's instruction stream is ONE continuous chain of MFMAs: while the 16 x NT x (4 | 2) MFMAs of stage i issue, the same wave
//   * reads the 4x4 patch of stage i+1 (16 ds_read, issued during the first four xi steps),
//   * transforms it with packed math in the shadow of the MFMAs (B^T d across the patch rows in place, then (.) B row by row straight into
//     the operand registers of stage i as each group of four becomes dead),
//   * prefetches the next weight fragment, and -- once per unit -- issues the LDS-DMA of the unit after next.
// A stage is one (tile group, half-chunk) of a unit; the patch of the FIRST stage of unit u+1 is read during the LAST stage of unit u, so
// the unit barrier sits in front of that last stage (by then DMA(u+1) has landed and every wave has its last patch of unit u in
// registers) and buffer u%2 can be refilled for unit u+2 right behind it.  The weights of unit u are still being read during that
// stage, so they never share the band's double buffer: either all chunks are resident, or they travel through a ring of three.
// Tile geometry is per kernel, not per band (the tile -> LDS offsets do not depend on the band; only the validity of a row pair does).
#include "conv_shared.h"
#include <mutex>

#ifndef RV_W2_SIDE
#define RV_W2_SIDE 2
#endif

// (the compiler scalarises a <4 x float> add just like the subtraction: two v_pk_add_f32 by hand)
__device__ __forceinline__ f32x2 pk2_add(const f32x2 a, const f32x2 b) { return a + b; }
__device__ __forceinline__ f32x4 pk2_add(const f32x4 a, const f32x4 b) {
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"((f32x2){a[0], a[1]}), "v"((f32x2){b[0], b[1]}));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"((f32x2){a[2], a[3]}), "v"((f32x2){b[2], b[3]}));
    return (f32x4){lo[0], lo[1], hi[0], hi[1]};
}

// AB: timing ablations (wrong results by design; instantiated only in -DRV_W2_DEV builds, selected with RV_W2_ABL=mask):
//   1 no MFMAs | 2 no patch reads / transforms | 4 no weight-fragment reads | 8 no unit barrier | 16 no staging after the prologue |
//   32 no band epilogue | 64 no statistics tail
template <int NT, int MTW, int NW, bool HALF, bool BNZ, int AB = 0>
__global__ __launch_bounds__(NW * 64) void conv3x3_wino2_k(ConvLdsArgs aa) {
    constexpr int NTHR = NW * 64;
    constexpr int KC = 16;                       // channels per chunk
    constexpr int WFLOATS = 16 * NT * 256;       // 16 xi x NT fragments x 64 lanes x 4 floats
    constexpr int LAY = 1;
    const ConvArgs& a = aa.c;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int W = a.W, H = a.H, WT = (W + 1) >> 1, TH = aa.TH;
    const int NP = (2 * WT + 2 + 15) >> 4;               // 1 KiB pieces per staged row
    const int RP = NP * 256;                             // floats per row
    const int nrow = TH + 2;
    const int xfloats = nrow * RP;
    const int vid = aa.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int grp = vid / aa.nsplit, split = vid - grp * aa.nsplit;
    const int nt0 = split * NT;
    float* xs0 = smem;                                   // [2][nrow][NP][256]
    float* ws0 = smem + 2 * xfloats;                     // [nchunk (resident) | 3 (ring)][16][NT][64][4]
    const int band_lo = grp * aa.bands_per_wg;
    const int band_hi = min(band_lo + aa.bands_per_wg, aa.total_bands);
    if (band_lo >= band_hi) return;
    const int nchunk = a.nchunk;
    const bool wres = aa.wres != 0;
    const int nunits = (band_hi - band_lo) * nchunk;

    // ---- staging plan: slot i = (row, piece) of the unit's input rows = 1 KiB at LDS offset i KiB of the band buffer, dealt round-robin
    // over the waves.  The rows are fetched through a buffer resource over IMAGE b (chunk c) of the input view, rebuilt per unit: a lane
    // whose byte offset lies outside it gets ZEROS written to its LDS slot (tools/probes/buffer_lds_oob.hip), so the rows above / below the
    // image need no test at all (their offsets wrap below zero / run past the image) and a halo column is an offset bump of 2^30.  Per slot:
    // one vector add and the DMA instruction; per lane one offset register per slot. ----
    constexpr int TXF = NW == 12 ? 4 : 8;            // (three waves per SIMD leave 168 registers)
    constexpr int NWF = 16 * NT;                         // weight fragments per chunk
    constexpr int TW = (NWF + NW - 1) / NW;
    const int nx = nrow * NP;
    int lp, lq;
    wino_lane<LAY>(lane, lp, lq);
    constexpr unsigned OOB = 0x40000000u;
    const unsigned img_bytes = (unsigned)(((H * W - 1) * a.in_ld + KC) * 4);
    const long img_stride = (long)H * W * a.in_ld * 4;
    unsigned xs_goff[TXF];
#pragma unroll
    for (int t = 0; t < TXF; ++t) {
        const int i = wave + NW * t;
        const int row = i / NP, k = i - row * NP;
        const int px = k * 16 + lp - 1;
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
    int sg_u = 0, sg_buf = 0, sg_b = b_first, sg_y0 = y_first, sg_c = 0, sg_ws = 0;

    // One unit's staging is dealt over the xi steps of a stage in PIECES (each a handful of scalar instructions and one or two DMA
    // issues, in the shadow of that step's MFMAs): piece 0 builds the unit's buffer resource, pieces 1 .. TXF issue one band slot each
    // (TXF also sweeps up what a wide band has beyond TXF slots per wave), TXF + 1 issues the weight fragments and advances the cursor.
    const int nmine = (nx - wave + NW - 1) / NW;         // band slots of this wave
    constexpr int NPIECE = TXF + 2;
    rv_rsrc_t sg_rs = rv_make_rsrc(a.in, 0);
    unsigned sg_ubase = 0;
    auto stage_piece = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k == 0) {
            sg_rs = rv_make_rsrc(reinterpret_cast<const char*>(a.in) + sg_b * img_stride + sg_c * (KC * 4), img_bytes);
            sg_ubase = (unsigned)((sg_y0 - 1) * W * a.in_ld * 4);                  // (row y0 - 1; "negative" for the first band: wraps)
        } else if constexpr (k <= TXF) {
            constexpr int t = k - 1;
            if (t < nmine)
                rv_buf_lds16(sg_rs, reinterpret_cast<char*>(xs0 + sg_buf * xfloats) + wave * 1024 + t * (NW * 1024), xs_goff[t] + sg_ubase);
            if constexpr (k == TXF) {
                for (int i = wave + NW * TXF; i < nx; i += NW) {                    // bands of more than TXF slots per wave
                    const int row = i / NP, kk = i - row * NP;
                    const int px = kk * 16 + lp - 1;
                    const unsigned goff = (unsigned)px < (unsigned)W ? (unsigned)((row * W + px) * a.in_ld + lq * 4) * 4u : OOB;
                    rv_buf_lds16(sg_rs, reinterpret_cast<char*>(xs0 + sg_buf * xfloats) + i * 1024, goff + sg_ubase);
                }
            }
        } else {
            if (!wres || sg_u < nchunk) {
                // resident: chunk c arrives with unit c of the first band; ring: the chunk of this unit into slot sg_u % 3
                const int slot = wres ? sg_u : sg_ws;
                const char* wsrc = reinterpret_cast<const char*>(wino + (long)sg_c * a.ntile_n * 256);
#pragma unroll
                for (int t = 0; t < TW; ++t) {
                    if (wave + NW * t >= NWF) break;
                    glds16(reinterpret_cast<const float*>(wsrc + w_off[t]), ws0 + slot * WFLOATS + (wave + NW * t) * 256);
                }
            }
            ++sg_u;
            sg_buf ^= 1;
            sg_ws = sg_ws == 2 ? 0 : sg_ws + 1;
            if (++sg_c == nchunk) {
                sg_c = 0;
                sg_y0 += TH;
                if (sg_y0 >= H) { sg_y0 = 0; ++sg_b; }
            }
        }
    };
    auto stage = [&]() { sfor<0, NPIECE>([&](auto kc) { stage_piece(kc); }); };

    typedef typename VecR<HALF ? 2 : 4>::T pvec;         // what one patch / weight read delivers
    constexpr int NH = HALF ? 2 : 1;                     // half-chunk passes per tile group
    constexpr int S = MTW * NH;                          // stages per unit
    constexpr int RR = HALF ? 2 : 4;                     // k-steps (MFMAs per xi and n-tile) of a stage
    f32x4 acc[MTW][NT][16];
    pvec d[16];                                          // V = B^T d B of the CURRENT stage (MFMA B operands)
    pvec T[16];                                          // patch of the NEXT stage: raw, then B^T d in place
    pvec wf[2][NT];
    int lb0[MTW], lb1[MTW], tyx[MTW];
    bool tin[MTW];
    const int slots = (TH >> 1) * WT;
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const int t = (wave + NW * m) * 16 + j;
        tin[m] = t < slots;
        const unsigned tt = tin[m] ? (unsigned)t : 0u;
        const int ty = (int)fastdiv(tt, a.fd_pw), tx = (int)tt - ty * WT;       // fd_pw divides by WT here
        tyx[m] = (ty << 16) | tx;
        lb0[m] = (2 * ty) * RP * 4 + wino_pair_off<LAY>(tx, g);
        lb1[m] = (2 * ty) * RP * 4 + wino_pair_off<LAY>(tx + 1, g);
    }
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

    const unsigned rp4 = (unsigned)RP * 4u;
    const unsigned xs_b0 = lds_addr(xs0), xbytes = (unsigned)xfloats * 4u;
    const unsigned ws_b0 = lds_addr(ws0) + lane * 16;

    // patch row `er` of the stage whose pair addresses (patch row 0) are pa0 / pa1, half-chunk h -> T[4 er .. 4 er + 3]
    auto patch_row = [&](auto erc, auto hc, unsigned pa0, unsigned pa1) {
        constexpr int er = decltype(erc)::value, h = decltype(hc)::value;
        const unsigned r0 = pa0 + er * rp4, r1 = pa1 + er * rp4;
        lds_read_o<h * 8>(T[er * 4 + 0], r0);
        lds_read_o<512 + h * 8>(T[er * 4 + 1], r0);
        lds_read_o<h * 8>(T[er * 4 + 2], r1);
        lds_read_o<512 + h * 8>(T[er * 4 + 3], r1);
    };
    // weight fragments of (xi, half h) from the chunk at LDS address wsa (lane offset included) -> wf[slot]
    auto ldw = [&](auto xic, auto hc, auto slotc, unsigned wsa) {
        constexpr int xi = decltype(xic)::value, h = decltype(hc)::value, sl = decltype(slotc)::value;
        sfor<0, NT>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            lds_read_o<(xi * NT + n) * 1024 + h * 8>(wf[sl][n], wsa);
        });
    };
    // the two transform passes, one packed op (q = 0..3) at a time so that they can be dealt over the MFMAs of a step
    auto pass1 = [&](auto ccc, auto qc) {                // across the patch rows, column cc, in place
        constexpr int cc = decltype(ccc)::value, q = decltype(qc)::value;
        if constexpr (q == 0) T[cc] = pk_sub(T[cc], T[8 + cc]);
        if constexpr (q == 1) T[12 + cc] = pk_sub(T[4 + cc], T[12 + cc]);
        if constexpr (q == 2) { const pvec s = pk2_add(T[4 + cc], T[8 + cc]); T[8 + cc] = pk_sub(T[8 + cc], T[4 + cc]); T[4 + cc] = s; }
    };
    auto pass2 = [&](auto rrc, auto qc) {                // along patch row rr, into the operand registers
        constexpr int rr = decltype(rrc)::value, q = decltype(qc)::value;
        if constexpr (q == 0) d[4 * rr] = pk_sub(T[4 * rr], T[4 * rr + 2]);
        if constexpr (q == 1) d[4 * rr + 1] = pk2_add(T[4 * rr + 1], T[4 * rr + 2]);
        if constexpr (q == 2) d[4 * rr + 2] = pk_sub(T[4 * rr + 2], T[4 * rr + 1]);
        if constexpr (q == 3) d[4 * rr + 3] = pk_sub(T[4 * rr + 1], T[4 * rr + 3]);
    };
    // transform work that rides on MFMA group r of step xi (pass 1 has three ops per column: q = 2 does two of the four results)
    auto side_valu = [&](auto xic, auto rc) {
        constexpr int xi = decltype(xic)::value, r = decltype(rc)::value;
        constexpr int q0 = r * 4 / RR, q1 = (r + 1) * 4 / RR;          // ops [q0, q1) of this step ride on group r
        sfor<q0, q1>([&](auto qc) {
            if constexpr (xi >= 4 && xi < 8) pass1(std::integral_constant<int, xi - 4>{}, qc);
            if constexpr (xi == 8) pass2(std::integral_constant<int, 0>{}, qc);
            if constexpr (xi == 10) pass2(std::integral_constant<int, 1>{}, qc);
            if constexpr (xi == 12) pass2(std::integral_constant<int, 2>{}, qc);
        });
    };

    if constexpr ((AB & 4) != 0) {
#pragma unroll
        for (int n = 0; n < NT; ++n) { wf[0][n] = (pvec)(float)lane; wf[1][n] = (pvec)(float)(lane + 1); }
    }
    // ---- prologue: unit 0 (and 1) on their way, the first patch transformed without cover ----
    stage();
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (sg_u < nunits) stage();
    {
        const unsigned pa0 = xs_b0 + lb0[0], pa1 = xs_b0 + lb1[0];
        sfor<0, 4>([&](auto erc) { patch_row(erc, std::integral_constant<int, 0>{}, pa0, pa1); });
        ldw(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, ws_b0);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NT) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        sfor<0, 4>([&](auto cc) { sfor<0, 3>([&](auto qc) { pass1(cc, qc); }); });
        sfor<0, 4>([&](auto rr) { sfor<0, 4>([&](auto qc) { pass2(rr, qc); }); });
    }

    int cu_b = b_first, cu_y0 = y_first, cu_ws = 0;
    unsigned cu_x = xs_b0;                               // LDS address of the current unit's band
    for (int bi = band_lo; bi < band_hi; ++bi) {
        const int b = cu_b, y0 = cu_y0;
        // One unit = one 16-channel chunk of the band.  The FIRST chunk starts its accumulators with a ZERO C operand on the first k-step
        // (an inline constant of the MFMA) instead of 64 v_mov per tile group and n-tile -- vector-ALU instructions are not free next to
        // f32 MFMAs on this chip (tools/probes/mfma_valu_overlap.hip) -- so the unit body exists twice: chunk 0, then the loop over the rest.
        auto unit_body = [&](auto firstc, const int c) {
        // the next unit's chunk / weight slot / band buffer (for the cross-unit prefetches of the last stage)
        const int nc_ = c + 1 == nchunk ? 0 : c + 1;
        const int nws = cu_ws == 2 ? 0 : cu_ws + 1;
        const unsigned ws_cur = ws_b0 + (unsigned)((wres ? c : cu_ws) * WFLOATS * 4);
        const unsigned ws_nxt = ws_b0 + (unsigned)((wres ? nc_ : nws) * WFLOATS * 4);
        const unsigned nx_x = cu_x ^ (xs_b0 ^ (xs_b0 + xbytes));
        constexpr bool first = decltype(firstc)::value;
        sfor<0, S>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int m = s / NH, h = s % NH;
            constexpr bool last = s == S - 1;
            constexpr int m2 = last ? 0 : (s + 1) / NH, h2 = last ? 0 : (s + 1) % NH;
            if constexpr (last && !(AB & 8)) {
                // DMA(u+1) has landed (issued a whole unit ago) and every wave holds its last patch of unit u in registers
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
            const bool more = sg_u < nunits;               // (unit u+2 exists; read before the pieces advance the cursor)
            const unsigned px = last ? nx_x : cu_x;
            const unsigned pa0 = px + lb0[m2], pa1 = px + lb1[m2];
            const unsigned wsn = last ? ws_nxt : ws_cur;
            sfor<0, 16>([&](auto xic) {
                constexpr int xi = decltype(xic)::value;
                // A: the next weight fragment (the first of the next stage at the end)
                if constexpr (!(AB & 4)) {
                    if constexpr (xi < 15) ldw(std::integral_constant<int, xi + 1>{}, std::integral_constant<int, h>{}, std::integral_constant<int, (xi + 1) & 1>{}, ws_cur);
                    else ldw(std::integral_constant<int, 0>{}, std::integral_constant<int, h2>{}, std::integral_constant<int, 0>{}, wsn);
                }
                // B: the next stage's patch, one row per step
                if constexpr (xi < 4 && !(AB & 2)) patch_row(xic, std::integral_constant<int, h2>{}, pa0, pa1);
                // C: wf[xi] has landed (LDS returns in order: count what was issued behind it); at xi == 4 the whole patch as well
                constexpr int behind = ((AB & 4) ? 0 : NT) + ((AB & 2) ? 0 : (xi < 4 ? 4 : 0) + ((xi >= 1 && xi < 4) ? 4 : 0));
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(behind) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (last && xi >= 1 && xi <= NPIECE) {
                    if (more && !(AB & 16)) stage_piece(std::integral_constant<int, xi - 1>{});       // unit u+2 into the buffer every wave has finished reading
                    __builtin_amdgcn_sched_barrier(0);
                }
                // D: the multiplies of this step, the transform riding along (RV_W2_SIDE: 0 = one packed op behind every MFMA group,
                // 1 = the step's ops as one block behind its MFMAs, 2 = the whole transform as one block behind the stage's last MFMA)
                sfor<0, RR>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    if constexpr (!(AB & 1)) {
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            if constexpr (first && h == 0 && r == 0)
                                acc[m][n][xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[xi & 1][n][r], d[xi][r], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                            else
                                acc[m][n][xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[xi & 1][n][r], d[xi][r], acc[m][n][xi], 0, 0, 0);
                        }
                    }
                    if constexpr (!(AB & 2) && RV_W2_SIDE == 0) { side_valu(xic, rc); __builtin_amdgcn_sched_barrier(0); }
                });
                if constexpr (!(AB & 2) && RV_W2_SIDE == 1) sfor<0, RR>([&](auto rc) { side_valu(xic, rc); });
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (!(AB & 2)) {
                if constexpr (RV_W2_SIDE == 2) {
                    sfor<0, 4>([&](auto cc) { sfor<0, 3>([&](auto qc) { pass1(cc, qc); }); });
                    sfor<0, 3>([&](auto rr) { sfor<0, 4>([&](auto qc) { pass2(rr, qc); }); });
                }
                sfor<0, 4>([&](auto qc) { pass2(std::integral_constant<int, 3>{}, qc); });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        cu_x = nx_x;
        cu_ws = nws;
        };
        unit_body(std::true_type{}, 0);
        for (int c = 1; c < nchunk; ++c) unit_body(std::false_type{}, c);
        cu_y0 += TH;
        if (cu_y0 >= H) { cu_y0 = 0; ++cu_b; }
        if (AB & 32) continue;
        // ---- epilogue of this band: Y = A^T M A (packed math on channel pairs), then bias / statistics / store per output pixel ----
        const int th = min(TH, H - y0);
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int ty = tyx[m] >> 16, tx = tyx[m] & 0xffff;
            const int oy = 2 * ty, ox = 2 * tx;
            if (!tin[m] || oy >= th) continue;
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
                    f32x2 bq;
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
    if constexpr ((AB & 32) != 0) {                        // keep the accumulators alive without the epilogue
        f32x4 s4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int xi = 0; xi < 16; ++xi) s4 += acc[m][n][xi];
        if (s4[0] + s4[1] + s4[2] + s4[3] == 1.2345f) a.out[tid] = s4[0];
    }
    if (a.bn_sums && !(AB & 64)) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float uu = st1[n][r], q = st2[n][r];
#pragma unroll
                for (int dd = 1; dd < 16; dd <<= 1) {
                    uu += __shfl_xor(uu, dd, 64);
                    q += __shfl_xor(q, dd, 64);
                }
                st1[n][r] = uu; st2[n][r] = q;
            }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
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

static size_t wino2_bytes(int NT, int TH, int W, int wslots) {
    const int NP = (2 * ((W + 1) / 2) + 2 + 15) / 16;
    return (size_t)2 * (TH + 2) * NP * 1024 + (size_t)wslots * 16 * NT * 1024;
}

template <int NT, int MTW, int NW, bool HALF>
static int launch_wino2(const ConvLdsArgs& aa, dim3 grid, size_t lds, hipStream_t st) {
    // (launches come from the autograd thread as well as from the main thread: once_flag, not a plain static bool)
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        if (hipFuncSetAttribute((const void*)conv3x3_wino2_k<NT, MTW, NW, HALF, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess)
            (void)hipGetLastError();
        if (hipFuncSetAttribute((const void*)conv3x3_wino2_k<NT, MTW, NW, HALF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess)
            (void)hipGetLastError();
    });
    if (aa.c.bn_z) hipLaunchKernelGGL((conv3x3_wino2_k<NT, MTW, NW, HALF, true>), grid, dim3(NW * 64), lds, st, aa);
    else hipLaunchKernelGGL((conv3x3_wino2_k<NT, MTW, NW, HALF, false>), grid, dim3(NW * 64), lds, st, aa);
    return RV_OK;
}

// A band of TH (even) rows holds (TH/2) x ceil(W/2) tiles of 2x2 outputs, NW x MTW groups of 16 tiles per unit.  force_th = 0: as many
// rows as the tile slots (and the LDS) hold.
int rv_launch_conv3x3_wino2(const ConvArgs& a0, int NT, int MTW, int nw, int half, int force_th, hipStream_t st) {
    if (NT < 1 || a0.ntile_n % NT) return RV_EUNSUPPORTED;
    ConvLdsArgs aa;
    aa.c = a0;
    const long in_bytes = (((long)a0.B * a0.H * a0.W - 1) * a0.in_ld + a0.Cin) * 4;
    if (in_bytes >= 0x3f000000L) return RV_EUNSUPPORTED;      // the staging loads address the input view with 30-bit offsets
    aa.in_bytes = (unsigned)in_bytes;
    const int WT = (a0.W + 1) / 2;
    aa.c.fd_pw = fastdiv_make((unsigned)WT);
    const int trows = (nw * MTW * 16) / WT;
    if (trows < 1) return RV_EUNSUPPORTED;
    int TH = 2 * trows;
    if (TH > a0.H) TH = (a0.H + 1) & ~1;
    if (force_th) {
        if (force_th > TH || (force_th & 1)) return RV_EUNSUPPORTED;
        TH = force_th;
    }
    const size_t cap = 154 * 1024;
    // weights: resident when all chunks fit next to the two band buffers, else a ring of three chunk slots (the chunk of unit u is still
    // being read while unit u+2 is on its way)
    const int ring = a0.nchunk < 3 ? a0.nchunk : 3;
    size_t lds = wino2_bytes(NT, TH, a0.W, ring);
    while (!force_th && lds > cap && TH > 2) {
        TH -= 2;
        lds = wino2_bytes(NT, TH, a0.W, ring);
    }
    aa.wres = a0.nchunk <= 3 ? 1 : 0;
    if (!aa.wres) {
        const size_t lds_res = wino2_bytes(NT, TH, a0.W, a0.nchunk);
        if (lds_res <= cap) { aa.wres = 1; lds = lds_res; }
    }
    if (lds > cap) return RV_EUNSUPPORTED;
    aa.TH = TH; aa.nbands = cdiv(a0.H, TH);
    aa.total_bands = a0.B * aa.nbands;
    const int nsplit = a0.ntile_n / NT;
    int wgs = 256 / nsplit;
    if (wgs < 1) wgs = 1;
    if (wgs > aa.total_bands) wgs = aa.total_bands;
    aa.bands_per_wg = cdiv(aa.total_bands, wgs);
    wgs = cdiv(aa.total_bands, aa.bands_per_wg);
    aa.nbuf = 2; aa.skew = 0; aa.ablate = 0;
    static const int xcd_env = getenv("RV_CONV_XCD") ? atoi(getenv("RV_CONV_XCD")) : 1;
    aa.nsplit = nsplit; aa.xcd = xcd_env;
    const dim3 grid(wgs * nsplit);
#define RV_W2(nt, mt, nwv, hf) \
    if (NT == nt && MTW == mt && nw == nwv && (half != 0) == hf) return launch_wino2<nt, mt, nwv, hf>(aa, grid, lds, st);
    // (instantiated: the tiles that fit the register file without scratch; the 8-wave NT = 2 tile does not carry the fused
    // BatchNorm-backward epilogue -- refused, like the 12-wave tile of conv3x3_wino_k)
    if (nw == 8 && half && NT == 2 && a0.bn_z) return RV_EUNSUPPORTED;
#ifdef RV_W2_DEV
    if (const char* e = getenv("RV_W2_ABL")) {
        const int mask = atoi(e);
        if (NT == 1 && MTW == 1 && nw == 8 && !half && !a0.bn_z) {
#define RV_W2A(m) if (mask == m) { hipFuncSetAttribute((const void*)conv3x3_wino2_k<1, 1, 8, false, false, m>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024); \
            hipLaunchKernelGGL((conv3x3_wino2_k<1, 1, 8, false, false, m>), grid, dim3(512), lds, st, aa); return RV_OK; }
            RV_W2A(126) RV_W2A(122) RV_W2A(120) RV_W2A(96) RV_W2A(64) RV_W2A(1) RV_W2A(2) RV_W2A(26) RV_W2A(24) RV_W2A(30) RV_W2A(124) RV_W2A(32)
#undef RV_W2A
        }
    }
#endif
    RV_W2(1, 1, 8, false) RV_W2(1, 1, 8, true) RV_W2(2, 1, 8, true) RV_W2(1, 2, 8, true)
    RV_W2(1, 2, 4, false) RV_W2(2, 1, 4, false)
    if (nw == 12 && a0.bn_z) return RV_EUNSUPPORTED;       // (three waves per SIMD: no room for the fused BatchNorm-backward epilogue)
    if (NT == 1 && MTW == 1 && nw == 12 && half) {
        static std::once_flag attr12;
        std::call_once(attr12, [] { (void)hipFuncSetAttribute((const void*)conv3x3_wino2_k<1, 1, 12, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024); });
        hipLaunchKernelGGL((conv3x3_wino2_k<1, 1, 12, true, false>), grid, dim3(768), lds, st, aa);
        return RV_OK;
    }
#undef RV_W2
    return RV_EUNSUPPORTED;
}
