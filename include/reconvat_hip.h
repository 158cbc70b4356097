/*
 * reconvat_hip.h -- C ABI of libreconvat_hip.so (gfx950 / MI355X only).
 *
 * The reference (KinWaiCheuk/ReconVAT) has no native layer: its hot path is a chain of generic
 * PyTorch ops.  Each entry point below replaces the PyTorch call sites cited next to it (paths are
 * relative to the reference root).  Conventions, for every function:
 *   - plain `extern "C"`, raw DEVICE pointers, explicit sizes/strides, fp32 data;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); the call only ENQUEUES work:
 *     it never allocates, never synchronises, never owns memory (the host framework's allocator owns
 *     every buffer; scratch is passed in as `workspace`, sized by the *_workspace_bytes queries);
 *   - returns 0 on success, <0 on error (-1 bad argument, -2 launch failure, -3 unsupported shape);
 *     `rv_last_error()` returns the text of the last error on the calling thread;
 *   - re-entrant per stream; the ONE piece of process-global mutable state is the weight-gradient plan table written by
 *     rv_conv_wgrad_set_plan (configure it once per shape before the first launch of that shape: the shipped plan table does
 *     exactly that; it is not synchronised against concurrent launches of the same shape).
 *
 * Activations are NHWC with an explicit pixel stride `*_ld` (floats between consecutive pixels), so a
 * tensor may be a channel slice of a wider buffer (the decoder's concat buffers).
 */
#ifndef RECONVAT_HIP_H
#define RECONVAT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

int rv_abi_version(void);
const char* rv_last_error(void);
/* first 16 hex digits of the sha256 over the sources (reconvat_amd/csrc: *.hip, *.h, *.cpp) this library was built from */
const char* rv_source_digest(void);

/* ---- log-Mel front-end ------------------------------------------------------------------------
 * replaces nnAudio MelSpectrogram.forward / STFT.forward (model/Spectrogram.py:187-231, :443-461),
 * torch.log(spec + 1e-5) and Normalization('imagewise').transform (model/UNet_onset.py:419-423,
 * :432-442; model/utils.py:94-100).  audio [B, nsamp] -> out [B, T, n_mels], T = 1 + nsamp/hop.
 * twiddle: [1024][2] = (cos, -sin)(2*pi*k/2048); window: [2048]; sparse mel rows: start/len/w[n_mels][mel_ld].
 * workspace: 2*B uint32. */
int rv_melspec_lognorm_fwd(const float* audio, long audio_stride, int B, int nsamp, const float* window,
                           const float* twiddle, const int* mel_start, const int* mel_len, const float* mel_w, int mel_ld,
                           int n_mels, int hop, int do_log, int normalise, float* out, int T, void* workspace, void* stream);

/* ---- convolutions (nn.Conv2d / nn.ConvTranspose2d call sites, model/UNet_onset.py:173-224) ------
 * rv_pack_weights: PyTorch-layout weight -> MFMA fragment order for a logical Wm[tap][k][n]
 *   value(tap,k,n) = w[k*s_k + n*s_n + (flip ? taps-1-tap : tap)];  scatter_cmid>0: 2x2/s2 scatter GEMM.
 *   3x3 weights with kdim % 16 == 0 also carry their Winograd F(2x2,3x3) transform U = G g G^T (16 fragments per chunk and
 *   n-tile) behind the nine tap fragments; rv_packed_weight_floats counts both.
 * rv_conv_fwd mode: 0 = 3x3 s1 p1, 1 = 1x1, 2 = 2x2/s2 gather (down fwd, up dgrad), 3 = 2x2/s2 scatter
 *   (ConvTranspose2d(k=2,s=2)(x, output_size=...) fwd, down dgrad).  Forward AND input-gradient of
 *   every layer are instances of it (the packing decides which).  algo: 0 default, 1 LDS-free direct kernel,
 *   2 LDS/DMA-pipelined kernel (3x3 only); forced tiles for the host autotuner (ops._conv_call), RV_EUNSUPPORTED when the tile
 *   does not fit the shape: 0x100|NT<<4|MT direct kernel, 0x200 / 0x300 / 0x400 / 0x700 |NT<<4|MTW LDS kernel with 4 / 8 / 16 /
 *   12 waves per workgroup (MTW in {3,5,6} for 12 waves: 9/15/18 tiles per SIMD and band), optionally TH<<12 = rows per band
 *   (<= what the tile slots hold); 0x600 |NT<<4|MTW: Winograd F(2x2,3x3) form of the persistent 3x3 kernel with 8 waves per
 *   workgroup (Cin % 16 == 0; NT = MTW = 1; TH<<12 = even rows per band), same epilogue fusions; 0xA00 / 0xC00: the same with the
 *   patch read half a chunk at a time, 8 waves (NT in {1,2}, MTW = 1) / 12 waves (NT = MTW = 1); 0x800 / 0x900 / 0xB00 / 0xD00 |NT<<4|MTW (round 5,
 *   conv_wino2.hip): the software-pipelined Winograd kernel -- 8 waves with the full-chunk patch (NT = MTW = 1), 8 waves with the
 *   half-chunk patch ((NT, MTW) in {(1,1), (2,1), (1,2)}), 4 waves with 512 registers ((1,2), (2,1)), 12 waves with the half-chunk patch (NT = MTW = 1); same results as 0x600 up to the
 *   fp32 summation order, same epilogue fusions (the 8-wave NT = 2 tile refuses bn_z).  bn_sums (nullable,
 *   rv_bn_workspace_bytes(Cout) bytes of fp64 = 8 replicas of [2*Cout] that the consumer adds up, += ): per-channel sum / sum of squares of the written output, i.e. the batch statistics of the
 *   BatchNorm2d that consumes it (conv -> bn call sites, model/UNet_onset.py:196-198,221-223), produced in the conv
 *   epilogue; pass the same buffer to rv_bn_lrelu_fwd as `workspace` with sums_ready = 1.  bn_z (nullable; with
 *   bn_z_ld, bn_coef = that layer's saved [5C] coefficients, bn_slope): the call is an INPUT-GRADIENT whose result is
 *   d(loss)/d(output of a BatchNorm+leaky_relu with pre-normalisation input bn_z); bn_sums then receives that layer's
 *   backward reduction (sum dd, sum dd*xhat) and rv_bn_lrelu_bwd(..., sums_ready = 1) skips its own pass over dy, z.
 * rv_conv_wgrad: G[tap][a][b] = sum_p U[f(p,tap)][a]*V[p][b] (+ column sums of V for the bias), written
 *   to dw[a*s_a + b*s_b + tap'] / dbias[b]; mode 0 = 3x3, 1 = 1x1, 2 = 2x2/s2. */
long rv_packed_weight_floats(int taps, int kdim, int ndim);
int rv_pack_weights(const float* w, float* out, int taps, int kdim, int ndim, long s_k, long s_n, int flip,
                    int scatter_cmid, int force_plain, void* stream);
/* batched packing (all weights of the model, one launch per optimiser step): fill a HOST table entry by entry
 * (i = 0, 1, ... in order; returns the running workgroup total, <0 on error), copy it to the device, run it. */
long rv_pack_table_entry_bytes(void);
long rv_pack_table_fill(void* table_host, int i, const float* w, float* out, int taps, int kdim, int ndim, long s_k, long s_n,
                        int flip, int scatter_cmid, int force_plain);
int rv_pack_table_run(const void* table_dev, int count, long total_blocks, void* stream);
int rv_conv_fwd(int mode, const float* in, int in_ld, int B, int H, int W, int Cin, float* out, int out_ld, int Ho,
                int Wo, int Cout, const float* wpack, const float* bias, int accumulate, int algo, double* bn_sums,
                const float* bn_z, int bn_z_ld, const float* bn_coef, float bn_slope, void* stream);
long rv_conv_wgrad_workspace_bytes(int taps, int B, int Hv, int Ca, int Cb);
/* Host autotuner hook (reconvat_amd/ops.py conv_wgrad): the launch partition of the MFMA weight-gradient kernel for this shape
 * from now on -- nw = waves per workgroup (4 / 8, 0 = default; 24 = eight waves, Winograd F(3x3, 2x2) form of the 3x3 kernel,
 * wgrad_wino_k: even Hv, falls back to the direct form where its six-row ring does not fit LDS), wgs = workgroups on the chip
 * (0 = default 256).  Changes the result of rv_conv_wgrad_workspace_bytes for the shape; not thread-safe against concurrent
 * weight-gradient calls. */
int rv_conv_wgrad_set_plan(int taps, int B, int Hv, int Ca, int Cb, int nw, int wgs);
int rv_conv_wgrad(int mode, const float* U, int u_ld, int Hu, int Wu, int Ca, const float* V, int v_ld, int Hv, int Wv,
                  int Cb, int B, float* dw, long s_a, long s_b, int flip, float* dbias, int accumulate, void* workspace,
                  long workspace_bytes, void* stream);
/* Deferred weight-gradient reductions: rv_conv_wgrad_deferred launches only the partial-sum kernel of rv_conv_wgrad and
 * writes the pending reduction (which ADDS into dw / dbias, fp32 atomics) to *entry_host (rv_wgrad_table_entry_bytes() bytes
 * of HOST memory); it returns the number of workgroups that reduction needs (> 0) or a negative status.  After the last
 * entry, rv_wgrad_table_finalize(table_host, count) turns the counts into block offsets and returns the total; a DEVICE copy
 * of the table then runs every reduction of a backward pass in ONE launch (rv_wgrad_reduce_table).  Workspaces must stay
 * untouched until then.  (One reduction launch per layer is launch-latency bound: 161 launches of ~5 us per step.) */
long rv_conv_wgrad_deferred(int mode, const float* U, int u_ld, int Hu, int Wu, int Ca, const float* V, int v_ld, int Hv, int Wv,
                            int Cb, int B, float* dw, long s_a, long s_b, int flip, float* dbias, void* workspace,
                            long workspace_bytes, void* entry_host, void* stream);
/* Segmented forms: the pixel reduction runs over nseg (1..4) runs of Bseg images, run s at U[s] / V[s] (identical geometry and strides):
 * the (input, dY) pairs of ONE layer from several backward passes of a training step (model/UNet_onset.py:383,117-146: the transcriber
 * is back-propagated three times per step on one stream) folded by ONE launch instead of one per pass -- the fixed cost of a
 * weight-gradient launch (prologue, accumulator fold, partial sums, reduction entry) is paid once.  Plan and workspace: those of
 * B = nseg * Bseg images.  RV_EUNSUPPORTED for the small-channel layers (launch those per pass). */
int rv_conv_wgrad_seg(int mode, int nseg, const float* const* U, const float* const* V, int u_ld, int Hu, int Wu, int Ca, int v_ld, int Hv,
                      int Wv, int Cb, int Bseg, float* dw, long s_a, long s_b, int flip, float* dbias, int accumulate, void* workspace,
                      long workspace_bytes, void* stream);
long rv_conv_wgrad_deferred_seg(int mode, int nseg, const float* const* U, const float* const* V, int u_ld, int Hu, int Wu, int Ca, int v_ld,
                                int Hv, int Wv, int Cb, int Bseg, float* dw, long s_a, long s_b, int flip, float* dbias, void* workspace,
                                long workspace_bytes, void* entry_host, void* stream);
long rv_wgrad_table_entry_bytes(void);
long rv_wgrad_table_finalize(void* table_host, int count);
int rv_wgrad_reduce_table(const void* table_dev, int count, long total_blocks, void* stream);


/* ---- BatchNorm2d(momentum=0.1) + leaky_relu (+ residual) (model/UNet_onset.py:183,196-199,221-223) --
 * coef [5C] = mean, invstd, scale, shift, unbiased batch variance (saved for backward).  training: 0 eval, 1 train,
 * 2 train without touching the running statistics (rv_bn_running_update applies that update later, in sequence).  workspace: rv_bn_workspace_bytes(C) bytes that are
 * ALL-ZERO on entry (fp64 per-channel sums accumulate there); the host carves them from one arena cleared per step.
 * sums_ready != 0: the workspace already holds the sums of z (rv_conv_fwd's bn_sums) and the statistics pass is skipped. */
long rv_bn_workspace_bytes(int C);
int rv_bn_lrelu_fwd(const float* z, int z_ld, long P, int C, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, long* num_batches_tracked, float momentum, float eps, int training, float slope,
                    const float* res, int res_ld, float* y, int y_ld, float* coef, void* workspace, int sums_ready,
                    void* stream);
/* the same with the residual evaluated in place: y = leaky_relu(bn(z)) + skip(x), skip = the 1x1 convolution of an encoder block (model/UNet_onset.py:186-201:
 * `x12 += self.skip(x)`), x [P][cin] at pixel stride x_ld (floats), w [C][cin] (the PyTorch Conv2d weight), b [C] nullable; cin = 1 (block 1: the single-channel
 * spectrogram, one fma per element) or 16 / 32 / 64 (an fmaf chain per element in the k order of rv_conv_fwd's MFMA kernel; ksplit != 0: the K-split form of
 * that kernel, algo family 0x5NM).  BIT-IDENTICAL to rv_conv_fwd(mode 1) + rv_bn_lrelu_fwd(res = its output): v_mfma_f32_16x16x4_f32 is a chain of fused
 * multiply-adds in k order (tools/probes/mfma_f32_order.hip).  Saves the conv launch, the write of its result and the read of it here. */
int rv_bn_lrelu_fwd_skip(const float* z, int z_ld, long P, int C, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, long* num_batches_tracked, float momentum, float eps, int training, float slope,
                         const float* x, int x_ld, int cin, const float* w, const float* b, int ksplit, float* y, int y_ld, float* coef,
                         void* workspace, int sums_ready, void* stream);
int rv_bn_running_update(float* running_mean, float* running_var, long* num_batches_tracked, const float* coef, int C,
                         float momentum, void* stream);
/* all deferred running-statistic updates of a step in one launch: table (DEVICE int64 words) = nlayers x {running_mean,
 * running_var, num_batches_tracked, C, first, count} followed by the coef pointers; layer l applies coefs[first .. first+count)
 * in order. */
int rv_bn_running_update_table(const long* table, int nlayers, float momentum, void* stream);
int rv_bn_lrelu_bwd(const float* dy, int dy_ld, const float* z, int z_ld, long P, int C, const float* coef, float slope,
                    int frozen, float* dz, int dz_ld, float* dgamma, float* dbeta, int param_accumulate, void* workspace,
                    int sums_ready, void* stream);

/* ---- linear layers (nn.Linear / torch.sigmoid call sites, model/UNet_onset.py:50-52,62-64,275,
 * 292-293,307-313,324,330): C[m*scm+n*scn] (+)= act(sum_k A[m*sam+k*sak]*B[k*sbk+n*sbn] + bias[n]);
 * splitk > 1: the reduction is split over workgroups; splitk_ws == NULL: fp32 atomic accumulation (arrival-order rounding; act 0,
 * no C2; the parameter-gradient GEMMs); splitk_ws != NULL: DETERMINISTICALLY -- every k slice parks its partial tile in splitk_ws
 * (rv_gemm_splitk_workspace_bytes, uninitialised) and a second launch on the same stream adds the slices in k order and runs the
 * epilogue, so the result does not depend on arrival order and no atomics touch C (splitk_tickets: unused since round 5, may be
 * NULL; rv_gemm_splitk_ticket_bytes returns 0 -- the in-kernel ticketed fold it served needed device-scope fences that cost
 * ~30 us per slice on this multi-XCD part); batch > 1: problem z of `batch` equal-shape problems lives at A + z*bsa,
 * B + z*bsb, C + z*bsc (the per-head relative-position gradient of the attention).  a_rowsum (nullable, batch == 1):
 * a_rowsum[m] += sum_k A[m][k] (k slices folded in order): the bias gradient of a linear layer rides on its weight-gradient
 * GEMM (A = dY^T). */
long rv_gemm_splitk_workspace_bytes(int M, int N, int splitk, int batch);
long rv_gemm_splitk_ticket_bytes(int M, int N, int splitk, int batch);
int rv_gemm(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long scm, long scn, float* C2,
            long sc2m, long sc2n, const float* bias, int M, int N, int K, int act, int accumulate, int splitk, int batch,
            long bsa, long bsb, long bsc, float* a_rowsum, void* splitk_ws, void* splitk_tickets, void* stream);
/* grouped launch of independent ACCUMULATING problems of one operand orientation (the parameter-gradient GEMMs of a backward
 * pass, each too small to fill the chip): fill HOST entries (returns the orientation code, <0 on error), finalize (returns the
 * total workgroup count; *fold_blocks <- that of the fold launch), copy the table to the device, run.  The problems ADD into their
 * destinations (two entries may share one): final adds are fp32 atomics.  Split-K inside a table: with a per-entry workspace
 * (rv_gemm_splitk_workspace_bytes) the k slices are parked and a second grouped launch folds them in k order -- one atomic per
 * output element; splitk_ws == NULL: every slice adds atomically. */
long rv_gemm_table_entry_bytes(void);
long rv_gemm_table_fill(void* entry_host, const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long scm,
                        long scn, const float* bias, int M, int N, int K, int splitk, int batch, long bsa, long bsb, long bsc,
                        float* a_rowsum, void* splitk_ws);
long rv_gemm_table_finalize(void* table_host, int count, long* fold_blocks);
int rv_gemm_table_run(const void* table_dev, int count, long total_blocks, long fold_blocks, int orientation, void* stream);
int rv_sigmoid_bwd(const float* g1, int ld1, const float* g2, int ld2, const float* y, int ldy, float* dz, int ldz, long M,
                   int N, void* stream);
int rv_colsum(const float* x, int ld, long M, int N, float* out, int accumulate, void* stream);
/* the same without atomics (the host side's RV_DETERMINISTIC=1 mode: bit-reproducible parameter gradients): row blocks leave their
 * partial sums in ws (rv_colsum_ordered_workspace_bytes, uninitialised), a second pass folds them in block order */
long rv_colsum_ordered_workspace_bytes(long M, int N);
int rv_colsum_ordered(const float* x, int ld, long M, int N, float* out, int accumulate, void* ws, void* stream);
/* out[i] (+)= per-channel sum c0 + i of a statistics workspace a conv's fused epilogue filled (rv_conv_fwd bn_sums, C channels): the
 * column sums of that conv's output = the bias gradient of the layer consuming it as dY (the 2x2 up-conv), without re-reading it */
int rv_sums_fold(const double* sums, int C, int c0, int n, float* out, int accumulate, void* stream);

/* ---- 31-frame local multi-head attention (MutliHeadAttention1D.forward, model/UNet_onset.py:56-91)
 * q,k,v (and dq,dk,dv): rows of G*dh floats with row stride ld / dld (column slices of one fused projection buffer
 * are fine); out, dout [B,L,G*dh] contiguous; relT [31,G*dh] = the `rel` parameter transposed (rv_pack_weights with
 * taps=1, kdim=31, ndim=G*dh, s_k=1, s_n=31, force_plain=1 produces it); att, de [B,L,G,31]. */
int rv_local_attn_fwd(const float* q, const float* k, const float* v, long ld, const float* relT, float* out, float* att, int B,
                      int L, int G, int dh, void* stream);
int rv_local_attn_bwd(const float* dout, const float* q, const float* k, const float* v, long ld, const float* relT,
                      const float* att, float* dq, float* dk, float* dv, long dld, float* de, int B, int L, int G, int dh,
                      void* stream);

/* ---- VAT primitives (UNet_VAT.forward / _l2_normalize, model/UNet_onset.py:126-151,165-171) --------
 * x_adv = clamp(x + scale * rownormalise(prescale*d), 0, 1) over rows of n elements. */
int rv_vat_perturb_fwd(const float* x, const float* d, long rows, int n, float prescale, float scale, float* x_adv,
                       float* r_out, float* dn_out, int* nan_flag, void* stream);
int rv_vat_perturb_bwd(const float* g, const float* x, const float* d, long rows, int n, float prescale, float scale,
                       float* gd, void* stream);

/* ---- losses (F.binary_cross_entropy / F.mse_loss / .abs().mean(), model/UNet_onset.py:136-137,
 * 157-158,471-483); kind 0 BCE, 1 MSE, 2 mean|p|, 3 sqrt(sum p^2).  workspace: rv_reduce_workspace_bytes(n).  ticket (nullable): a
 * zeroed device word (left zero): the last workgroup folds the partial sums itself -- one launch instead of two. */
long rv_reduce_workspace_bytes(long n);
int rv_reduce_mean(int kind, const float* p, const float* t, long n, float* out, void* workspace, unsigned* ticket, void* stream);
int rv_loss_bwd(int kind, const float* p, const float* t, long n, const float* gout, float* gp, void* stream);

/* ---- optimiser (torch.optim.Adam + StepLR + clip_grad_norm_, train_UNet_Onset_VAT.py:113,124;
 * model/helper_functions.py:602-607) on flat buffers; *step = optimiser steps already taken.  skip (nullable): device int;
 * when *skip != 0 (a kernel of this step flagged its results invalid, see rv_lstm_fwd) the update is not applied. */
int rv_adam_step(float* p, const float* g, float* m, float* v, long n, const long* step, float lr0, long decay_steps,
                 float decay_rate, float beta1, float beta2, float eps, float grad_scale, const int* skip, void* stream);
int rv_counter_add(long* counter, long inc, const int* skip_if_set, void* stream);   /* skip_if_set nullable: no add while *skip != 0 */
int rv_clip_scale(float* g, long n, const float* total_norm, float max_norm, void* stream);

/* ---- data feed: PianoRollAudioDataset.__getitem__ (model/dataset.py:35-69) for a whole batch on the device.
 * audio: int16 corpus; label, velocity (nullable): uint8 corpora ([steps, n_keys] rolls, label 3 = onset, 2 = frame,
 * 1 = offset); audio_begin / label_begin: DEVICE arrays of B element offsets (the host draws them with the reference's
 * RandomState rule).  out_audio [B, seq_len] = int16 / 32768; onset / offset (nullable) / frame / out_velocity (nullable)
 * [B, n_steps, n_keys] = (label == 3) / (label == 1) / (label > 1) / velocity / 128, all float32, bit-exact. */
int rv_crop_segments(const short* audio, const unsigned char* label, const unsigned char* velocity, const long* audio_begin,
                     const long* label_begin, int B, long seq_len, int n_steps, int n_keys, float* out_audio, float* onset,
                     float* offset, float* frame, float* out_velocity, void* stream);

/* ---- Onsets&Frames baseline pieces (model/onset_frame_VAT.py:321-415,603-635) -------------------------------------
 * Bidirectional one-layer nn.LSTM(batch_first=True) recurrence (the `sequence_model` of Onset_Stack / Combine_Stack,
 * model/onset_frame_VAT.py:614,370-381,401-410).  The caller computes the input projections of every step with rv_gemm:
 *   xg [B,T,2,4H] = x W_ih^T + b_ih + b_hh   (direction-major halves, PyTorch gate order i,f,g,o)
 * and this entry runs the T sequential steps of both directions in ONE persistent launch (W_hh [4H,H] per direction held
 * in registers as MFMA operands, h exchanged through `out`, per-workgroup step counters in `flags`).
 * out [B,T,2H] (forward half | reverse half, as nn.LSTM returns it); gates [B,T,2,4,H] (activated) and cs [B,T,2,H]
 * are saved for the backward pass (both NULL = inference).  flags: rv_lstm_flag_bytes(H) bytes of device scratch, reset by
 * the call itself; the last int is non-zero afterwards if a workgroup gave up waiting (co-residency violated).  sticky_err
 * (nullable): ONE persistent device int per device that the kernels atomicOr a time-out into and never clear -- the host
 * reads and clears it at its own sync points, and rv_adam_step(skip = sticky_err) refuses to apply a poisoned step.
 * B <= 16 (the batch is the N side of one 16-wide MFMA tile); H in {384, 32}.  rv_lstm_bwd: dout [B,T,2H] -> dxg [B,T,2,4H]; dW_ih, dW_hh, db and dx are GEMMs / column sums
 * of dxg done by the caller. */
long rv_lstm_flag_bytes(int H);
int rv_lstm_fwd(const float* xg, const float* whh_fwd, const float* whh_rev, float* out, float* gates, float* cs, int* flags,
                int* sticky_err, int B, int T, int H, void* stream);
int rv_lstm_bwd(const float* dout, const float* whh_fwd, const float* whh_rev, const float* gates, const float* cs, float* dxg,
                int* flags, int* sticky_err, int B, int T, int H, void* stream);

/* nn.MaxPool2d((1,2)) over the frequency axis of an NHWC tensor x [rows, W, C] -> y [rows, W/2, C] fused with the
 * nn.Dropout(p) that follows it in ConvStack (model/onset_frame_VAT.py:336-343; p = 0 disables the drop).  code
 * [rows, W/2, C] bytes: bit 0 = the odd column was the max, bit 1 = kept.  Kept values are scaled by 1/(1-p).  The mask is a
 * counter hash of (seed, *epoch, index); epoch (nullable) is a device counter bumped per training step, which keeps the
 * draws fresh when the launch is replayed from a hipGraph. */
int rv_maxpool_w2_dropout_fwd(const float* x, float* y, unsigned char* code, long rows, int W, int C, float p, unsigned seed,
                              const long* epoch, void* stream);
int rv_maxpool_w2_dropout_bwd(const float* dy, const unsigned char* code, float* dx, long rows, int W, int C, float p,
                              void* stream);
/* nn.Dropout(p) (model/onset_frame_VAT.py:346-348).  Forward: code_in NULL, code_out receives the keep mask.  Backward:
 * pass the saved mask as code_in (seed unused) and dy as x. */
int rv_dropout(const float* x, float* y, unsigned char* code_out, const unsigned char* code_in, long n, float p, unsigned seed,
               const long* epoch, void* stream);

#ifdef __cplusplus
}
#endif
#endif
