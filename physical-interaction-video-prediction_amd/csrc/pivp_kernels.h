// Internal launcher interface between the C-ABI layer (pivp_c_api.hip) and the gfx950 kernels.
// Internal data layout (DESIGN.md "Data layout in HBM"):
//   * feature maps: NHWC fp32, `ld` = floats between consecutive pixels (lets a producer write
//     straight into a channel slice of a skip-concat buffer; TM:569-576 never materialises a copy)
//   * frames (prev image, generated image, mask logits, enc7): planar [B][planes][H*W] fp32, i.e.
//     the reference's NCHW, because the flat-11 mask softmax (TM:720-722) and the per-plane 5x5 CDNA
//     transform (TM:341) are defined on that order
//   * conv / deconv weights of the MFMA kernels: [tap][Cin/32][Cout][32] (K-inner packed, so a K chunk of a
//     column is one 128-B line); enc0 / enc3 / 1x1 heads: [tap][Cin][Cout]; Linear weights K-major;
//     LN gamma/beta NHWC-flat.
#pragma once
#include "pivp_common.h"

namespace pivp {

struct IgemmDesc {
    const float* x0; const float* x1;   // input sources (channel-concatenated: x0 then x1)
    int c0, ld0, c1, ld1;
    int wcin;                            // Cin the weight was packed with (>= c0+c1: trailing sources may be skipped)
    const float* w;                      // [wtaps][(c0+c1)/32][N][32]  (K-inner packed)
    const float* bias;                   // [N] or null
    int B, Hin, Win;                     // input feature-map size
    int Hg, Wg, in_step;                 // anchor grid; input coord = anchor*in_step + (dy,dx)
    int N, M;                            // output columns; M = B*Hg*Wg
    int wN;                              // columns the weight was packed with (0: N): N < wN computes the first N columns only
    int nphase;                          // 1 (conv) or 4 (sub-pixel phases of the stride-2 transposed conv)
    int deconv, ksize, pad;              // tap set: conv ksize x ksize with `pad`, or transposed 3x3 s2 p1
    int bytes0, bytes1, bytesw;          // extents of x0 / x1 / w for the buffer descriptors (< 2^31)
    int out_step, Hout, Wout;            // output coord = anchor*out_step + phase parity
    float* out; int ldo; int relu;
    int accum;                           // 1: out += result (gradient accumulation)
    int ksplit_ok;                       // 1: `out` is pre-zeroed and may be produced by K-split blocks with atomic adds
    // ConvLSTM epilogue
    const float* cstate_in; float* cstate_out; float* hout; int C;
    float* gates_out;                    // optional [M][4C]: tanh(j), sigma(i), sigma(f+1), sigma(o) for the backward pass
    // LayerNorm statistics of the OUTPUT, fused into the epilogue (the norm that follows every ConvLSTM and enc6, TM:595-601):
    // each block writes (count, mean, M2) of its tile to ln_part[(b * ln_nparts + slot) * 4]; ln_apply merges them.
    // The launcher fills ln_nparts (0 = tiles straddle samples or exceed ln_cap: not fused, run ln_stats instead).
    float* ln_part; int ln_cap; int ln_nparts;
    int bf16;                            // transposed conv only, when the tile kernel takes the call: 1 = bf16 operands, 2 = split (two bf16 pieces),
                                         // 3 = two FP16 pieces (fp32-grade), the weights times the power of two of wscale_part
    const float* wscale_part;            // bf16 == 3: absmax_partials(w) (64 partial maxima at [2..65]); conv5x5_bf16 with planes = -2: absmax_partials(x0),
                                         // the ACTIVATIONS' scale (the weights' one travels in the pack's tail)
    // LayerNorm of the INPUT applied while it is staged (inference rollouts: the norm's own launch disappears).  x0 then is the RAW tensor
    // [B][Hin*Win][c0] (all c0 channels are normalised), in_g / in_b the norm's per-element gamma / beta ([Hin*Win][c0], the checkpoint's
    // flat order), in_part the producer's (count, mean, M2) partials [B][in_np][4].  Served by igemm_small only (igemm_in_ln_ok).
    const float* in_g; const float* in_b; const float* in_part; int in_np; float in_eps;
    // deconv_tile only (training plans): also WRITE the normalised x0 ([B][Hin*Win] pixels at stride in_out_ld; each pixel by the one block
    // that owns it) and the samples' (mean, rstd) ([B][2]) -- what ln_apply would have left behind for the backward sweep
    float* in_out; int in_out_ld; float* in_stat_out;
    // Plain 5x5 bf16 / split-precision convolution only (the ConvLSTM data gradient), unsplit grids only (no atomics): a second tensor met in the
    // epilogue, so that the pass that would follow the launch disappears.  ep_mode 1: ReLU mask -- out = ep_src > 0 ? out : 0 (the x columns of a
    // cell's input gradient are the dY of the enc conv that produced x: relu_mask_kernel's job); 2: out += ep_src (a second gradient path into the
    // same tensor: add_strided_kernel's job).  Applies to output columns < ep_cols; ep_src[pixel * ep_ld + column].
    const float* ep_src; int ep_ld, ep_cols, ep_mode;
    // deconv_tile only: "rider" blocks behind the launch's own tiles (round 6).  rd_mode 1 / 2: the last rd_blocks (= B) blocks of the grid run the CDNA /
    // STP finisher of the motion head (cdna_finish_block / stp_finish_block, skinny_linear.h) for one sample each and return -- per-sample work that
    // frame_head otherwise repeats in each of its 16 bands per sample (128 KB of partial sums per block), put where the chip has idle CUs (enc5's grid
    // is 192 tiles on 256 CUs at B = 32).  The partial sums must be complete before the launch (they are: the Linear runs in front of lstm6).
    // igemm_f32 only, filled by its launcher (round 6): exact division by a run-time divisor as multiply-high + add + shift (pivp_fastdiv): the tile
    // column count n_mblk, the anchors per sample Hg * Wg and the anchor row Wg -- integer division is a ~35-instruction sequence, and a block's
    // prologue is priced by its instruction count
    int n_mblk; unsigned fd_mb_mul, fd_mb_sh, fd_hw_mul, fd_hw_sh, fd_w_mul, fd_w_sh, fd_cc_mul, fd_cc_sh;   // (.., and the 32-channel chunks per tap)
    // igemm_small only (inference plans, round 6): group 3 of the op program (TM:598: smear(state_action) -> concat -> 1x1 conv 74 -> 64 -> ReLU) and the
    // state predictor (TM:730) run in the epilogue of the conv that produces their input (enc2, TM:502), on the block's own 32 pixels x 64 channels:
    // f3_out [M][64] = relu(b3 + W3s . sa + W3x . out_tile); the launch that enc3_state_kernel was (6.3 us + a boundary per timestep) disappears.
    // Needs N == 64 in one column block and 32-anchor tiles inside one sample; `out` is still written.
    const float* f3_w; const float* f3_b; const float* f3_action; const float* f3_state; const float* f3_wcs; const float* f3_bcs;
    float* f3_out; float* f3_state_out; int f3_use_state;
    int rd_mode, rd_blocks, rd_KS, rd_nout;
    const float* rd_partials; const float* rd_bias; const float* rd_w2; const float* rd_b2; float* rd_out; float* rd_vpre;
};

// weight gradient of a conv / transposed conv (csrc/igemm_wgrad.hip)
// Where the WEIGHT-gradient half of a conv backward runs.  A weight gradient feeds nothing but the optimizer, while the data gradient
// is on the backward sweep's critical path: with a fork the weight (and bias) gradient kernels are enqueued on `side` behind `ready`
// (recorded on the main stream as soon as dY is final) and `done` is recorded behind them; the caller makes whoever next overwrites
// dY, or reads dW, wait for `done`.
struct SideFork {
    hipStream_t side;
    hipEvent_t ready, done;
};

struct WgradDesc {
    const float* x0; const float* x1;    // forward input sources (channel-concatenated), NHWC
    int c0, ld0, c1, ld1, cin, wcin;     // cin = c0 + c1 channels differentiated; wcin = Cin of the packed weight
    const float* dy; int ldy, N;         // output gradient NHWC (N columns, pixel stride ldy)
    float* dw;                           // packed gradient [tap][wcin/32][N][32], accumulated with atomics
    int B, Hx, Wx, Hy, Wy;               // input / output feature-map sizes
    int Hg, Wg, M;                       // anchor grid (conv: output pixels, deconv: input pixels), M = B*Hg*Wg
    int deconv, ksize, pad, stride;
    int bytes0, bytes1, bytesy;
    float* db;                           // optional bias gradient [N] (column sums of dy), accumulated with atomics when the
                                         // kernel that runs can do it on the side (*bias_done = 1), else left to bias_grad
    // A batch of timesteps in one launch (the reduction of a weight gradient runs over pixels AND timesteps): operand j = 0..tcount-1
    // lives at x0 + j*ts_x0, x1 + j*ts_x1, dy + j*ts_dy (byte strides, may be negative); M, B, bytes* describe ONE timestep.
    // 0 / 1 = a single timestep.  Fewer, longer launches: one block epilogue (LDS reduction + atomics) per batch instead of per step.
    int tcount;
    long long ts_x0, ts_x1, ts_dy;
    // Generic kernel only: per-block partial sums instead of atomics.  Every (tile, pixel-split) block owns one [64][64 or 128] slot of
    // `part` (igemm_wgrad_part_floats(d) floats, zeroed by the caller before the first launch) and adds its tile into it with plain
    // loads and stores; igemm_wgrad_reduce(d) then sums the splits into dw.  The scattered atomics of the direct path cost 54 of the
    // 85 us of an enc5 / enc6 launch (64 adds per address from 500 blocks).  The launches that share a slot must be stream-ordered.
    float* part;
    int part_overwrite;                  // 1: this launch is the slot's first since the caller last consumed it: store instead of add (no zeroing needed)
    // wgrad5x5_bf16 only: two FP16 pieces per operand, three MFMAs per product (the fp16x3 mode's weight gradient).  dy_absmax = the absmax_partials tail of
    // timestep j's dy at dy_absmax + j * dy_absmax_stride floats (64 partial maxima at [2..65]): dy is staged times the power of two that puts the largest
    // |value| of the whole batch into [2^14, 2^15) -- gradients lie far below fp16's normal range -- and the sums are scaled back exactly.  null: bf16 operands.
    const float* dy_absmax; int dy_absmax_stride;
    int pieces;                          // wgrad5x5_bf16 only: 3 = three bf16 pieces per operand, six MFMAs per product (the bf16x6 mode's weight gradient)
    int form;                            // wgrad5x5_bf16, plain bf16 operands, a batch of timesteps: 0 = by size (four-wave blocks that co-reside with the main stream's
                                         // kernels for maps of up to 32 x 32 x 32 pixels per timestep, else eight-wave blocks), 1 = four-wave, 2 = eight-wave
};
int igemm_wgrad(const WgradDesc& d, hipStream_t s, int* bias_done = nullptr);
long long igemm_wgrad_part_floats(const WgradDesc& d);    // 0 when the ConvLSTM fast path would take this descriptor
int igemm_wgrad_reduce(const WgradDesc& d, hipStream_t s);   // dw += sum over the splits of d.part
// the stride-2 3x3 convs / transposed convs with all nine taps from one staging of the operands (csrc/wgrad3x3s2.hip): partial-sum path only;
// igemm_wgrad / igemm_wgrad_part_floats / igemm_wgrad_reduce route to it when wgrad3x3s2_ok(d) and d.part is given
bool wgrad3x3s2_ok(const WgradDesc& d);
long long wgrad3x3s2_part_floats(const WgradDesc& d);
int wgrad3x3s2(const WgradDesc& d, hipStream_t s);
int wgrad3x3s2_reduce(const WgradDesc& d, hipStream_t s);
// the fp32 ConvLSTM weight gradient with LDS-DMA staging, an XCD-aware balanced partition and per-segment partial slots (csrc/wgrad5x5p.hip): partial-sum
// path only; igemm_wgrad / igemm_wgrad_part_floats / igemm_wgrad_reduce route to it when wgrad5x5p_ok(d) and d.part is given (db: column sums in the slots)
bool wgrad5x5p_ok(const WgradDesc& d);
long long wgrad5x5p_part_floats(const WgradDesc& d);
int wgrad5x5p(const WgradDesc& d, hipStream_t s);
int wgrad5x5p_reduce(const WgradDesc& d, hipStream_t s);
int wgrad5x5p_partition(const WgradDesc& d, int* geom8, int* segs, int seg_cap, int* nsegs, int* slots, int slot_cap, int* nslots);   // host walk of the partition (tests)
// bf16-operand form of the ConvLSTM weight gradient (csrc/wgrad_bf16.hip); the bias gradient is left to bias_grad
bool wgrad5x5_bf16_ok(const WgradDesc& d);
int wgrad5x5_bf16(const WgradDesc& d, hipStream_t s);
int repack_transpose(const float* w, float* wt, int taps, int cin, int N, int flip, hipStream_t s);
// Several weight preparations in ONE launch (the weights change once per optimizer step, so every train step rebuilds its packs: 7 + 12 + 7 launches
// of a few microseconds each in the bf16 mode before round 5).  kind 0: repack_transpose(src, dst, taps = p0, cin = p1, N = p2, flip = p3);
// kind 1: pack_lstm_bf16(src, dst, wcin = p0, N = p1, Np = p2) with one bf16 plane.  Jobs of one call must not depend on each other.
struct WeightPrepJob { int kind; const float* src; void* dst; int p0, p1, p2, p3; };
constexpr int WEIGHT_PREP_MAX = 16;
int weight_prep_batch(const WeightPrepJob* jobs, int n, hipStream_t s);

// ln_nparts (optional): receives the number of LayerNorm partials per sample the launch writes to d.ln_part (0: none)
int igemm_lstm(const IgemmDesc& d, hipStream_t stream, int variant = 0, int* ln_nparts = nullptr);  // 0 auto, 1: 4x1 waves, 2: 2x2, 3: 1x4
int igemm_conv(const IgemmDesc& d, hipStream_t stream, int* ln_nparts = nullptr);
int igemm_conv_ksplit(const IgemmDesc& d);   // the K split igemm_conv will use for d (> 1: atomics into a destination the caller must zero)
int igemm_validate(const IgemmDesc& d, bool lstm);   // argument checks shared by the igemm launchers (igemm_f32.hip)
int igemm_small(const IgemmDesc& d, hipStream_t stream, int* ln_nparts = nullptr);
bool igemm_in_ln_ok(const IgemmDesc& d);   // can igemm_small apply d.in_g's LayerNorm while staging x0?
// transposed 3x3 s2 conv, all four output parities per block (csrc/deconv_tile.hip); d validated by igemm_validate
bool deconv_tile_ok(const IgemmDesc& d);
int deconv_tile(const IgemmDesc& d, hipStream_t stream, int* ln_nparts = nullptr, int prec = 0);   // 0 fp32, 1 bf16 operands, 2 split (2 bf16 pieces), 3 two fp16 pieces
int absmax_partials(const float* w, long n, float* tail, hipStream_t stream);   // 64 partial maxima of |w| into tail[2..65] (tail: 66 floats)
// bf16-operand ConvLSTM (csrc/convlstm_bf16.hip): wb = pack_lstm_bf16 of d.w; nch: 0 auto, 16 / 32 channels per block
size_t lstm_bf16_weight_elems(int wcin, int N);
int pack_lstm_bf16(const float* w, unsigned short* wb, int wcin, int N, hipStream_t s, int Np = 0, int planes = 1, int plain = 0);   // plain: planes = 3 only (a plain conv's columns)
int conv5x5_bf16_rows(int N);
int conv5x5_bf16_ksplit(const IgemmDesc& d, int planes = 1);   // > 1: the launch will split K and needs d.out zeroed
int conv5x5_bf16(const IgemmDesc& d, const unsigned short* wb, hipStream_t stream, int planes = 1);   // plain 5x5 s1 conv (ConvLSTM data gradient)
bool convlstm_bf16_ok(const IgemmDesc& d);
bool convlstm_bf16x6_ok(const IgemmDesc& d);   // the three-piece form: 16-wide tiles only
int convlstm_bf16(const IgemmDesc& d, const unsigned short* wb, hipStream_t stream, int* ln_nparts = nullptr, int nch = 0, int planes = 1);

// enc0: 5x5 stride-2 pad-2 conv on a planar 3-channel frame -> NHWC 32 channels (TM:500)
// ln_part (optional): the launch also writes *ln_nparts LayerNorm partials per sample of its output (0 = not supported for the shape)
int conv_enc0(const float* img, const float* w, const float* bias, float* out, int B, int H, int W, hipStream_t s,
              float* ln_part = nullptr, int ln_cap = 0, int* ln_nparts = nullptr);

// LayerNorm over the flattened C*H*W vector of each sample with per-element gamma/beta (TM:203-208)
int ln_stats_slices(int n);  // number of partial slices per sample for n elements
int ln_stats(const float* x, float* partials, int B, int n, hipStream_t s);
int ln_apply(const float* x, const float* partials, const float* gamma, const float* beta, float* out,
             int B, int n, int C, int ldo, float eps, int relu, hipStream_t s, float* stat_out = nullptr,
             int nparts = 0);   // nparts > 0: `partials` holds that many producer-written partials per sample

// enc3: smear(action,state) + 1x1 conv + ReLU (TM:556-567, TM:503) and the state predictor (TM:730)
int enc3_state(const float* e2, const float* action, const float* state, const float* w3, const float* b3,
               const float* wcs, const float* bcs, float* e3, float* state_out,
               int B, int HW8, int use_state, hipStream_t s);

// 1x1 heads on enc6: mask logits (relu) and enc7 (TM:288/315-317, TM:429/454-455, TM:364/387-388, TM:718-719)
// enc7_mode: 0 = CDNA (relu; layer0 = sigmoid), 1 = STP (no relu; layer0 = sigmoid), 2 = DNA (relu; no layer0)
int heads_1x1(const float* e6, const float* wm, const float* bm, const float* we, const float* be,
              float* mask_logits, float* enc7, float* layer0, int B, int HW, int nmask_planes, int nenc7,
              int enc7_mode, hipStream_t s,
              // optional fused relu(LayerNorm(e6)) input stage: e6 is then the raw enc6 map, ln_part its ln_nparts
              // (count, mean, M2) partials per sample; y_out (optional) receives the normalised map, stat_out [B][2]
              const float* ln_part = nullptr, int ln_nparts = 0, const float* gamma = nullptr, const float* beta = nullptr,
              float eps = 0.f, float* y_out = nullptr, float* stat_out = nullptr);

// CDNA kernel generator: Linear(hidden5) -> relu shift -> per-kernel normalisation (TM:321-329)
int cdna_kernel_partials_slices(int K);
int cdna_kernels(const float* hidden5, const float* wt, const float* bias, float* partials, float* kerns,
                 int B, int K, int num_masks, hipStream_t s, float* vpre = nullptr);   // vpre [B][256]: pre-activation, kept for backward

// STP parameters: Linear -> relu -> shared Linear(6) + identity (TM:457-468)
int stp_params(const float* hidden5, const float* wt1, const float* b1, const float* w2, const float* b2,
               float* partials, float* theta, int B, int K, hipStream_t s, float* s1_out = nullptr);   // s1_out [B][256] kept for backward

// flat-11 softmax + transform + compositing -> next frame (TM:720-728 with TM:341-349 / TM:469-470 / TM:392-415)
// mode 0 = CDNA (kerns [B][num_masks][25]), 1 = STP (theta [B][6]), 2 = DNA (enc7 planes [B][25][HW])
int composite(const float* prev, const float* mask_logits, const float* layer0, const float* aux,
              float* out, float* masks_out, int B, int H, int W, int num_masks, int mode, int stp_zero_border,
              hipStream_t s);

// One launch for the output side of a timestep (csrc/frame_head.hip): norm_enc6 + ReLU + the 1x1 heads + the motion head's finisher +
// flat softmax + transform + compositing; bit-identical to heads_1x1 + cdna_kernels / stp_params + composite.
struct FrameHeadArgs {
    const float* e6raw; const float* ln_part; int ln_nparts; const float* gamma; const float* beta; float eps;
    const float* wm; const float* bm; const float* we; const float* be;
    const float* prev;
    const float* partials; int KS; const float* hbias;   // K-slice partial sums [B][KS][256] of the motion head's Linear + its bias; null: `aux` is final
    const float* w2; const float* b2;                    // STP: identity_params (6,100), (6)
    const float* aux;                                    // partials == null: kernels [B][NM][25] (CDNA) / theta [B][6] (STP)
    float* out; float* masks_out;
    float* enc7;
    float* logits_out; float* layer0_out; float* y_out; float* stat_out;   // optional (training keeps them)
    float* kerns_out; float* vpre_out;                   // optional: finished kernels / theta, and the Linear's pre-activation [B][256]
    int B, H, W, NM, stp_zero;
};
bool frame_head_ok(int mode, int B, int H, int W, int num_masks);   // can the fused launch serve this geometry?
bool frame_head_pays(int mode, int B, int H, int W, int num_masks); // ... and does it beat the separate kernels there (two blocks per CU)?
long long motion_partials_floats(int B, int K);                     // floats of the partial-sum buffer (tail padding included)
bool frame_head_finishes(int K);                                    // does its in-kernel finisher take a Linear over K inputs?
int frame_head(const FrameHeadArgs& a, int mode, hipStream_t s);
int motion_partials(const float* hidden5, const float* wt, float* partials, int B, int K, int dbl, hipStream_t s);   // skinny Linear, partial sums only

// loss / PSNR (TM:737-759)
int loss_partials_count(int n);
int frame_sqerr_partials(const float* a, const float* b, float* partials, int n, hipStream_t s);
int loss_finalize(const float* frame_partials, int nparts, int nframes, int frame_numel,
                  const float* states_true, const float* states_gen, int state_numel,
                  float denom, float* results, hipStream_t s);

// ---- backward (csrc/backward.hip) ----
// LayerNorm backward folded into the ConvLSTM gate backward: the cell's dh is the dx of the norm behind it (see lstm_gates_bwd_kernel)
struct LnFuse {
    const float* dy; int lddy;       // gradient arriving at the norm's OUTPUT (may be a channel slice of a concat buffer)
    const float* gamma;              // [n] NHWC-flat
    const float* stat;               // [B][2] mean, rstd of the forward pass
    const float* partials; int S;    // [B][S][2] sums of (g, g * xhat), g = dy * gamma, over the S = ln_bwd_slices(n) slices of a sample
                                     // (ln_bwd_sums_params_kernel, which also forms the norm's parameter gradients)
    const float* h;                  // the norm's input = the cell's h_t, [M][C]
};
int lstm_gates_bwd(const float* gates, const float* c_old, const float* c_new, const float* dh_a, int lda,
                   const float* dh_b, int ldb, float* dc, int dc_valid, float* dG, int M, int C, hipStream_t s, int B = 1,
                   const LnFuse* ln = nullptr,
                   float* zero = nullptr, long long zero_floats = 0);
int bias_grad(const float* dy, int ld, int N, int M, float* db, hipStream_t s);
int relu_mask(float* dy, int lddy, const float* y, int ldy, int C, long npix, hipStream_t s,
              const float* add = nullptr, int ldadd = 0);   // add: dy = (dy + add) masked -- a second gradient path into the same activation
int ln_bwd_slices(int n);
// param_part (optional, ln_bwd_param_part_floats(n) floats, zeroed before the first launch of a sweep): the parameter gradients are
// accumulated there without atomics and reach dgamma / dbeta only through ln_bwd_params_reduce (once per sweep)
int ln_backward(const float* dy, int lddy, const float* y, int ldy, const float* x, const float* stat, const float* gamma,
                float* partials, float* dx, float* dgamma, float* dbeta, int B, int n, int C, int relu, hipStream_t s,
                float* param_part = nullptr);
long long ln_bwd_param_part_floats(int n);
int ln_bwd_params_reduce(const float* part, float* dgamma, float* dbeta, int B, int n, hipStream_t s);
int adam_step(float* p, const float* g, float* m, float* v, long n, double lr_t, double beta1, double beta2, double eps,
              double gscale, hipStream_t s);
int grad_pack_bf16(const float* src, void* dst, long n, hipStream_t s);     // fp32 -> bf16 (RNE): the all-reduce payload of config 3
int grad_unpack_bf16(const void* src, float* dst, long n, hipStream_t s);   // bf16 -> fp32, into the flat gradient buffer
int grad_sum_shards(const void* src, int src_bf16, int nsh, long len, void* dst, int dst_bf16, hipStream_t s);   // sum of nsh peer shards in fp32, rounded once

// ---- backward of the heads / trunk ends (csrc/backward_heads.hip) ----
int scaled_diff(const float* a, const float* b, float* out, long n, float scale, int accum, hipStream_t s);
int composite_bwd_tiles(int H, int W);
int composite_bwd_cdna(const float* prev, const float* logits, const float* layer0, const float* kerns, const float* go,
                       float* dmk, float* dz, float* dkpart, float* dprev, int dprev_accum, int B, int H, int W, int NM, hipStream_t s);
int composite_bwd_stp(const float* prev, const float* logits, const float* layer0, const float* theta, const float* go,
                      float* dmk, float* dz, float* dthpart, float* dprev, int B, int H, int W, int NM, int stp_zero, hipStream_t s);
int stp_params_bwd(const float* hidden5, const float* wt1, const float* s1, const float* w2, const float* dthpart, int ntiles, float* dv,
                   float* dhidden5, float* dwt1, float* db1, float* dw2, float* db2, int B, int K, hipStream_t s);
int composite_bwd_dna(const float* prev, const float* logits, const float* e7, const float* go, float* dmk, float* dz,
                      float* dprev, int dprev_accum, int B, int H, int W, hipStream_t s);
int mask_softmax_bwd(const float* logits, float* dmk, int B, int HW, int NP, hipStream_t s);
int heads_bwd(const float* e6, const float* wm, const float* we, const float* dpm, const float* dpe, float* de6,
              float* dwm, float* dbm, float* dwe, float* dbe, int B, int HW, int NP, int NE, hipStream_t s);
int cdna_kernels_bwd(const float* hidden5, const float* wt, const float* vpre, const float* dkpart, int ntiles, float* dv,
                     float* dhidden5, int accum_dx, float* dwt, float* db, int B, int K, int NM, hipStream_t s, const SideFork* fork = nullptr);   // fork: where the WEIGHT-gradient kernel runs (behind dv)
int enc3_state_bwd(const float* e2, const float* e3, const float* de3, int ldd3, const float* action, const float* state, const float* w3,
                   const float* wcs, const float* dsnew, float* de2, float* dw3, float* db3, float* dwcs, float* dbcs,
                   float* dstate_prev, int B, int HW8, int use_state, hipStream_t s,
                   int mask_e2 = 0);   // 1: de2 is masked by (e2 > 0), i.e. it is the gradient in front of enc2's ReLU
int enc0_bwd(const float* img, const float* w, const float* d, float* dw, float* db, float* dimg, int dimg_accum, int B, int H, int W,
             hipStream_t s, const SideFork* fork = nullptr);   // fork: where the weight / bias gradient kernel runs
int add_strided(float* dst, int ldd, const float* src, int lds_, int C, long npix, hipStream_t s);

// planar NCHW <-> NHWC helpers for taps (conv_res) and tests
int resize_bilinear(const float* in, float* out, int planes, int Hin, int Win, int Hout, int Wout, float scale, hipStream_t s);
int nhwc_to_nchw(const float* in, float* out, int B, int C, int HW, int ld, hipStream_t s);

}  // namespace pivp
