/* pivp_hip.h -- C ABI of the MI355X (gfx950) ConvLSTM + CDNA/STP/DNA video-prediction hot path.
 *
 * Drop-in scope: the reference (kristofbc/physical-interaction-video-prediction) has no FFI or
 * plugin interface; its hot path is the Python method Model.__call__ (src/models/train_model.py,
 * "TM", lines 620-764) built on Chainer ops.  This library replaces the arithmetic under that
 * method.  Every entry point cites the reference code it stands in for.  Signatures carry plain
 * device pointers, sizes and a hipStream_t passed as void*; no torch / C++ types.
 *
 * Conventions
 *   - all pointers are DEVICE pointers to fp32 unless stated; the caller owns every buffer
 *   - return value: 0 = ok, <0 = error (PIVP_ERR_*); functions never throw and never synchronise
 *   - re-entrant per plan + stream.  Process-global state is limited to (a) write-once-per-DEVICE caches of kernel attributes and the
 *     CU count (csrc/pivp_common.h: a process may drive several devices) and (b) PIVP_* tuning knobs read once with getenv(); both
 *     are idempotent.  A plan that runs a backward sweep owns one internal low-priority stream and its events for the weight
 *     gradients (created on the first pivp_rollout_backward, destroyed with the plan); everything it enqueues there is fenced
 *     against the caller's stream with events, so the caller still only ever synchronises its own stream.  PIVP_SIDE_STREAM=0
 *     (read at pivp_plan_create) keeps all work on the caller's stream.  PIVP_FINISH_RIDER=0 (read there too) runs the motion head's finisher
 *     inside the frame-head launch instead of as extra blocks of enc5's launch: bit-identical results either way, it exists for A/B timing.
 *     PIVP_FUSE_ENC3=0 likewise keeps group 3 (smear + 1x1 conv) and the state predictor in a launch of their own instead of enc2's epilogue.
 *     PIVP_LN_FOLD_TRAIN=0: training plans apply the norms of hidden2 / hidden4 with ln_apply launches instead of inside enc1 / enc2's launches.
 *   - feature maps are NHWC with an explicit pixel stride `ld` (floats); frames and mask planes are
 *     planar NCHW exactly as the reference holds them
 */
#ifndef PIVP_HIP_H
#define PIVP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define PIVP_OK 0
#define PIVP_ERR_BADARG (-1)
#define PIVP_ERR_LAUNCH (-2)
#define PIVP_ERR_STATE (-3)

#define PIVP_MODEL_CDNA 0
#define PIVP_MODEL_STP 1
#define PIVP_MODEL_DNA 2

#define PIVP_PRECISION_F32 0
#define PIVP_PRECISION_BF16 1
#define PIVP_PRECISION_BF16X3 2
#define PIVP_PRECISION_BF16X6 3
#define PIVP_PRECISION_FP16X3 4

int pivp_abi_version(void);   /* 17 (17: + pivp_wgrad5x5_f32_part_floats, pivp_wgrad5x5_f32_batch, pivp_wgrad5x5_f32_reduce, pivp_wgrad5x5_f32_partition, pivp_conv5x5_f32; 16: + pivp_conv_wgrad_partial_batch, pivp_conv_wgrad_partial_reduce; 15: + pivp_wgrad5x5_bf16_batch_form; 14: + pivp_build_flags, pivp_plan_set_main_priority; 13: + pivp_wgrad5x5_bf16x6_batch; 12: + pivp_wgrad5x5_fp16x3_batch; 11: + pivp_conv5x5_fp16x3; 10: + PIVP_PRECISION_BF16X6 / _FP16X3, pivp_pack_lstm_bf16x6, pivp_convlstm_bf16x6, pivp_conv5x5_bf16x6, pivp_pack_lstm_fp16x3, pivp_convlstm_fp16x3, pivp_deconv3x3s2_fp16x3, pivp_plan_set_pack_cache, pivp_plan_params_changed; 9: + pivp_build_digest, pivp_grad_sum_shards, pivp_frame_head; 8: + pivp_gates_backward_ln, pivp_deconv3x3s2_ln; 7: + bf16 gradient payload, batched bf16 weight gradient, partial-plane / dx-only op entries; 2: + training entry points, 3: + pivp_convlstm_ln, 4: + gradient groups / callback,
                                 5: + bf16 ConvLSTM, pivp_plan_set_precision, 6: + pivp_plan_set_group_join / pivp_plan_group_wait) */

/* sha256 (hex) of the sources this library was compiled from (every .hip and .h under csrc/, and this header), embedded by build.py.  The Python
 * binding recomputes it from the sources shipped next to the library and refuses to load on a mismatch: a stale build can never stand
 * in for the code under test.  "unstamped" for a build that bypassed build.py. */
const char* pivp_build_digest(void);
/* The compile flags this library was built with beyond build.py's standard set (PIVP_EXTRA_FLAGS: instrumented / timing-only variants); "" for the
 * product build.  A variant has the same source digest as the product: bench.py reports this string, the GPU tests refuse a non-empty one. */
const char* pivp_build_flags(void);

/* ------------------------------------------------------------------------------------------
 * Plan = Model.__init__ (TM:484-602): layer table, op program, variant head.
 * ------------------------------------------------------------------------------------------ */
typedef struct pivp_plan pivp_plan_t;

typedef struct pivp_config {
    int batch;             /* B, sequences per call on this device                               */
    int seq_len;           /* T, frames per sequence incl. context (TM:659 iterates T-1 steps)   */
    int height, width;     /* frame size; multiples of 8 (TM:505-507 hard-codes 64, generalised) */
    int num_masks;         /* TM:484 num_masks (10 for CDNA/STP, 1 for DNA)                      */
    int model_type;        /* PIVP_MODEL_*; precedence cdna > stp > dna is resolved by the host  */
    int use_state;         /* TM:556-567 smear of [action,state] before enc3                     */
    int context_frames;    /* TM:484 num_frame_before_prediction                                 */
    int keep_activations;  /* 1: keep every timestep's activations (training/BPTT); 0: rolling   */
    float ln_eps;          /* chainer.links.LayerNormalization eps (1e-6 in 2.0.x)               */
    int stp_zero_border;   /* 0: clamp sampling coords (Chainer 2.0.x), 1: zero outside          */
} pivp_config_t;

int pivp_plan_create(const pivp_config_t* cfg, pivp_plan_t** out);
void pivp_plan_destroy(pivp_plan_t* plan);

/* Parameters, in the library's internal layout (the host permutes the reference's
 * Chainer-npz arrays, keys = SURVEY.md App. B, when loading a checkpoint). */
int pivp_param_count(const pivp_plan_t* plan);
const char* pivp_param_name(const pivp_plan_t* plan, int idx);      /* Chainer save_npz key        */
long long pivp_param_numel(const pivp_plan_t* plan, int idx);       /* elements in internal layout */
int pivp_plan_set_param(pivp_plan_t* plan, int idx, const float* dptr);

/* Precision of the seven ConvLSTM gate convolutions and of the enc5 / enc6 transposed convs (BASELINE.json config 3 asks for bf16): PIVP_PRECISION_F32 (default, the
 * parity path) or PIVP_PRECISION_BF16 = x, h and the weights rounded to bf16 on the way into the matrix pipe, fp32 accumulation,
 * gates and state; in the backward pass the ConvLSTM data and weight gradients likewise (operands rounded to bf16, fp32 accumulation
 * into the fp32 gradients).  All other ops stay fp32, as do the parameters and Adam (the bf16 weight packs are rebuilt at the start of
 * every rollout / backward sweep).
 * PIVP_PRECISION_BF16X3 = the split mode: only the FORWARD gate convolutions and the enc5 / enc6 transposed convs change -- each fp32 operand travels as two bf16 numbers (hi, lo)
 * and a product is three bf16 MFMAs, 16 bits of product mantissa, fp32 accumulation -- and the result stays inside the 1e-4 per-pixel gate
 * (4.4e-5 on the config 1 rollout; tests/test_gpu_bf16.py); the backward pass and every other op are the fp32 ones.
 * PIVP_PRECISION_BF16X6 = three bf16 pieces per fp32 operand (hi + mid + lo = v exactly) and the six products of weight >= 2^-16, i.e. fp32-grade
 * gate pre-activations computed on the bf16 matrix cores: the gate convolutions and, in the backward sweep, their DATA gradients (maps a multiple of 16 wide,
 * or -- since round 5 -- 8 wide with an even batch: lstm5 on 64 x 64 frames; anything else runs the fp32 kernels), and their WEIGHT gradients (two timesteps per
 * launch); every other op is the fp32 one.
 * PIVP_PRECISION_FP16X3 = the forward gate convolutions (and the enc5 / enc6 transposed convs) with every fp32 operand as TWO FP16 pieces (22 bits of mantissa; a layer's weights are packed times the
 * power of two that puts the largest in [2^14, 2^15), so that the second piece of any weight down to 2^-18 of it stays a normal fp16 number; the sum is scaled
 * back exactly) and three MFMAs per product; activations beyond +-65504 saturate, activations below 0.06 carry up to 3e-8 of absolute error.  Its truncation error is a quarter of the fp32 path's own rounding error (scripts/split_fp16_study.py).  The backward sweep's
 * ConvLSTM data and weight gradients take the same form: gradients do not fit fp16's exponent range as they are, so dG is staged times the power of two that puts
 * its largest |value| (per cell and timestep / batch of timesteps: one absmax launch) into [2^14, 2^15) and the sums are scaled back exactly; an 8-wide map with
 * an odd batch keeps the fp32 kernels.  PIVP_ERR_BADARG when a layer's map does not fit the bf16 kernel (8-wide maps need an even batch).
 * ORDER: pivp_plan_set_precision -> pivp_plan_workspace_bytes -> pivp_plan_set_workspace.  The workspace size depends on the precision (the modes that batch
 * their ConvLSTM weight gradients keep deeper gate-gradient rings), so pivp_plan_workspace_bytes changes with this call while no workspace is bound.  With a
 * workspace bound, a mode that needs deeper rings than the bound layout holds is refused with PIVP_ERR_STATE (nothing changes); modes that fit switch in place. */
int pivp_plan_set_precision(pivp_plan_t* plan, int precision);
int pivp_plan_get_precision(const pivp_plan_t* plan);
/* The precision modes re-pack the ConvLSTM weights (bf16 / split pieces) at the start of every rollout, because the parameters may have changed.  With
 * constant weights (serving) pivp_plan_set_pack_cache(plan, 1) keeps the packs; the caller then calls pivp_plan_params_changed after EVERY modification of a
 * parameter tensor.  The Python Model does both by itself (torch's in-place version counter + its own optimizer steps). */
int pivp_plan_set_pack_cache(pivp_plan_t* plan, int on);
int pivp_plan_params_changed(pivp_plan_t* plan);

/* Size of a TRAINING plan's workspace, for orientation (fp32 mode): T - 1 activation slabs + the gradient workspace.  Since round 5 the dY tensors of
 * the five stride-2 3x3 layers are kept as 2 x eg rings of timesteps (one batched weight-gradient launch per ring), eg = min(T - 2, 8) cut so that the
 * rings together stay within 2 GiB: +0.97 GB at B = 32, T = 10 on 64 x 64 frames (eg = 8), +1.9 GB at B = 32, T = 20 on 128 x 128 (eg = 4).  All of it is
 * caller-owned memory: the plan never allocates. */
long long pivp_plan_workspace_bytes(const pivp_plan_t* plan);
int pivp_plan_set_workspace(pivp_plan_t* plan, void* dptr, long long bytes);

/* Model.reset_state (TM:604-618): zero the seven ConvLSTM (c, h) pairs. */
int pivp_reset_state(pivp_plan_t* plan, void* stream);

/* Model.__call__ forward (TM:620-764): T-1 timesteps, loss and PSNR.
 *   images  [T][B][3][H][W], actions [T][B][5], states [T][B][5]   (time-major, as concat_examples TM:51-71)
 *   gt_select [T-1][B] bytes or NULL: 1 = feed the ground-truth frame at that step (scheduled sampling,
 *             TM:73-122; the host draws the indices from NumPy's global RNG exactly as the reference);
 *             NULL = feed-self after the context frames (TM:664-666)
 *   gen_images [T-1][B][3][H][W], gen_states [T-1][B][5]
 *   results [2 + 3*(T-ctx)]: loss, psnr_all, recon_cost[i], psnr[i], state_cost[i]   (TM:739-759)
 */
int pivp_rollout_forward(pivp_plan_t* plan, const float* images, const float* actions, const float* states,
                         const unsigned char* gt_select, float* gen_images, float* gen_states, float* results,
                         void* stream);

/* Wave priority of the caller-stream kernels of pivp_rollout_backward (s_setprio 3 against the side stream's weight-gradient waves: the single-GPU train
 * step gains 1.5 %).  mode -1 (default): on unless a gradient listener is registered (pivp_plan_set_grad_callback: a data-parallel rank, whose collective's
 * waves must not be starved); 0 / 1: off / on.  The switch is one word per device and process, rewritten on the call's stream when the wanted value
 * changes; it never changes results. */
int pivp_plan_set_main_priority(pivp_plan_t* plan, int mode);

/* Backward through time of the last pivp_rollout_forward (same arguments; the plan must have keep_activations = 1):
 * what loss.backward() does inside Chainer's optimizer.update (TM:950).  Gradients are ACCUMULATED into the buffers
 * registered with pivp_plan_set_grad (same internal layouts as the parameters), so the host clears them first
 * (Chainer: model.cleargrads()).  Frames fed by scheduled sampling are detached as in the reference (TM:669-670);
 * in feed-self mode the gradient flows through the generated frames.  CDNA variant only in this round. */
int pivp_plan_set_grad(pivp_plan_t* plan, int idx, float* dptr);
/* Overlapping the data-parallel gradient all-reduce with the backward sweep.  Every parameter is shared by all timesteps, so a
 * gradient is only final once the sweep has passed its layer at t = 0; the layers finish in reverse program order, in
 * PIVP_GRAD_GROUPS groups: 0 heads + motion-parameter head + norm_enc6 + enc6, 1 hidden7 + lstm7, 2 enc5 + hidden6 + lstm6,
 * 3 enc4 + hidden5 + lstm5, 4 enc3 + current_state + enc2 + hidden4/3 + lstm4/3, 5 enc1 + hidden2/1 + lstm2/1 + norm_enc0 + enc0.
 * pivp_param_group tells the host which group a parameter belongs to (so it can lay the flat gradient buffer out group by
 * group); the callback is invoked on the calling thread from inside pivp_rollout_backward right after the last kernel that
 * touches group g has been ENQUEUED on the stream (the host makes its communication stream wait on an event it records
 * there).  cb = NULL removes it. */
#define PIVP_GRAD_GROUPS 6
typedef void (*pivp_grad_group_cb)(void* user, int group);
int pivp_param_group(const pivp_plan_t* plan, int idx);
int pivp_param_group_by_name(const char* name);            /* same mapping, from the checkpoint key alone */
int pivp_plan_set_grad_callback(pivp_plan_t* plan, pivp_grad_group_cb cb, void* user);
/* The weight gradients of a group may still be running on the plan's internal side stream when the group's last kernel has been
 * enqueued on the caller's.  Default (join = 1): before the callback the CALLER'S stream is made to wait for them, so "announced"
 * means "final in stream order on the caller's stream" -- simple, and it stalls the sweep's last timestep by ~0.3 ms per step.
 * join = 0: the caller's stream is not held up; the host must then call pivp_plan_group_wait(plan, g, comm_stream) from inside the
 * callback, which makes ONLY that stream wait for the side stream's work of group g (in addition to the event the host records on
 * the caller's stream).  pivp_rollout_backward still joins everything before it returns, so whatever the caller enqueues after the
 * sweep (the optimizer step) is ordered behind every gradient either way. */
int pivp_plan_set_group_join(pivp_plan_t* plan, int join);
int pivp_plan_group_wait(pivp_plan_t* plan, int group, void* stream);
int pivp_rollout_backward(pivp_plan_t* plan, const float* images, const float* actions, const float* states,
                          const unsigned char* gt_select, const float* gen_images, const float* gen_states, void* stream);

/* Measurement hooks (bench.py): when enabled, every ConvLSTM launch of pivp_rollout_forward is bracketed
 * by hipEvents recorded on the launch stream.  After the caller has synchronised the stream,
 * pivp_plan_profile_read returns, per ConvLSTM layer (lstm1..lstm7): summed milliseconds, launch count
 * and the SUM of the algorithmic flops of those launches, 2*M*4C*25*(Cx+C) each (TM:224 conv inside TM:262-272;
 * the first step after reset_state skips the all-zero h half of K and counts 2*M*4C*25*Cx). */
int pivp_plan_set_profiling(pivp_plan_t* plan, int enable);
int pivp_plan_profile_read(pivp_plan_t* plan, double* ms_per_layer, int* launches_per_layer, double* flops_per_layer);

/* Activation taps of a timestep still held in the workspace, returned planar NCHW like the
 * reference's conv_res / hiddens (TM:703-708, TM:734).  name: enc0..enc7, hidden1..hidden7, masks,
 * cdna_kerns, lstm1_h.. lstm7_h, lstm1_c..lstm7_c.  Returns the number of floats written or <0. */
long long pivp_get_tap(pivp_plan_t* plan, const char* name, int step, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-op entry points (used by the parity tests and usable on their own).
 * ------------------------------------------------------------------------------------------ */

/* BasicConvLSTMCell.__call__ (TM:234-276): gates = conv5x5(concat(x,h)) ; c,h update, forget bias 1.
 * x NHWC (cx channels, stride ldx), h_prev/c NHWC [B][H][W][C]; h_prev may be NULL = all zeros (first step); w [25][(cx+C)/32][4C][32] (K-inner packed), gate order j,i,f,o. */
int pivp_convlstm(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* bias,
                  const float* c_in, float* c_out, float* h_out, int B, int H, int W, void* stream);

/* Same, with the block-tile variant forced (0 auto, 1: BM 128, 2: BM 64, 3: BM 32 as two K groups of 4 waves, 4: BM 64 as two K groups) -- tests and tuning. */
int pivp_convlstm_v(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* bias,
                    const float* c_in, float* c_out, float* h_out, int B, int H, int W, int variant, void* stream);

/* hidden = norm(lstm(x)) as the reference writes it at TM:596-601: the ConvLSTM above followed by its
 * LayerNormalizationConv2D, with the per-sample statistics taken from the ConvLSTM kernel's epilogue (one (count, mean,
 * M2) partial per block tile) instead of a second pass over h.  ln_out NHWC with pixel stride ldo; partials: scratch of
 * pivp_convlstm_ln_scratch_floats(B,H,W,C) floats; *fused (optional) receives 1 when the statistics were fused, 0 when the
 * tile shape made the call fall back to the separate statistics pass (same result either way). */
long long pivp_convlstm_ln_scratch_floats(int B, int H, int W, int C);
int pivp_convlstm_ln(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* bias,
                     const float* c_in, float* c_out, float* h_out, const float* gamma, const float* beta, float* ln_out,
                     int ldo, float* partials, float eps, int B, int H, int W, int variant, int* fused, void* stream);

/* bf16-operand form of pivp_convlstm (BASELINE.json config 3): x, h and the weights are rounded to bf16 (nearest even) on the
 * way into the matrix pipe; accumulation, gates, c and h stay fp32.  w_bf16 = pivp_pack_lstm_bf16(w) holds
 * pivp_lstm_bf16_weight_elems(cx + C, C) 2-byte elements.  gates_out / ln_part / ln_nparts may be NULL (as pivp_convlstm_train /
 * pivp_convlstm_ln: ln_part receives ln_cap-bounded (count, mean, M2, 0) partials per sample, *ln_nparts how many, 0 = none).
 * nch: 0 automatic, 16 or 32 channels per block.  Needs H % 8 == 0 and W % 16 == 0 (or W % 8 == 0 with B even). */
long long pivp_lstm_bf16_weight_elems(int cin_total, int C);
int pivp_pack_lstm_bf16(const float* w, void* w_bf16, int cin_total, int C, void* stream);
int pivp_convlstm_bf16(const float* x, int cx, int ldx, const float* h_prev, int C, const void* w_bf16, const float* bias,
                       const float* c_in, float* c_out, float* h_out, float* gates_out, float* ln_part, int ln_cap,
                       int* ln_nparts, int B, int H, int W, int nch, void* stream);

/* Split form of pivp_convlstm_bf16 (precision mode PIVP_PRECISION_BF16X3): every fp32 operand travels as two bf16 numbers, hi = bf16(v) and
 * lo = bf16(v - hi), and a product is formed on three bf16 MFMAs (lo*hi + hi*lo + hi*hi, fp32 accumulation): 16 bits of product mantissa.
 * w_bf16 = pivp_pack_lstm_bf16x3(w): 2 * pivp_lstm_bf16_weight_elems(cx + C, C) 2-byte elements.  Other arguments as pivp_convlstm_bf16. */
int pivp_pack_lstm_bf16x3(const float* w, void* w_bf16, int cin_total, int C, void* stream);
int pivp_convlstm_bf16x3(const float* x, int cx, int ldx, const float* h_prev, int C, const void* w_bf16, const float* bias,
                         const float* c_in, float* c_out, float* h_out, float* gates_out, float* ln_part, int ln_cap,
                         int* ln_nparts, int B, int H, int W, int nch, void* stream);

/* Three-piece form (precision mode PIVP_PRECISION_BF16X6): hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid); a product is hi*hi on the main
 * accumulator plus lo*hi + hi*lo + mid*mid + mid*hi + hi*mid on a second one (six bf16 MFMAs, fp32 accumulation): what is dropped is below 2^-24 of the
 * product.  C % 16 == 0; W % 16 == 0, or W % 8 == 0 with an even batch (tiles of two images).  w_bf16 = pivp_pack_lstm_bf16x6(w): 3 * pivp_lstm_bf16_weight_elems(cx + C, C) 2-byte elements. */
int pivp_pack_lstm_bf16x6(const float* w, void* w_bf16, int cin_total, int C, void* stream);
int pivp_convlstm_bf16x6(const float* x, int cx, int ldx, const float* h_prev, int C, const void* w_bf16, const float* bias,
                         const float* c_in, float* c_out, float* h_out, float* gates_out, float* ln_part, int ln_cap,
                         int* ln_nparts, int B, int H, int W, int nch, void* stream);   /* nch: 0 automatic; 16 / 32 = 16- / 32-channel blocks
                         (C % 32 == 0 for 32), weights from L2 straight into the operand registers */

/* Two-fp16-piece form (precision mode PIVP_PRECISION_FP16X3): hi = fp16(v), lo = fp16(v - hi); hi*hi on the main accumulator, lo*hi + hi*lo on a second one
 * (three fp16 MFMAs, fp32 accumulation).  C % 16 == 0; W % 16 == 0, or W % 8 == 0 with an even batch (a second kernel and pack layout).  w_bf16 = pivp_pack_lstm_fp16x3(w): 2 * pivp_lstm_bf16_weight_elems(cx + C, C) + 256 2-byte
 * elements (s w as two fp16 pieces, fragment-major; s and the maxima it was taken from in the 512-byte tail).  nch: 0 automatic, 16 / 32 channels per block (tiles of 8 x 16 anchors). */
int pivp_pack_lstm_fp16x3(const float* w, void* w_bf16, int cin_total, int C, int map_width, void* stream);   /* map_width: W of the call it is for (unused since round 5: one layout) */
int pivp_convlstm_fp16x3(const float* x, int cx, int ldx, const float* h_prev, int C, const void* w_bf16, const float* bias,
                         const float* c_in, float* c_out, float* h_out, float* gates_out, float* ln_part, int ln_cap,
                         int* ln_nparts, int B, int H, int W, int nch, void* stream);

/* pivp_deconv3x3s2 with bf16 operands (precision mode bf16): x and w are rounded to bf16 on the way into the matrix pipe, accumulation, bias
 * and ReLU stay fp32.  Only maps with Hin % 8 == 0 and Win % 16 == 0 (and at least 16 tiles x column blocks) run in bf16; the call is the fp32 op otherwise. */
int pivp_deconv3x3s2_bf16(const float* x, int cin, int ldx, const float* w, const float* bias, float* out, int cout,
                          int ldo, int relu, int B, int Hin, int Win, void* stream);
/* ... and in the split mode (two bf16 pieces per operand, three MFMAs per product; PIVP_PRECISION_BF16X3) */
int pivp_deconv3x3s2_bf16x3(const float* x, int cin, int ldx, const float* w, const float* bias, float* out, int cout,
                            int ldo, int relu, int B, int Hin, int Win, void* stream);
/* ... and as two FP16 pieces per operand, three MFMAs per product (fp32-grade; PIVP_PRECISION_FP16X3): the weights' power-of-two scale is taken from their
 * absolute maximum first; scratch: 66 floats */
int pivp_deconv3x3s2_fp16x3(const float* x, int cin, int ldx, const float* w, const float* bias, float* out, int cout,
                            int ldo, int relu, int B, int Hin, int Win, float* scratch, void* stream);

/* Plain 5x5 stride-1 "same" convolution with bf16 operands and fp32 accumulation, out[b,y,x,n] (+)= sum x[b,y+dy,x+dx,k] w[tap][k][n]
 * (the ConvLSTM data gradient of the bf16 mode: x = d gates, w = the flipped transposed weights).  x NHWC (cin channels, stride ldx);
 * w fp32 K-inner packed [25][cin/32][cout][32]; w_bf16 scratch of pivp_conv5x5_bf16_weight_elems(cin, cout) 2-byte elements, rebuilt
 * by the call; out NHWC with pixel stride ldo; accum != 0 adds into out.  Geometry as pivp_convlstm_bf16. */
long long pivp_conv5x5_bf16_weight_elems(int cin, int cout);
int pivp_conv5x5_bf16(const float* x, int cin, int ldx, const float* w, void* w_bf16, float* out, int cout, int ldo, int accum,
                      int B, int H, int W, void* stream);
/* ... in the split mode (two bf16 pieces per operand, three MFMAs per product); w_bf16: twice the elements */
int pivp_conv5x5_bf16x3(const float* x, int cin, int ldx, const float* w, void* w_bf16, float* out, int cout, int ldo, int accum,
                        int B, int H, int W, void* stream);
/* ... and as three pieces per operand, six MFMAs per product (fp32-grade; PIVP_PRECISION_BF16X6's data gradient): W % 16 == 0, or W % 8 == 0 with an even batch; w_bf16 holds
 * 3 * pivp_conv5x5_bf16_weight_elems(cin, cout) 2-byte elements */
int pivp_conv5x5_bf16x6(const float* x, int cin, int ldx, const float* w, void* w_bf16, float* out, int cout, int ldo, int accum,
                        int B, int H, int W, void* stream);
/* ... and as two fp16 pieces per operand, three MFMAs per product (fp32-grade; PIVP_PRECISION_FP16X3's data gradient): x is staged times the power of two
 * that puts its largest |value| into [2^14, 2^15) -- gradients lie far below fp16's normal range --, w as in pivp_pack_lstm_fp16x3, the sums scaled back
 * exactly.  x contiguous (ldx == cin), W % 16 == 0 (or W % 8 == 0 with an even batch); w_bf16 holds 2 * pivp_conv5x5_bf16_weight_elems(cin, cout) + 256 2-byte elements; scratch: 66 floats */
int pivp_conv5x5_fp16x3(const float* x, int cin, int ldx, const float* w, void* w_bf16, float* out, int cout, int ldo, int accum,
                        int B, int H, int W, float* scratch, void* stream);

/* ConvLSTM weight gradient with bf16 operands and fp32 accumulation (bf16 mode): dW[tap][ci][n] += sum_m concat(x, h_prev)[m + tap][ci]
 * dG[m][n]; dW K-inner packed like the weight, ACCUMULATED; h_prev may be NULL (first timestep: only the x rows are touched);
 * dG [B*H*W][4C]; db (optional, [4C]) += the column sums of dG, taken in fp32.  Needs 4C % 128 == 0, cx % 32 == 0, C % 32 == 0 and
 * the map geometry of pivp_convlstm_bf16. */
int pivp_wgrad5x5_bf16(const float* x, int cx, int ldx, const float* h_prev, int C, const float* dG, float* dW, float* db,
                       int B, int H, int W, void* stream);
/* The same for a batch of `tcount` timesteps in one launch (the reduction of a weight gradient runs over pixels AND timesteps: the
 * BPTT sweep of optimizer.update, TM:950, visits every cell T-1 times): timestep j reads x + j*ts_x, h_prev + j*ts_h, dG + j*ts_dG
 * (byte strides, multiples of 16, may be negative). */
int pivp_wgrad5x5_bf16_batch(const float* x, int cx, int ldx, const float* h_prev, int C, const float* dG, float* dW, float* db,
                             int B, int H, int W, int tcount, long long ts_x, long long ts_h, long long ts_dG, void* stream);
/* ... with the block form chosen by the caller: 0 = by size (what pivp_wgrad5x5_bf16_batch and the plan do), 1 = four-wave blocks (one per CU, co-resident with
 * the sweep's small kernels: the form of frames up to 64 x 64 at B = 32), 2 = eight-wave blocks */
int pivp_wgrad5x5_bf16_batch_form(const float* x, int cx, int ldx, const float* h_prev, int C, const float* dG, float* dW, float* db,
                                  int B, int H, int W, int tcount, long long ts_x, long long ts_h, long long ts_dG, int form, void* stream);
/* ... with every operand as three bf16 pieces and six MFMAs per product (fp32-grade, fp32's exponent range; PIVP_PRECISION_BF16X6's weight gradient) */
int pivp_wgrad5x5_bf16x6_batch(const float* x, int cx, int ldx, const float* h_prev, int C, const float* dG, float* dW, float* db,
                               int B, int H, int W, int tcount, long long ts_x, long long ts_h, long long ts_dG, void* stream);
/* ... with every operand as two fp16 pieces and three MFMAs per product (fp32-grade; PIVP_PRECISION_FP16X3's weight gradient): dG is staged times the
 * power of two that puts the largest |value| of the batch into [2^14, 2^15), x and h_prev as they are (|values| < 65504), the sums scaled back exactly.
 * scratch: 72 * tcount floats (every timestep's partial maxima). */
int pivp_wgrad5x5_fp16x3_batch(const float* x, int cx, int ldx, const float* h_prev, int C, const float* dG, float* dW, float* db,
                               int B, int H, int W, int tcount, long long ts_x, long long ts_h, long long ts_dG, float* scratch, void* stream);

/* The fp32 ConvLSTM weight gradient on its own (backward of TM:262-266; the sweep of pivp_rollout_backward runs it on its side stream): a batch of
 * `tcount` timesteps per launch, operands as pivp_wgrad5x5_bf16_batch.  part == NULL: the round-2 kernel, fp32 atomics straight into dW / db.
 * part != NULL (pivp_wgrad5x5_f32_part_floats floats, caller-owned): the round-6 kernel (csrc/wgrad5x5p.hip: LDS-DMA staging, an XCD-aware balanced
 * partition of the (tile, pixel) work, no atomics) ADDS its segments into their slots of `part` -- overwrite != 0: stores them, so the first launch of a
 * sweep needs no zeroing -- and pivp_wgrad5x5_f32_reduce adds the slots' sum into dW and, when db is given, the column sums of dG into db, in a fixed
 * order: the result is bit-identical from run to run.  Launches that share `part` must be stream-ordered and describe the same (cx, C, B, H, W) AND agree on
 * whether h_prev is given (has_h): a launch without it (the sweep's t = 0) cuts the x rows' tiles its own way, so reduce before switching and store first.
 * form (the same in all three calls): 0 = by shape, 1 / 2 = 32 / 64 gate columns per wave.
 * Needs H, W powers of two (W >= 8), B*H*W % 16 == 0, cx % 32 == 0, C % 32 == 0; pivp_wgrad5x5_f32_part_floats returns 0 for shapes it does not serve. */
long long pivp_wgrad5x5_f32_part_floats(int cx, int C, int B, int H, int W, int form);
int pivp_wgrad5x5_f32_batch(const float* x, int cx, int ldx, const float* h_prev, int C, const float* dG, float* part, int overwrite,
                            float* dW, float* db, int B, int H, int W, int tcount, long long ts_x, long long ts_h, long long ts_dG, int form, void* stream);
int pivp_wgrad5x5_f32_reduce(int cx, int C, int has_h, float* part, float* dW, float* db, int B, int H, int W, int form, void* stream);
/* Host-only (no GPU work): the slot kernel's partition as the kernel and the reduction walk it, for tests and for sizing.  geom8 = {blocks per XCD, pixel parts,
 * tile parts, 32-column tiles per wave, tiles, 16-pixel chunks per timestep, slots per block, floats per slot}; segs (may be NULL): (block, segment, tile, first
 * chunk, end chunk) per segment; slots (may be NULL): (tile, slot) pairs in the reduction's order; at most seg_cap / slot_cap entries are written, the counts are
 * always returned. */
int pivp_wgrad5x5_f32_partition(int cx, int C, int has_h, int B, int H, int W, int form, int* geom8, int* segs, int seg_cap, int* nsegs,
                                int* slots, int slot_cap, int* nslots);
/* The fp32 ConvLSTM data gradient on its own: plain 5x5 stride-1 pad-2 convolution, x [B*H*W][cin] (pixel stride ldx), wt packed [25][cin/32][cout][32]
 * (for the data gradient: the flipped, transposed weight the sweep builds once per backward), out [B*H*W][cout] contiguous, overwritten. */
int pivp_conv5x5_f32(const float* x, int cin, int ldx, const float* wt, float* out, int cout, int B, int H, int W, void* stream);

/* --- training (what Chainer's autograd does under optimizer.update, TM:950) ---------------------------------- */
/* pivp_convlstm with the gate activations kept for BPTT: gates_out [B*H*W][4C] = tanh(j), s(i), s(f+1), s(o). */
int pivp_convlstm_train(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* bias,
                        const float* c_in, float* c_out, float* h_out, float* gates_out, int B, int H, int W, void* stream);
/* Backward of one ConvLSTM cell.  dh arrives as dh_a (stride lda) + dh_b (stride ldb, may be NULL); dc is the incoming
 * d c_t (ignored when dc_valid == 0) and is overwritten with d c_{t-1}.  d_in [B*H*W][cx+C] receives d x | d h_{t-1};
 * dW (packed like w) and db are ACCUMULATED.  dG [B*H*W][4C] and wt (size of w) are scratch. */
int pivp_convlstm_backward(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* gates,
                           const float* c_old, const float* c_new, const float* dh_a, int lda, const float* dh_b, int ldb,
                           float* dc, int dc_valid, float* dG, float* wt, float* d_in, float* dW, float* db,
                           int B, int H, int W, void* stream);
/* Backward of conv3x3s2 (mode 0) / deconv3x3s2 (mode 1) given dy with the ReLU mask already applied: dx (NULL to skip;
 * accum_dx adds into it), dW and db accumulated.  wt (size of w) is scratch. */
int pivp_conv_backward(int mode, const float* x, int cin, int ldx, const float* w, const float* dy, int cout, int ldy,
                       float* wt, float* dx, int lddx, int accum_dx, float* dW, float* db, int B, int Hin, int Win, void* stream);
/* The weight-gradient half of pivp_conv_backward the way the BPTT sweep of optimizer.update (TM:950) runs it over the timesteps:
 * `repeats` launches add their tiles into per-block partial planes (`part`: pivp_conv_backward_part_floats floats, zeroed by the caller;
 * no atomics), then one reduction adds the sum into dW; db accumulated on the side.  dW += repeats * dW_1, db += repeats * db_1. */
long long pivp_conv_backward_part_floats(int mode, int cin, int cout, int B, int Hin, int Win);
int pivp_conv_wgrad_partial(int mode, const float* x, int cin, int ldx, const float* dy, int cout, int ldy, float* part,
                            float* dW, float* db, int B, int Hin, int Win, int repeats, void* stream);
/* ... and as the sweep runs it since round 5: ONE launch takes a batch of `tcount` timesteps (operand j at x + j * x_step_bytes and
 * dy + j * dy_step_bytes; the steps may be negative) into the partial planes -- overwrite != 0: stored instead of added, so the planes' first launch
 * needs no zeroing -- and pivp_conv_wgrad_partial_reduce adds the planes' sum into dW (and the column sums of dy the launches left there into db;
 * the per-tap kernel adds those into db at launch).  The stride-2 3x3 shapes of the model (anchor map a multiple
 * of 4 x 8, channels multiples of 32) run all nine taps from one staging of the operands (csrc/wgrad3x3s2.hip). */
int pivp_conv_wgrad_partial_batch(int mode, const float* x, int cin, int ldx, long long x_step_bytes, const float* dy, int cout, int ldy,
                                  long long dy_step_bytes, int tcount, int overwrite, float* part, float* dW, float* db, int B, int Hin, int Win,
                                  void* stream);
int pivp_conv_wgrad_partial_reduce(int mode, int cin, int cout, float* part, float* dW, float* db, int B, int Hin, int Win, void* stream);
/* pivp_convlstm_backward for the sweep's last timestep (t = 0; TM:254-257: the state before it is zero and nothing reads its
 * gradient): only the cx columns of d_in (d x) are computed, the C columns of d h_{-1} are not computed (left as they are, or cleared). */
int pivp_convlstm_backward_dx_only(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* gates,
                                   const float* c_old, const float* c_new, const float* dh_a, int lda, const float* dh_b, int ldb,
                                   float* dc, int dc_valid, float* dG, float* wt, float* d_in, float* dW, float* db,
                                   int B, int H, int W, void* stream);
/* LayerNorm forward keeping (mean, rstd) per sample in stat [B][2], and its backward (dgamma/dbeta accumulated). */
int pivp_layernorm_train(const float* x, const float* gamma, const float* beta, float* out, float* partials, float* stat,
                         int B, int n, int C, int ldo, float eps, int relu, void* stream);
long long pivp_layernorm_backward_scratch_floats(int B, int n);
int pivp_layernorm_backward(const float* dy, int lddy, const float* y, int ldy, const float* x, const float* stat,
                            const float* gamma, float* partials, float* dx, float* dgamma, float* dbeta,
                            int B, int n, int C, int relu, void* stream);
/* The pair "LayerNorm backward of hidden<k> (TM:203-208) + gate backward of lstm<k> (TM:269-272)" as the BPTT sweep runs it: dy (stride
 * lddy) is the gradient at the norm's output, h the cell's h_t = the norm's input, stat its (mean, rstd); the cell's dh is the norm's dx,
 * formed inside the gate kernel from the sums of a first launch (+ dh_b, the recurrent part, may be NULL).  Writes dG [B*npix][4C],
 * updates dc in place, ACCUMULATES dgamma / dbeta.  scratch: pivp_gates_backward_ln_scratch_floats floats. */
long long pivp_gates_backward_ln_scratch_floats(int B, int n);
int pivp_gates_backward_ln(const float* gates, const float* c_old, const float* c_new, const float* dy, int lddy,
                           const float* gamma, const float* stat, const float* h, const float* dh_b, int ldb, float* dc,
                           int dc_valid, float* dG, float* dgamma, float* dbeta, float* scratch, int B, int npix, int C,
                           void* stream);
/* Chainer 2 Adam over a flat buffer (TM:860): lr_t = alpha*sqrt(1-beta2^t)/(1-beta1^t) from the host; g is scaled by gscale. */
int pivp_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr_t, double beta1, double beta2,
                   double eps, double gscale, void* stream);
/* bf16 payload of the data-parallel gradient all-reduce (BASELINE.json config 3; the reference is single-device, so this wraps
 * the gradients its optimizer.update(), TM:950, consumes): pack n fp32 gradients (16-B aligned) into a bf16 send buffer (round to
 * nearest even), and unpack the summed bf16 buffer back into the fp32 gradient buffer that pivp_adam_step reads. */
int pivp_grad_pack_bf16(const float* src, void* dst_bf16, long long n, void* stream);
int pivp_grad_unpack_bf16(const void* src_bf16, float* dst, long long n, void* stream);
/* The local half of a reduce-scatter over all links (parallel.py: GradAllReduce(algo='rs_ag'); SURVEY.md 5 / 8e, wraps TM:950 like
 * the two above): src holds `nshards` shards of `shard_len` elements each, one per peer (bf16 if src_bf16, else fp32; 16-B aligned,
 * shard_len % 8 == 0 for bf16 / % 4 for fp32); dst[i] = sum over shards in the fixed order 0 .. nshards-1, accumulated in fp32 and
 * rounded once to bf16 (dst_bf16) or kept fp32.  An all-reduce of a bf16 buffer rounds its running sum at every hop instead. */
int pivp_grad_sum_shards(const void* src, int src_bf16, int nshards, long long shard_len, void* dst, int dst_bf16, void* stream);

/* L.Convolution2D(cout,(3,3),stride=2,pad=1) (TM:501-502) + optional ReLU; w [9][cin/32][cout][32]. */
int pivp_conv3x3s2(const float* x, int cin, int ldx, const float* w, const float* bias, float* out, int cout,
                   int ldo, int relu, int B, int Hin, int Win, void* stream);

/* L.Deconvolution2D(cout,(3,3),stride=2,pad=1,outsize=2*in) (TM:505-507) + optional ReLU; w [9][cin/32][cout][32]. */
int pivp_deconv3x3s2(const float* x, int cin, int ldx, const float* w, const float* bias, float* out, int cout,
                     int ldo, int relu, int B, int Hin, int Win, void* stream);
/* L.Deconvolution2D(...) of concat(LayerNormalizationConv2D(h_raw), x1) (TM:565-566 / 574-575: [hidden6 | enc1] -> enc5, [hidden7 | enc0] ->
 * enc6) with the norm applied while the conv stages its input: one launch, the normalised tensor is never written; bit-identical to
 * pivp_layernorm + pivp_deconv3x3s2.  h_raw [B][Hin*Win][c_ln]; x1 (may be NULL) c1 channels at stride ld1; gamma / beta [Hin*Win][c_ln];
 * partials: scratch of pivp_layernorm_scratch_floats(B, Hin*Win*c_ln) floats; precision 0 fp32, 1 bf16 operands, 2 split.  Only for
 * geometries pivp_deconv3x3s2_ln_fits accepts (whole 32-channel chunks, Hin % 8 == 0, Win % 16 == 0, enough tiles). */
int pivp_deconv3x3s2_ln_fits(int c_ln, int c1, int cout, int B, int Hin, int Win);
int pivp_deconv3x3s2_ln(const float* h_raw, int c_ln, const float* x1, int c1, int ld1, const float* w, const float* bias,
                        const float* gamma, const float* beta, float eps, float* partials, float* out, int cout, int ldo, int relu,
                        int B, int Hin, int Win, int precision, void* stream);

/* L.Convolution2D(32,(5,5),stride=2,pad=2) on a planar 3-channel frame (TM:500); w [75][32]. */
int pivp_conv_enc0(const float* img, const float* w, const float* bias, float* out, int B, int H, int W, void* stream);

/* LayerNormalizationConv2D.__call__ (TM:203-208); x NHWC-flat [B][n], gamma/beta NHWC-flat [n];
 * partials: scratch of pivp_layernorm_scratch_floats(B,n) floats. */
long long pivp_layernorm_scratch_floats(int B, int n);
int pivp_layernorm(const float* x, const float* gamma, const float* beta, float* out, float* partials,
                   int B, int n, int C, int ldo, float eps, int relu, void* stream);

/* smear + enc3 1x1 + ReLU and current_state Linear (TM:556-567, TM:503, TM:676, TM:730). */
int pivp_enc3_state(const float* e2, const float* action, const float* state, const float* w3, const float* b3,
                    const float* wcs, const float* bcs, float* e3, float* state_out, int B, int HW8, int use_state,
                    void* stream);

/* masks / enc7 1x1 heads (TM:718-719, TM:315-317 | TM:454-455 | TM:387-388). */
int pivp_heads(const float* e6, const float* wm, const float* bm, const float* we, const float* be,
               float* mask_logits, float* enc7, float* layer0, int B, int HW, int num_masks, int model_type,
               void* stream);

/* CDNA kernel generator (TM:321-329); wt [K][256]; partials scratch of pivp_linear_scratch_floats(B,K). */
long long pivp_linear_scratch_floats(int B, int K);
int pivp_cdna_kernels(const float* hidden5, const float* wt, const float* bias, float* partials, float* kerns,
                      int B, int K, int num_masks, void* stream);

/* STP affine parameters (TM:457-468). */
int pivp_stp_params(const float* hidden5, const float* wt1, const float* b1, const float* w2, const float* b2,
                    float* partials, float* theta, int B, int K, void* stream);

/* flat softmax + transform + compositing (TM:720-728). aux: CDNA kerns | STP theta | DNA enc7 planes. */
int pivp_composite(const float* prev, const float* mask_logits, const float* layer0, const float* aux, float* out,
                   float* masks_out, int B, int H, int W, int num_masks, int model_type, int stp_zero_border,
                   void* stream);

/* The output side of one timestep in ONE launch: relu(norm_enc6(raw enc6)) (TM:601) -> mask logits + enc7 (TM:718-719, TM:315-317 /
 * 454-455 / 387-388) -> the motion head's finisher on the K-slice partial sums of its Linear (TM:321-329 CDNA kernels / TM:458-468 STP
 * parameters) -> flat softmax + transform + compositing (TM:720-728).  Bit-identical to pivp_heads + pivp_cdna_kernels / pivp_stp_params +
 * pivp_composite, without the logit / enc7 / layer0 round trip through HBM and two launches fewer; the plan's forward step uses it whenever
 * pivp_frame_head_fits.  Optional outputs may be null.  partials == null: `aux` holds finished kernels [B][num_masks][25] / theta [B][6]. */
typedef struct pivp_frame_head_args {
    const float* e6raw;            /* [B][H*W][64] NHWC, raw output of enc6                                             */
    const float* ln_part; int ln_nparts;   /* its (count, mean, M2, -) LayerNorm partials, ln_nparts per sample           */
    const float* gamma; const float* beta; float ln_eps;   /* norm_enc6, NHWC-flat                                        */
    const float* masks_w; const float* masks_b;            /* [64][num_masks+1], [num_masks+1]                            */
    const float* enc7_w; const float* enc7_b;              /* [64][3 | 25], [3 | 25]                                      */
    const float* prev;             /* previous frame, planar [B][3][H*W]                                                  */
    const float* partials; int kslices; const float* head_bias;   /* pivp_motion_partials output [B][kslices][256], Linear bias */
    const float* w2; const float* b2;      /* STP: identity_params/W (6,100), /b (6)                                      */
    const float* aux;
    float* out;                    /* next frame, planar [B][3][H*W]                                                      */
    float* masks_out;              /* optional softmaxed masks [B][num_masks+1][H*W]                                      */
    float* enc7;                   /* [B][3 | 25][H*W]                                                                    */
    float* logits_out; float* layer0_out; float* enc6_out; float* stat_out;   /* optional: what a training plan keeps for BPTT */
    float* kerns_out; float* vpre_out;     /* optional: finished kernels / theta; the Linear's pre-activation [B][256]    */
    int B, H, W, num_masks, model_type, stp_zero_border;
} pivp_frame_head_args_t;
int pivp_frame_head_fits(int model_type, int B, int H, int W, int num_masks, int K);   /* 1: supported, finisher included; 2: supported with
                                                                                          finished kernels in `aux`; 0: not supported */
int pivp_motion_partials(const float* hidden5, const float* wt, float* partials, int B, int K, int fp64_accumulate, void* stream);
int pivp_frame_head(const pivp_frame_head_args_t* args, void* stream);

/* F.resize_images(frame, (Hout, Wout)) of the predict path (predict_model.py:119-122): bilinear, align-corners sample grid,
 * planar [planes][Hin][Win] -> [planes][Hout][Wout], result multiplied by `scale` (1/255 there). */
int pivp_resize_images(const float* in, float* out, int planes, int Hin, int Win, int Hout, int Wout, float scale, void* stream);

/* scheduled_sample (TM:73-122) as an on-device per-sample select. */
int pivp_select_frames(const float* ground_truth, const float* generated, const unsigned char* take_gt, float* out,
                       int B, int frame_numel, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PIVP_HIP_H */
