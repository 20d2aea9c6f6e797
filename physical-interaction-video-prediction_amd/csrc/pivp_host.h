// Host-side launch helpers shared by the per-op C ABI (pivp_c_api.hip) and the plan (pivp_plan.hip).
#pragma once
#include "pivp_kernels.h"

namespace pivp {

long long view_bytes(int B, int H, int W, int ld);
bool fits31(long long v);
// the x operand of a ConvLSTM launch as a RAW tensor whose LayerNorm is applied while it is staged (split precision modes' eight-wave kernels)
struct LnIn { const float* gamma; const float* beta; const float* part; int np; float eps; };
bool convlstm_ln_in_ok(int planes, int cx, int ldx, int C, int B, int H, int W);
int run_convlstm(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* bias,
                 const float* c_in, float* c_out, float* h_out, int B, int H, int W, hipStream_t s, int variant = 0,
                 float* gates_out = nullptr, float* ln_part = nullptr, int ln_cap = 0, int* ln_nparts = nullptr,
                 const unsigned short* w_bf16 = nullptr, int bf16_planes = 1, const LnIn* ln_in = nullptr);
int run_conv3x3s2(const float* x, int cin, int ldx, const float* w, const float* bias, float* out, int cout,
                  int ldo, int relu, int B, int Hin, int Win, hipStream_t s, int accum = 0);
// conv3x3s2 of LayerNorm(x_raw) with the norm applied while the input is staged (IgemmDesc::in_g); conv3x3s2_ln_ok tells whether the
// geometry qualifies (output tiles of 32 anchors inside one sample)
bool conv3x3s2_ln_ok(int cin, int cout, int B, int Hin, int Win);
// group 3 (1x1 conv with the smeared action / state) + the state predictor in the epilogue of the conv that feeds them (IgemmDesc::f3_*)
struct Enc3Fuse { const float* w3; const float* b3; const float* action; const float* state; const float* wcs; const float* bcs; float* e3; float* state_out; int use_state; };
int run_conv3x3s2_ln(const float* x_raw, int cin, const float* w, const float* bias, float* out, int cout, int ldo, int relu,
                     int B, int Hin, int Win, hipStream_t s, const float* gamma, const float* beta, const float* partials, int nparts, float eps,
                     const Enc3Fuse* fuse3 = nullptr, float* norm_out = nullptr, int norm_ld = 0, float* stat_out = nullptr);      // training plans: the normalised input and (mean, rstd) are kept
int run_deconv3x3s2(const float* x, int cin, int ldx, const float* w, const float* bias, float* out, int cout,
                    int ldo, int relu, int B, int Hin, int Win, hipStream_t s, int accum = 0,
                    float* ln_part = nullptr, int ln_cap = 0, int* ln_nparts = nullptr, int bf16 = 0,
                    const float* wscale_part = nullptr);   // bf16 == 3 (two fp16 pieces): absmax_partials(w)
int run_deconv3x3s2_and_partials(const float* x, int cin, const float* w, const float* bias, float* out, int cout, int ldo, int relu,
                                 int B, int Hin, int Win, hipStream_t s, const float* wt, float* partials, int dbl);   // + motion_partials(x, wt, ...) in the same grid
// deconv3x3s2 of concat(LayerNorm(h_raw) [c_ln channels], x1 [c1 channels, stride ld1]) with the norm applied while the tile kernel stages its
// patch (IgemmDesc::in_g); `partials` = the producer's (count, mean, M2) partials of h_raw, never the same buffer as ln_part (the output's)
bool deconv3x3s2_ln_ok(int c_ln, int c1, int cout, int B, int Hin, int Win);
// the motion head's finisher carried by a transposed-conv launch as B blocks behind its tiles (IgemmDesc::rd_*): mode 1 CDNA (out = kerns [B][nout],
// vpre [B][256] or null), 2 STP (out = theta [B][6], vpre = relu(Linear(100)) [B][256] or null)
struct MotionRider { int mode, KS, nout; const float* partials; const float* bias; const float* w2; const float* b2; float* out; float* vpre; };
int run_deconv3x3s2_ln(const float* h_raw, int c_ln, const float* x1, int c1, int ld1, const float* w, const float* bias, float* out, int cout,
                       int ldo, int relu, int B, int Hin, int Win, hipStream_t s, const float* gamma, const float* beta, const float* partials,
                       int nparts, float eps, float* ln_part = nullptr, int ln_cap = 0, int* ln_nparts = nullptr, int bf16 = 0,
                       float* norm_out = nullptr, int norm_ld = 0, float* stat_out = nullptr,    // training plans: the normalised tensor and (mean, rstd) are kept
                       const float* wscale_part = nullptr, const MotionRider* rider = nullptr);
int run_conv_s1(const float* x, int cin, int ldx, const float* w, float* out, int cout, int ldo, int ksize, int B, int H, int W,
                hipStream_t s, int accum = 0, int wN = 0,    // wN: columns of the weight pack when only its first `cout` are wanted
                int dest_zeroed = 0);                        // 1: the caller has cleared `out` (see conv_s1_splits_k)
bool conv_s1_splits_k(int cin, int cout, int ldo, int ksize, int B, int H, int W, int wN);
bool conv5x5_bf16_splits_k(int cin, int cout, int ldo, int B, int H, int W, int planes = 1);
// IgemmDesc::ep_*: a second tensor met in the plain 5x5 bf16 convolution's epilogue (mode 1 ReLU mask, 2 add) on its first `cols` output columns.
// *applied (host) tells the caller whether the launch took it (unsplit grid) or the separate pass is still the caller's to run.
struct EpSpec { const float* src; int ld, cols, mode; int* applied; };
int run_wgrad(int mode, const float* x0, int c0, int ld0, const float* x1, int c1, int ld1, int wcin, const float* dy, int ldy, int N,
              float* dw, int B, int Hx, int Wx, int Hy, int Wy, int ksize, int pad, int stride, hipStream_t s,
              float* db = nullptr, int* bias_done = nullptr, int bf16 = 0,    // 5x5 ConvLSTM case: 1 = operands rounded to bf16, 3 = three bf16 pieces each (fp32-grade)
              int tcount = 1, long long ts_x0 = 0, long long ts_x1 = 0, long long ts_dy = 0,    // a batch of timesteps: WgradDesc
              float* part = nullptr, WgradDesc* desc_out = nullptr,    // part: WgradDesc::part; desc_out: the descriptor that was launched
              const float* dy_absmax = nullptr, int dy_absmax_stride = 0,    // two fp16 pieces per operand (WgradDesc::dy_absmax; 5x5 ConvLSTM case only)
              int form = 0,                                                   // WgradDesc::form (bf16 operands, a batch of timesteps: four- / eight-wave blocks)
              int part_overwrite = 0);                                        // WgradDesc::part_overwrite
int run_convlstm_backward(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* gates,
                          const float* c_old, const float* c_new, const float* dh_a, int lda, const float* dh_b, int ldb,
                          float* dc, int dc_valid, float* dG, float* wt, float* d_in, float* dW, float* db,
                          int B, int H, int W, hipStream_t s, int wt_ready = 0, unsigned short* wt_bf16 = nullptr, int bf16_planes = 1,
                          const SideFork* fork = nullptr, const LnFuse* ln = nullptr,    // ln: dh_a is formed from the LayerNorm behind the cell
                          int dx_only = 0,    // 1: d h_{t-1} is not needed (the sweep's last timestep): only the cx columns of d_in are computed
                          float* dg_absmax = nullptr,         // 66 floats: receives dG's partial maxima (absmax_partials), the scale of the fp16-piece data gradient
                          const EpSpec* ep = nullptr);        // the data gradient's epilogue hook (bf16 / split-precision data gradients on unsplit grids)
                                                              // (bf16_planes == -2 needs it) and of the fp16-piece weight gradient (WgradDesc::dy_absmax)
int run_conv5x5_bf16(const float* x, int cin, int ldx, const unsigned short* wb, float* out, int cout, int ldo, int accum,
                     int B, int H, int W, hipStream_t s, int planes = 1, int dest_zeroed = 0,
                     const float* ascale_part = nullptr,      // planes == -2: absmax_partials(x) (the activations' power-of-two scale)
                     const EpSpec* ep = nullptr);
int run_conv_backward(int mode, const float* x, int cin, int ldx, const float* w, float* dy, int cout, int ldy, const float* y, int ldyy,
                      float* wt, float* dx, int lddx, int accum_dx, float* dW, float* db, int B, int Hin, int Win, hipStream_t s,
                      int wt_ready = 0, const SideFork* fork = nullptr, float* part = nullptr, WgradDesc* desc_out = nullptr,
                      const float* dy_add = nullptr, int ld_add = 0,    // dy_add: a second gradient into the same output, added in the ReLU-mask pass
                      int prec = 0);                                    // 1 (the bf16 precision mode): the data gradient's operands rounded to bf16 where the transposed
                                                                        // conv's tile kernel takes it (enc1's); the weight gradient stays fp32 (its bf16 form -- transposing
                                                                        // LDS reads, 2 MFMAs per 32-pixel chunk -- was built and is slower: profiles/r05/NOTES.md)
// floats of WgradDesc::part a conv3x3s2 (mode 0) / deconv3x3s2 (mode 1) weight gradient of these sizes needs
long long conv_backward_part_floats(int mode, int cin, int cout, int B, int Hin, int Win);
// the fp32 ConvLSTM weight gradient's partial slots (csrc/wgrad5x5p.hip): floats of WgradDesc::part (0: the shape is not served), and the reduction
// (has_h = 0: of launches without an h operand -- the sweep's t = 0 --, which cut the x rows' tiles their own way: reduce before switching)
long long lstm_wgrad_part_floats(int cx, int C, int B, int H, int W, int form = 0);      // form: WgradDesc::form (0 by shape, 1 / 2 = 32 / 64 columns per wave)
int lstm_wgrad_reduce(int cx, int C, int has_h, float* part, float* dW, float* db, int B, int H, int W, hipStream_t s, int form = 0);
int run_layernorm(const float* x, const float* g, const float* b, float* out, float* partials, int B, int n, int C,
                  int ldo, float eps, int relu, hipStream_t s, float* stat_out = nullptr, int fused_nparts = 0);
// run-time wave priority of the main stream's kernels (pivp_common.h): every translation unit with such kernels, on stream s
int main_prio_set_backward(int on, hipStream_t s);
int main_prio_set_backward_heads(int on, hipStream_t s);
int main_prio_set_convlstm_bf16(int on, hipStream_t s);
int main_prio_set_conv5x5_bf16(int on, hipStream_t s);
int main_prio_set_deconv_tile(int on, hipStream_t s);
int main_prio_set_igemm_f32(int on, hipStream_t s);
int main_prio_set_igemm_small(int on, hipStream_t s);
int run_select_frames(const float* gt, const float* gen, const unsigned char* take, float* out, int B, int frame_numel, hipStream_t s);

}  // namespace pivp
