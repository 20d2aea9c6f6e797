// C-ABI layer: the plan (Model.__init__ / reset_state / __call__ of the reference, TM:484-764) and
// the per-op entry points declared in include/pivp_hip.h.  The whole per-timestep op program is
// sequenced here in native code on one HIP stream, so the Python host only passes pointers.
#include <string.h>
#include <string>
#include <vector>

#include "../../include/pivp_hip.h"
#include "pivp_host.h"

using namespace pivp;

namespace pivp {

// extent in bytes of an NHWC view with pixel stride ld (for the kernels' buffer descriptors)
long long view_bytes(int B, int H, int W, int ld) { return (long long)B * H * W * ld * 4; }
bool fits31(long long v) { return v > 0 && v < (1LL << 31); }
int run_convlstm(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* bias,
                 const float* c_in, float* c_out, float* h_out, int B, int H, int W, hipStream_t s, int variant,
                 float* gates_out, float* ln_part, int ln_cap, int* ln_nparts, const unsigned short* w_bf16, int bf16_planes, const LnIn* ln_in) {
    IgemmDesc d;
    memset(&d, 0, sizeof(d));
    if (ln_in) {
        if (!w_bf16 || !convlstm_ln_in_ok(bf16_planes, cx, ldx, C, B, H, W) || !ln_in->gamma || !ln_in->beta || !ln_in->part || ln_in->np <= 0 || ln_in->part == ln_part)
            return PIVP_ERR_BADARG;
        d.in_g = ln_in->gamma; d.in_b = ln_in->beta; d.in_part = ln_in->part; d.in_np = ln_in->np; d.in_eps = ln_in->eps;
    }
    // h_prev == nullptr: the recurrent input is identically zero (first timestep after reset_state, TM:254-257);
    // its K range contributes exactly 0 and is skipped
    d.x0 = x; d.c0 = cx; d.ld0 = ldx; d.x1 = h_prev; d.c1 = h_prev ? C : 0; d.ld1 = C; d.wcin = cx + C;
    d.w = w; d.bias = bias;
    d.B = B; d.Hin = H; d.Win = W; d.Hg = H; d.Wg = W; d.in_step = 1;
    d.N = 4 * C; d.M = B * H * W;
    d.nphase = 1; d.deconv = 0; d.ksize = 5; d.pad = 2;
    const long long b0 = view_bytes(B, H, W, ldx), b1 = view_bytes(B, H, W, C), bw = 25LL * (cx + C) * 4 * C * 4;
    if (!fits31(b0) || !fits31(b1) || !fits31(bw) || (gates_out && !fits31(4 * b1))) return PIVP_ERR_BADARG;   // (epilogue: 32-bit buffer offsets)
    d.bytes0 = (int)b0; d.bytes1 = (int)b1; d.bytesw = (int)bw;
    d.out_step = 1; d.Hout = H; d.Wout = W;
    d.cstate_in = c_in; d.cstate_out = c_out; d.hout = h_out; d.C = C; d.gates_out = gates_out;
    d.ln_part = ln_part; d.ln_cap = ln_cap;
    // w_bf16: the bf16 pack of w (pack_lstm_bf16) selects the bf16-operand kernel; variant then is its channels per block
    // (three pieces: maps the three-plane tile does not serve -- 8 wide -- take the fp32 kernel, which is what that mode stands in for)
    // (two fp16 pieces: 8-wide maps need an even batch for the tile to fit)
    if (w_bf16 && (bf16_planes == 3 || bf16_planes == -2) && !convlstm_bf16_ok(d)) return w ? igemm_lstm(d, s, 0, ln_nparts) : PIVP_ERR_BADARG;      // (an 8-wide map with an odd batch: the fp32 kernel)
    if (w_bf16) return convlstm_bf16(d, w_bf16, s, ln_nparts, ((bf16_planes == 3 || bf16_planes == -2) && variant != 16 && variant != 32) ? 0 : variant, bf16_planes);
    return igemm_lstm(d, s, variant, ln_nparts);
}

// the eight-wave L2-direct kernels of the split modes take it: a 16-wide map, x contiguous in one 64-channel group, and a grid that picks those kernels
bool convlstm_ln_in_ok(int planes, int cx, int ldx, int C, int B, int H, int W) {
    return (planes == 3 || planes == -2) && cx <= 64 && ldx == cx && cx % 8 == 0 && C % 16 == 0 && H % 8 == 0 && W % 16 == 0;
}
int run_conv3x3s2(const float* x, int cin, int ldx, const float* w, const float* bias, float* out, int cout,
                  int ldo, int relu, int B, int Hin, int Win, hipStream_t s, int accum) {
    if (Hin % 2 || Win % 2) return PIVP_ERR_BADARG;
    IgemmDesc d;
    memset(&d, 0, sizeof(d));
    d.x0 = x; d.c0 = cin; d.ld0 = ldx; d.wcin = cin; d.w = w; d.bias = bias;
    d.B = B; d.Hin = Hin; d.Win = Win; d.Hg = Hin / 2; d.Wg = Win / 2; d.in_step = 2;
    d.N = cout; d.M = B * d.Hg * d.Wg;
    d.nphase = 1; d.deconv = 0; d.ksize = 3; d.pad = 1;
    const long long b0 = view_bytes(B, Hin, Win, ldx), bw = 9LL * cin * cout * 4;
    if (!fits31(b0) || !fits31(bw)) return PIVP_ERR_BADARG;
    d.bytes0 = (int)b0; d.bytesw = (int)bw;
    d.out_step = 1; d.Hout = d.Hg; d.Wout = d.Wg; d.out = out; d.ldo = ldo; d.relu = relu; d.accum = accum;
    return igemm_conv(d, s);
}

// conv3x3s2 of LayerNorm(x_raw): the norm (per-element gamma / beta, statistics from the producer's partials) is applied while the
// input is staged -- no launch of its own (inference rollouts).  x_raw [B][Hin*Win][cin] contiguous.
static void conv3x3s2_ln_desc(IgemmDesc& d, const float* x_raw, int cin, const float* w, const float* bias, float* out, int cout, int ldo,
                              int relu, int B, int Hin, int Win) {
    memset(&d, 0, sizeof(d));
    d.x0 = x_raw; d.c0 = cin; d.ld0 = cin; d.wcin = cin; d.w = w; d.bias = bias;
    d.B = B; d.Hin = Hin; d.Win = Win; d.Hg = Hin / 2; d.Wg = Win / 2; d.in_step = 2;
    d.N = cout; d.M = B * d.Hg * d.Wg;
    d.nphase = 1; d.deconv = 0; d.ksize = 3; d.pad = 1;
    d.out_step = 1; d.Hout = d.Hg; d.Wout = d.Wg; d.out = out; d.ldo = ldo; d.relu = relu;
}
bool conv3x3s2_ln_ok(int cin, int cout, int B, int Hin, int Win) {
    if (Hin % 2 || Win % 2 || cin % 32 || cout % 32) return false;
    IgemmDesc d;
    conv3x3s2_ln_desc(d, nullptr, cin, nullptr, nullptr, nullptr, cout, cout, 0, B, Hin, Win);
    return igemm_in_ln_ok(d) && fits31(view_bytes(B, Hin, Win, cin)) && fits31(9LL * cin * cout * 4);
}
int run_conv3x3s2_ln(const float* x_raw, int cin, const float* w, const float* bias, float* out, int cout, int ldo, int relu,
                     int B, int Hin, int Win, hipStream_t s, const float* gamma, const float* beta, const float* partials, int nparts, float eps,
                     const Enc3Fuse* fuse3, float* norm_out, int norm_ld, float* stat_out) {
    if (!x_raw || !w || !out || !gamma || !beta || !partials || nparts <= 0 || !conv3x3s2_ln_ok(cin, cout, B, Hin, Win)) return PIVP_ERR_BADARG;
    IgemmDesc d;
    conv3x3s2_ln_desc(d, x_raw, cin, w, bias, out, cout, ldo, relu, B, Hin, Win);
    d.bytes0 = (int)view_bytes(B, Hin, Win, cin); d.bytesw = (int)(9LL * cin * cout * 4);
    d.in_g = gamma; d.in_b = beta; d.in_part = partials; d.in_np = nparts; d.in_eps = eps;
    d.in_out = norm_out; d.in_out_ld = norm_ld; d.in_stat_out = stat_out;
    if (fuse3 && fuse3->e3) {
        if (cout != 64 || ldo != 64 || !relu || !bias) return PIVP_ERR_BADARG;
        d.f3_w = fuse3->w3; d.f3_b = fuse3->b3; d.f3_action = fuse3->action; d.f3_state = fuse3->state; d.f3_wcs = fuse3->wcs; d.f3_bcs = fuse3->bcs;
        d.f3_out = fuse3->e3; d.f3_state_out = fuse3->state_out; d.f3_use_state = fuse3->use_state;
    }
    int rc = igemm_validate(d, false);
    if (rc != PIVP_OK) return rc;
    return igemm_small(d, s);
}

int run_deconv3x3s2(const float* x, int cin, int ldx, const float* w, const float* bias, float* out, int cout,
                    int ldo, int relu, int B, int Hin, int Win, hipStream_t s, int accum, float* ln_part, int ln_cap,
                    int* ln_nparts, int bf16, const float* wscale_part) {
    IgemmDesc d;
    memset(&d, 0, sizeof(d));
    if (bf16 == 3 && !wscale_part) return PIVP_ERR_BADARG;
    d.wscale_part = wscale_part;
    d.bf16 = bf16;                   // precision mode bf16: honoured by the all-parities tile kernel (deconv_tile.hip), fp32 otherwise
    d.x0 = x; d.c0 = cin; d.ld0 = ldx; d.wcin = cin; d.w = w; d.bias = bias;
    d.B = B; d.Hin = Hin; d.Win = Win; d.Hg = Hin; d.Wg = Win; d.in_step = 1;
    d.N = cout; d.M = B * Hin * Win;
    d.nphase = 4; d.deconv = 1; d.ksize = 3; d.pad = 1;
    const long long b0 = view_bytes(B, Hin, Win, ldx), bw = 9LL * cin * cout * 4;
    if (!fits31(b0) || !fits31(bw)) return PIVP_ERR_BADARG;
    d.bytes0 = (int)b0; d.bytesw = (int)bw;
    d.out_step = 2; d.Hout = 2 * Hin; d.Wout = 2 * Win; d.out = out; d.ldo = ldo; d.relu = relu; d.accum = accum;
    d.ln_part = ln_part; d.ln_cap = ln_cap;
    return igemm_conv(d, s, ln_nparts);
}

// run_deconv3x3s2(x, ...) AND motion_partials(x, wt, partials, ...) -- two independent launches that read the same tensor and fill a
// fraction of the chip each (enc4 and the motion head's Linear on hidden5) -- as ONE grid when the conv is igemm_small's (else the two
// launches, in that order).  x [B][Hin*Win][cin] contiguous = the Linear's [B][K] input, K = Hin * Win * cin.
int run_deconv3x3s2_and_partials(const float* x, int cin, const float* w, const float* bias, float* out, int cout, int ldo, int relu,
                                 int B, int Hin, int Win, hipStream_t s, const float* wt, float* partials, int dbl) {
    IgemmDesc d;
    memset(&d, 0, sizeof(d));
    d.x0 = x; d.c0 = cin; d.ld0 = cin; d.wcin = cin; d.w = w; d.bias = bias;
    d.B = B; d.Hin = Hin; d.Win = Win; d.Hg = Hin; d.Wg = Win; d.in_step = 1;
    d.N = cout; d.M = B * Hin * Win;
    d.nphase = 4; d.deconv = 1; d.ksize = 3; d.pad = 1;
    const long long b0 = view_bytes(B, Hin, Win, cin), bw = 9LL * cin * cout * 4;
    if (!fits31(b0) || !fits31(bw)) return PIVP_ERR_BADARG;
    d.bytes0 = (int)b0; d.bytesw = (int)bw;
    d.out_step = 2; d.Hout = 2 * Hin; d.Wout = 2 * Win; d.out = out; d.ldo = ldo; d.relu = relu;
    const int K = Hin * Win * cin;
    // (Both in ONE grid -- round 4's igemm_small_partials_kernel -- took 21.4-22.0 us against 12.7 + 9.2 for the two launches, whichever kind of block
    // was dispatched first, and the rollout did not move: profiles/r04/NOTES.md.  Two launches; the fused grid is in the history.)
    int rc = igemm_conv(d, s);
    if (rc != PIVP_OK) return rc;
    return motion_partials(x, wt, partials, B, K, dbl, s);
}

// deconv3x3s2 of concat(LayerNorm(h_raw), x1): the norm of the first c_ln channels (per-element gamma / beta, statistics from the
// producer's partials) is applied while the all-parities tile kernel stages its patch -- no launch of its own, and the normalised tensor
// is never written (inference rollouts: [hidden6 | enc1] -> enc5, [hidden7 | enc0] -> enc6).  h_raw [B][Hin*Win][c_ln] contiguous.
static int deconv3x3s2_ln_desc(IgemmDesc& d, const float* h_raw, int c_ln, const float* x1, int c1, int ld1, const float* w, const float* bias,
                               float* out, int cout, int ldo, int relu, int B, int Hin, int Win) {
    memset(&d, 0, sizeof(d));
    d.x0 = h_raw; d.c0 = c_ln; d.ld0 = c_ln; d.x1 = x1; d.c1 = x1 ? c1 : 0; d.ld1 = ld1; d.wcin = c_ln + d.c1; d.w = w; d.bias = bias;
    d.B = B; d.Hin = Hin; d.Win = Win; d.Hg = Hin; d.Wg = Win; d.in_step = 1;
    d.N = cout; d.M = B * Hin * Win;
    d.nphase = 4; d.deconv = 1; d.ksize = 3; d.pad = 1;
    const long long b0 = view_bytes(B, Hin, Win, c_ln), b1 = d.c1 ? view_bytes(B, Hin, Win, ld1) : 0, bw = 9LL * d.wcin * cout * 4;
    if (!fits31(b0) || (d.c1 && !fits31(b1)) || !fits31(bw)) return PIVP_ERR_BADARG;
    d.bytes0 = (int)b0; d.bytes1 = (int)b1; d.bytesw = (int)bw;
    d.out_step = 2; d.Hout = 2 * Hin; d.Wout = 2 * Win; d.out = out; d.ldo = ldo; d.relu = relu;
    return PIVP_OK;
}
bool deconv3x3s2_ln_ok(int c_ln, int c1, int cout, int B, int Hin, int Win) {
    if (c_ln <= 0 || c_ln % 32 || c1 < 0 || c1 % 32 || cout % 32 || Hin % 8 || Win % 16) return false;
    return (long)B * (Hin / 8) * (Win / 16) * (cout / 32) >= 16;
}
int run_deconv3x3s2_ln(const float* h_raw, int c_ln, const float* x1, int c1, int ld1, const float* w, const float* bias, float* out, int cout,
                       int ldo, int relu, int B, int Hin, int Win, hipStream_t s, const float* gamma, const float* beta, const float* partials,
                       int nparts, float eps, float* ln_part, int ln_cap, int* ln_nparts, int bf16, float* norm_out, int norm_ld, float* stat_out,
                       const float* wscale_part, const MotionRider* rider) {
    if (!h_raw || !w || !out || !gamma || !beta || !partials || nparts <= 0 || !deconv3x3s2_ln_ok(c_ln, x1 ? c1 : 0, cout, B, Hin, Win)) return PIVP_ERR_BADARG;
    if (norm_out && (norm_ld < c_ln || norm_ld % 4 || ((uintptr_t)norm_out & 15))) return PIVP_ERR_BADARG;
    if (ln_part && ln_part == partials) return PIVP_ERR_BADARG;      // blocks finish (and write their output partial) while others still read the input's
    IgemmDesc d;
    int rc = deconv3x3s2_ln_desc(d, h_raw, c_ln, x1, c1, ld1, w, bias, out, cout, ldo, relu, B, Hin, Win);
    if (rc != PIVP_OK) return rc;
    if (bf16 == 3 && !wscale_part) return PIVP_ERR_BADARG;
    d.bf16 = bf16; d.wscale_part = wscale_part;
    d.in_g = gamma; d.in_b = beta; d.in_part = partials; d.in_np = nparts; d.in_eps = eps;
    d.ln_part = ln_part; d.ln_cap = ln_cap;
    d.in_out = norm_out; d.in_out_ld = norm_ld; d.in_stat_out = stat_out;
    if (rider && rider->mode) {
        d.rd_mode = rider->mode; d.rd_blocks = B; d.rd_KS = rider->KS; d.rd_nout = rider->nout;
        d.rd_partials = rider->partials; d.rd_bias = rider->bias; d.rd_w2 = rider->w2; d.rd_b2 = rider->b2; d.rd_out = rider->out; d.rd_vpre = rider->vpre;
    }
    rc = igemm_validate(d, false);
    if (rc != PIVP_OK) return rc;
    if (!deconv_tile_ok(d)) return PIVP_ERR_BADARG;
    return deconv_tile(d, s, ln_nparts, bf16);
}

// stride-1 K x K "same" convolution through the generic kernel (used as the ConvLSTM data gradient)
static int conv_s1_desc(IgemmDesc& d, const float* x, int cin, int ldx, const float* w, float* out, int cout, int ldo, int ksize, int B, int H, int W,
                        int accum, int wN) {
    memset(&d, 0, sizeof(d));
    d.x0 = x; d.c0 = cin; d.ld0 = ldx; d.wcin = cin; d.w = w; d.bias = nullptr;
    d.B = B; d.Hin = H; d.Win = W; d.Hg = H; d.Wg = W; d.in_step = 1;
    d.N = cout; d.M = B * H * W; d.wN = wN > cout ? wN : 0;
    d.nphase = 1; d.deconv = 0; d.ksize = ksize; d.pad = ksize / 2;
    const long long b0 = view_bytes(B, H, W, ldx), bw = (long long)ksize * ksize * cin * (wN > cout ? wN : cout) * 4;
    if (!fits31(b0) || !fits31(bw)) return PIVP_ERR_BADARG;
    d.bytes0 = (int)b0; d.bytesw = (int)bw;
    d.out_step = 1; d.Hout = H; d.Wout = W; d.out = out; d.ldo = ldo; d.relu = 0; d.accum = accum;
    // fresh output (contiguous, or the leading columns of a buffer whose rest nobody reads): the K-split path is allowed; it needs a zeroed destination
    if (!accum && (ldo == cout || d.wN)) d.ksplit_ok = 1;
    return PIVP_OK;
}
// true when run_conv_s1 with these arguments adds K-split partial sums into `out` (B * H * W * ldo floats to be zeroed first)
bool conv_s1_splits_k(int cin, int cout, int ldo, int ksize, int B, int H, int W, int wN) {
    IgemmDesc d;
    static float dummy;
    if (conv_s1_desc(d, &dummy, cin, cin, &dummy, &dummy, cout, ldo, ksize, B, H, W, 0, wN) != PIVP_OK) return false;
    return d.ksplit_ok && igemm_conv_ksplit(d) > 1;
}
int run_conv_s1(const float* x, int cin, int ldx, const float* w, float* out, int cout, int ldo, int ksize, int B, int H, int W,
                hipStream_t s, int accum, int wN, int dest_zeroed) {
    IgemmDesc d;
    int rc = conv_s1_desc(d, x, cin, ldx, w, out, cout, ldo, ksize, B, H, W, accum, wN);
    if (rc != PIVP_OK) return rc;
    if (d.ksplit_ok && !dest_zeroed && igemm_conv_ksplit(d) > 1 &&
        hipMemsetAsync(out, 0, (size_t)B * H * W * ldo * 4, s) != hipSuccess) return PIVP_ERR_LAUNCH;
    return igemm_conv(d, s);
}

// 5x5 stride-1 "same" convolution with bf16 operands (csrc/convlstm_bf16.hip); wb = pack_lstm_bf16(w, cin, cout, conv5x5_bf16_rows(cout))
static int conv5x5_bf16_desc(IgemmDesc& d, const float* x, int cin, int ldx, float* out, int cout, int ldo, int accum, int B, int H, int W) {
    memset(&d, 0, sizeof(d));
    d.x0 = x; d.c0 = cin; d.ld0 = ldx; d.wcin = cin;
    d.B = B; d.Hin = H; d.Win = W; d.Hg = H; d.Wg = W; d.in_step = 1;
    d.N = cout; d.M = B * H * W; d.nphase = 1; d.ksize = 5; d.pad = 2;
    const long long b0 = view_bytes(B, H, W, ldx);
    if (!fits31(b0)) return PIVP_ERR_BADARG;
    d.bytes0 = (int)b0;
    d.out_step = 1; d.Hout = H; d.Wout = W; d.out = out; d.ldo = ldo; d.accum = accum;
    if (!accum && ldo == cout) d.ksplit_ok = 1;     // contiguous fresh output: the K-split path may be used; it needs a zeroed destination
    return PIVP_OK;
}
bool conv5x5_bf16_splits_k(int cin, int cout, int ldo, int B, int H, int W, int planes) {
    IgemmDesc d;
    static float dummy;
    if (conv5x5_bf16_desc(d, &dummy, cin, cin, &dummy, cout, ldo, 0, B, H, W) != PIVP_OK) return false;
    return d.ksplit_ok && conv5x5_bf16_ksplit(d, planes) > 1;
}
int run_conv5x5_bf16(const float* x, int cin, int ldx, const unsigned short* wb, float* out, int cout, int ldo, int accum,
                     int B, int H, int W, hipStream_t s, int planes, int dest_zeroed, const float* ascale_part, const EpSpec* ep) {
    IgemmDesc d;
    int rc = conv5x5_bf16_desc(d, x, cin, ldx, out, cout, ldo, accum, B, H, W);
    if (rc != PIVP_OK) return rc;
    d.wscale_part = ascale_part;
    if (ep && ep->applied) *ep->applied = 0;
    if (ep && ep->src && ep->mode && !(d.ksplit_ok && conv5x5_bf16_ksplit(d, planes) > 1)) {
        d.ep_src = ep->src; d.ep_ld = ep->ld; d.ep_cols = ep->cols < cout ? ep->cols : cout; d.ep_mode = ep->mode;
        if (ep->applied) *ep->applied = 1;
    }
    if (d.ksplit_ok && !dest_zeroed && conv5x5_bf16_ksplit(d, planes) > 1 &&
        hipMemsetAsync(out, 0, (size_t)B * H * W * cout * 4, s) != hipSuccess) return PIVP_ERR_LAUNCH;
    return conv5x5_bf16(d, wb, s, planes);
}

// weight gradient of: mode 0 = conv K x K stride `stride` pad `pad`; mode 1 = transposed 3x3 s2 p1
int run_wgrad(int mode, const float* x0, int c0, int ld0, const float* x1, int c1, int ld1, int wcin, const float* dy, int ldy, int N,
              float* dw, int B, int Hx, int Wx, int Hy, int Wy, int ksize, int pad, int stride, hipStream_t s, float* db,
              int* bias_done, int bf16, int tcount, long long ts_x0, long long ts_x1, long long ts_dy, float* part, WgradDesc* desc_out,
              const float* dy_absmax, int dy_absmax_stride, int form, int part_overwrite) {
    WgradDesc d;
    memset(&d, 0, sizeof(d));
    d.x0 = x0; d.c0 = c0; d.ld0 = ld0; d.x1 = x1; d.c1 = x1 ? c1 : 0; d.ld1 = ld1; d.cin = c0 + (x1 ? c1 : 0); d.wcin = wcin;
    d.dy = dy; d.ldy = ldy; d.N = N; d.dw = dw;
    d.B = B; d.Hx = Hx; d.Wx = Wx; d.Hy = Hy; d.Wy = Wy;
    d.deconv = mode; d.ksize = ksize; d.pad = pad; d.stride = stride;
    d.Hg = mode ? Hx : Hy; d.Wg = mode ? Wx : Wy; d.M = B * d.Hg * d.Wg;
    const long long b0 = view_bytes(B, Hx, Wx, ld0), b1 = x1 ? view_bytes(B, Hx, Wx, ld1) : 0, by = view_bytes(B, Hy, Wy, ldy);
    if (!fits31(b0) || (x1 && !fits31(b1)) || !fits31(by)) return PIVP_ERR_BADARG;
    d.bytes0 = (int)b0; d.bytes1 = (int)b1; d.bytesy = (int)by;
    d.db = db;
    d.tcount = tcount; d.ts_x0 = ts_x0; d.ts_x1 = ts_x1; d.ts_dy = ts_dy;
    d.part = part; d.part_overwrite = part_overwrite;
    d.dy_absmax = dy_absmax; d.dy_absmax_stride = dy_absmax_stride;
    d.pieces = bf16 == 3 ? 3 : 0;       // (bf16: 1 = operands rounded to bf16; 3 = three bf16 pieces per operand, fp32-grade)
    d.form = form;
    if (desc_out) *desc_out = d;
    if (bf16 || dy_absmax) {   // bf16 precision mode (5x5 ConvLSTM case only): operands rounded to bf16, fp32 accumulation; db summed on the side in fp32
        if (bias_done) *bias_done = db != nullptr;
        return wgrad5x5_bf16(d, s);
    }
    return igemm_wgrad(d, s, bias_done);
}

// ConvLSTM cell backward (TM:262-272): gate math, data gradient d[x,h_prev], weight and bias gradients.
//   d_in [M][cx+C] receives d x (first cx channels) and d h_{t-1} (last C); dc is updated in place to d c_{t-1}.
static int fork_begin(const SideFork* f, hipStream_t s, hipStream_t* sw) {
    *sw = s;
    if (!f || !f->side) return PIVP_OK;
    if (hipEventRecord(f->ready, s) != hipSuccess || hipStreamWaitEvent(f->side, f->ready, 0) != hipSuccess) return PIVP_ERR_LAUNCH;
    *sw = f->side;
    return PIVP_OK;
}
static int fork_end(const SideFork* f) {
    if (!f || !f->side) return PIVP_OK;
    return hipEventRecord(f->done, f->side) == hipSuccess ? PIVP_OK : PIVP_ERR_LAUNCH;
}

int run_convlstm_backward(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* gates,
                          const float* c_old, const float* c_new, const float* dh_a, int lda, const float* dh_b, int ldb,
                          float* dc, int dc_valid, float* dG, float* wt, float* d_in, float* dW, float* db,
                          int B, int H, int W, hipStream_t s, int wt_ready, unsigned short* wt_bf16, int bf16_planes, const SideFork* fork,
                          const LnFuse* ln, int dx_only, float* dg_absmax, const EpSpec* ep) {
    const int M = B * H * W, cin = cx + C, N = 4 * C;
    if (wt_bf16 && bf16_planes == -2 && !dg_absmax) return PIVP_ERR_BADARG;
    // a K-split data gradient adds into d_in: the gate kernel clears it on the side (one launch less than a memset per cell and timestep)
    const bool zero = wt_bf16 ? conv5x5_bf16_splits_k(N, cin, cin, B, H, W, bf16_planes)
                              : (dx_only ? conv_s1_splits_k(N, cx, cin, 5, B, H, W, cin) : conv_s1_splits_k(N, cin, cin, 5, B, H, W, 0));
    int rc = lstm_gates_bwd(gates, c_old, c_new, dh_a, lda, dh_b, ldb, dc, dc_valid, dG, M, C, s, B, ln, zero ? d_in : nullptr, (long long)M * cin);
    if (rc != PIVP_OK) return rc;
    if (dg_absmax) {      // fp16 pieces: dG's power-of-two scale from its largest |value| (gradients lie far below fp16's normal range); in front of the
        // fork, because the weight gradient on the side stream reads it too.
        // (The maximum taken by the gate backward itself -- an atomic maximum of float bits per wave, behind a "can I raise it" load -- instead of this
        // launch was built and measured: the train step 21.5 ms against 21.1 with the launch, 22.5 with three bf16 pieces.  Removed.)
        rc = absmax_partials(dG, (long)M * N, dg_absmax, s);
        if (rc != PIVP_OK) return rc;
    }
    // dG is final: the weight gradient can start (on the side stream when forked), next to this layer's own data gradient
    hipStream_t sw;
    rc = fork_begin(fork, s, &sw);
    if (rc != PIVP_OK) return rc;
    if (!wt_ready) {
        rc = repack_transpose(w, wt, 25, cin, N, 1, s);                   // [25][cin/32][4C][32] -> flipped [25][4C/32][cin][32]
        if (rc != PIVP_OK) return rc;
    }
    if (wt_bf16) {   // bf16 precision mode: the data gradient with bf16 operands (wt_bf16 = bf16 pack of wt, built here unless wt_ready)
        if (!wt_ready) {
            rc = pack_lstm_bf16(wt, wt_bf16, N, cin, s, conv5x5_bf16_rows(cin), bf16_planes, 1);      // fragment-major plain pack, every map width
            if (rc != PIVP_OK) return rc;
        }
        rc = run_conv5x5_bf16(dG, N, N, wt_bf16, d_in, cin, cin, 0, B, H, W, s, bf16_planes, zero, dg_absmax, ep);
    } else {
        if (ep && ep->applied) *ep->applied = 0;      // (the fp32 data-gradient kernels have no such hook)
        // d[x,h] = conv5x5(dG, W^T flipped); dx_only: the x columns alone (the pack's first cx of cin; the h columns of d_in stay unwritten)
        rc = dx_only ? run_conv_s1(dG, N, N, wt, d_in, cx, cin, 5, B, H, W, s, 0, cin, zero)
                     : run_conv_s1(dG, N, N, wt, d_in, cin, cin, 5, B, H, W, s, 0, 0, zero);
    }
    if (rc != PIVP_OK) return rc;
    if (!dW) return PIVP_OK;   // the caller batches this layer's weight gradient over several timesteps itself (pivp_plan.hip)
    int bias_done = 0;   // the 5x5 weight-gradient kernel sums dG's columns on the side
    // (the weight gradient has a bf16 form but no split form: in the split mode it stays the fp32 kernel)
    rc = run_wgrad(0, x, cx, ldx, h_prev, C, C, cin, dG, N, N, dW, B, H, W, H, W, 5, 2, 1, sw, db, &bias_done, wt_bf16 != nullptr && bf16_planes == 1);
    if (rc != PIVP_OK) return rc;
    if (!bias_done) { rc = bias_grad(dG, N, N, M, db, sw); if (rc != PIVP_OK) return rc; }
    return fork_end(fork);
}

int run_layernorm(const float* x, const float* g, const float* b, float* out, float* partials, int B, int n, int C,
                  int ldo, float eps, int relu, hipStream_t s, float* stat_out, int fused_nparts) {
    // fused_nparts > 0: the kernel that produced x already wrote that many (count, mean, M2) partials per sample
    if (fused_nparts <= 0) {
        int rc = ln_stats(x, partials, B, n, s);
        if (rc != PIVP_OK) return rc;
    }
    return ln_apply(x, partials, g, b, out, B, n, C, ldo, eps, relu, s, stat_out, fused_nparts);
}

__global__ __launch_bounds__(256) void select_frames_kernel(const float* __restrict__ gt, const float* __restrict__ gen,
                                                            const unsigned char* __restrict__ take, float* __restrict__ out,
                                                            int frame_numel) {
    const int b = blockIdx.y;
    const float* src = take[b] ? gt : gen;
    const size_t base = (size_t)b * frame_numel;
    for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < frame_numel; i += gridDim.x * 1024)
        *reinterpret_cast<f32x4*>(out + base + i) = *reinterpret_cast<const f32x4*>(src + base + i);
}

int run_select_frames(const float* gt, const float* gen, const unsigned char* take, float* out, int B, int frame_numel, hipStream_t s) {
    if (!gt || !gen || !take || !out || B <= 0 || frame_numel <= 0 || frame_numel % 4) return PIVP_ERR_BADARG;
    hipLaunchKernelGGL(select_frames_kernel, dim3(8, B), dim3(256), 0, s, gt, gen, take, out, frame_numel);
    return PIVP_LAUNCH_STATUS();
}

// conv3x3s2 (mode 0) / deconv3x3s2 (mode 1) backward.  dy is masked in place by (y > 0) when y != null (fused ReLU).
int run_conv_backward(int mode, const float* x, int cin, int ldx, const float* w, float* dy, int cout, int ldy, const float* y, int ldyy,
                      float* wt, float* dx, int lddx, int accum_dx, float* dW, float* db, int B, int Hin, int Win, hipStream_t s,
                      int wt_ready, const SideFork* fork, float* part, WgradDesc* desc_out, const float* dy_add, int ld_add, int prec) {
    const int Hout = mode ? 2 * Hin : Hin / 2, Wout = mode ? 2 * Win : Win / 2;
    int rc = PIVP_OK;
    if (y) { rc = relu_mask(dy, ldy, y, ldyy, cout, (long)B * Hout * Wout, s, dy_add, ld_add); if (rc != PIVP_OK) return rc; }
    else if (dy_add) { rc = add_strided(dy, ldy, dy_add, ld_add, cout, (long)B * Hout * Wout, s); if (rc != PIVP_OK) return rc; }
    hipStream_t sw;
    rc = fork_begin(fork, s, &sw);      // dy is final here
    if (rc != PIVP_OK) return rc;
    if (dx) {
        if (!wt_ready) {
            rc = repack_transpose(w, wt, 9, cin, cout, 0, s);
            if (rc != PIVP_OK) return rc;
        }
        rc = mode ? run_conv3x3s2(dy, cout, ldy, wt, nullptr, dx, cin, lddx, 0, B, Hout, Wout, s, accum_dx)
                  : run_deconv3x3s2(dy, cout, ldy, wt, nullptr, dx, cin, lddx, 0, B, Hout, Wout, s, accum_dx, nullptr, 0, nullptr, prec == 1 ? 1 : 0);
        if (rc != PIVP_OK) return rc;
    }
    if (!dW) return PIVP_OK;     // the caller batches this layer's weight gradient over several timesteps itself (pivp_plan.hip); the fork's `ready` is recorded
    int bias_done = 0;     // the weight-gradient kernel sums dY's columns on the side when it can
    rc = run_wgrad(mode, x, cin, ldx, nullptr, 0, 0, cin, dy, ldy, cout, dW, B, Hin, Win, Hout, Wout, 3, 1, 2, sw, db, &bias_done, 0,
                   1, 0, 0, 0, part, desc_out);
    if (rc != PIVP_OK) return rc;
    if (!bias_done) { rc = bias_grad(dy, ldy, cout, B * Hout * Wout, db, sw); if (rc != PIVP_OK) return rc; }
    return fork_end(fork);
}

long long conv_backward_part_floats(int mode, int cin, int cout, int B, int Hin, int Win) {
    const int Hout = mode ? 2 * Hin : Hin / 2, Wout = mode ? 2 * Win : Win / 2;
    WgradDesc d;
    memset(&d, 0, sizeof(d));
    d.c0 = cin; d.ld0 = cin; d.cin = cin; d.wcin = cin; d.N = cout; d.ldy = cout; d.B = B; d.Hx = Hin; d.Wx = Win; d.Hy = Hout; d.Wy = Wout;
    d.deconv = mode; d.ksize = 3; d.pad = 1; d.stride = 2;
    d.Hg = mode ? Hin : Hout; d.Wg = mode ? Win : Wout; d.M = B * d.Hg * d.Wg;
    return igemm_wgrad_part_floats(d);
}
// descriptor of a ConvLSTM weight gradient's ONE-timestep geometry (what the partial buffer's size and the reduction depend on)
static void lstm_wgrad_geom(WgradDesc& d, int cx, int C, int B, int H, int W) {
    memset(&d, 0, sizeof(d));
    d.c0 = cx; d.ld0 = cx; d.c1 = C; d.ld1 = C; d.cin = cx + C; d.wcin = cx + C; d.N = 4 * C; d.ldy = 4 * C;
    d.B = B; d.Hx = H; d.Wx = W; d.Hy = H; d.Wy = W; d.Hg = H; d.Wg = W; d.M = B * H * W;
    d.ksize = 5; d.pad = 2; d.stride = 1;
}
// (the sweep's t = 0 has no h operand: its launch differentiates the x rows only -- fewer tiles, another partition of the same buffer, reduced on its own)
long long lstm_wgrad_part_floats(int cx, int C, int B, int H, int W, int form) {
    WgradDesc d;
    lstm_wgrad_geom(d, cx, C, B, H, W);
    d.form = form;
    if (!wgrad5x5p_ok(d)) return 0;
    const long long full = wgrad5x5p_part_floats(d);
    d.c1 = 0; d.cin = cx;
    const long long xonly = wgrad5x5p_part_floats(d);
    return full > xonly ? full : xonly;
}
int lstm_wgrad_reduce(int cx, int C, int has_h, float* part, float* dW, float* db, int B, int H, int W, hipStream_t s, int form) {
    WgradDesc d;
    lstm_wgrad_geom(d, cx, C, B, H, W);
    d.form = form;
    if (!has_h) { d.c1 = 0; d.cin = cx; }
    d.part = part; d.dw = dW; d.db = db;
    return igemm_wgrad_reduce(d, s);
}

}  // namespace pivp

#ifndef PIVP_BUILD_DIGEST
#define PIVP_BUILD_DIGEST "unstamped"      // a build that did not go through build.py: _lib.load() refuses it
#endif
extern "C" const char* pivp_build_digest(void) { return PIVP_BUILD_DIGEST; }
#ifndef PIVP_BUILD_FLAGS
#define PIVP_BUILD_FLAGS ""                // the compile flags beyond build.py's standard set (PIVP_EXTRA_FLAGS): "" = the product build
#endif
extern "C" const char* pivp_build_flags(void) { return PIVP_BUILD_FLAGS; }
extern "C" int pivp_abi_version(void) { return 17; }   // 9: + pivp_build_digest, pivp_grad_sum_shards, pivp_frame_head; 8: + pivp_gates_backward_ln (op entry of the norm + gate backward pair); 7: + bf16 gradient payload, batched bf16 weight gradient, partial-plane / dx-only op entries

extern "C" int pivp_convlstm(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* bias,
                             const float* c_in, float* c_out, float* h_out, int B, int H, int W, void* stream) {
    if (!x || !w || !bias || !c_in || !c_out || !h_out) return PIVP_ERR_BADARG;
    return run_convlstm(x, cx, ldx, h_prev, C, w, bias, c_in, c_out, h_out, B, H, W, (hipStream_t)stream);
}
extern "C" int pivp_convlstm_v(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* bias,
                               const float* c_in, float* c_out, float* h_out, int B, int H, int W, int variant, void* stream) {
    if (!x || !w || !bias || !c_in || !c_out || !h_out || variant < 0 || variant > 4) return PIVP_ERR_BADARG;
    return run_convlstm(x, cx, ldx, h_prev, C, w, bias, c_in, c_out, h_out, B, H, W, (hipStream_t)stream, variant);
}
extern "C" int pivp_convlstm_train(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* bias,
                                   const float* c_in, float* c_out, float* h_out, float* gates_out, int B, int H, int W, void* stream) {
    if (!x || !w || !bias || !c_in || !c_out || !h_out || !gates_out) return PIVP_ERR_BADARG;
    return run_convlstm(x, cx, ldx, h_prev, C, w, bias, c_in, c_out, h_out, B, H, W, (hipStream_t)stream, 0, gates_out);
}
extern "C" int pivp_convlstm_backward(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* gates,
                                      const float* c_old, const float* c_new, const float* dh_a, int lda, const float* dh_b, int ldb,
                                      float* dc, int dc_valid, float* dG, float* wt, float* d_in, float* dW, float* db,
                                      int B, int H, int W, void* stream) {
    if (!x || !w || !gates || !c_old || !c_new || !dc || !dG || !wt || !d_in || !dW || !db) return PIVP_ERR_BADARG;
    return run_convlstm_backward(x, cx, ldx, h_prev, C, w, gates, c_old, c_new, dh_a, lda, dh_b, ldb, dc, dc_valid, dG, wt, d_in,
                                 dW, db, B, H, W, (hipStream_t)stream);
}
// conv3x3s2 (mode 0) / deconv3x3s2 (mode 1) backward given dy (already ReLU-masked): dx (optionally accumulated), dW, db
extern "C" int pivp_conv_backward(int mode, const float* x, int cin, int ldx, const float* w, const float* dy, int cout, int ldy,
                                  float* wt, float* dx, int lddx, int accum_dx, float* dW, float* db, int B, int Hin, int Win,
                                  void* stream) {
    if (!x || !w || !dy || !wt || !dW || !db || mode < 0 || mode > 1) return PIVP_ERR_BADARG;
    return run_conv_backward(mode, x, cin, ldx, w, const_cast<float*>(dy), cout, ldy, nullptr, 0, wt, dx, lddx, accum_dx, dW, db, B, Hin, Win,
                             (hipStream_t)stream);
}
// The weight-gradient half of pivp_conv_backward as the BPTT sweep runs it: `repeats` launches (one per timestep there; here the same
// operands every time) add their tiles into the per-block partial planes `part` (pivp_conv_backward_part_floats floats, zeroed by the
// caller) with plain loads and stores, then ONE reduction sums the pixel splits into dW; db is accumulated on the side by the tap blocks
// that see every dy element once.  Result: dW += repeats * (x^T . dy per tap), db += repeats * column sums of dy.
extern "C" long long pivp_conv_backward_part_floats(int mode, int cin, int cout, int B, int Hin, int Win) {
    if (mode < 0 || mode > 1 || cin <= 0 || cout <= 0 || B <= 0 || Hin <= 0 || Win <= 0) return PIVP_ERR_BADARG;
    return conv_backward_part_floats(mode, cin, cout, B, Hin, Win);
}
extern "C" int pivp_conv_wgrad_partial(int mode, const float* x, int cin, int ldx, const float* dy, int cout, int ldy, float* part,
                                       float* dW, float* db, int B, int Hin, int Win, int repeats, void* stream) {
    if (!x || !dy || !part || !dW || !db || mode < 0 || mode > 1 || repeats < 1) return PIVP_ERR_BADARG;
    const int Hout = mode ? 2 * Hin : Hin / 2, Wout = mode ? 2 * Win : Win / 2;
    WgradDesc desc;
    for (int r = 0; r < repeats; ++r) {
        int bias_done = 0;
        int rc = run_wgrad(mode, x, cin, ldx, nullptr, 0, 0, cin, dy, ldy, cout, dW, B, Hin, Win, Hout, Wout, 3, 1, 2, (hipStream_t)stream, db,
                           &bias_done, 0, 1, 0, 0, 0, part, &desc);
        if (rc != PIVP_OK) return rc;
        if (!bias_done) { rc = bias_grad(dy, ldy, cout, B * Hout * Wout, db, (hipStream_t)stream); if (rc != PIVP_OK) return rc; }
    }
    return igemm_wgrad_reduce(desc, (hipStream_t)stream);
}
// One launch of that weight gradient over a BATCH of timesteps (WgradDesc::tcount; operand j at x + j * x_step_bytes, dy + j * dy_step_bytes,
// negative steps allowed: the sweep walks the forward slabs backwards) into the partial planes -- overwrite != 0: stored, not added (the planes'
// first launch needs no zeroing) -- and the reduction into dW as a call of its own.
extern "C" int pivp_conv_wgrad_partial_batch(int mode, const float* x, int cin, int ldx, long long x_step_bytes, const float* dy, int cout, int ldy,
                                             long long dy_step_bytes, int tcount, int overwrite, float* part, float* dW, float* db, int B, int Hin, int Win,
                                             void* stream) {
    if (!x || !dy || !part || !dW || !db || mode < 0 || mode > 1 || tcount < 1) return PIVP_ERR_BADARG;
    const int Hout = mode ? 2 * Hin : Hin / 2, Wout = mode ? 2 * Win : Win / 2;
    int bias_done = 0;
    int rc = run_wgrad(mode, x, cin, ldx, nullptr, 0, 0, cin, dy, ldy, cout, dW, B, Hin, Win, Hout, Wout, 3, 1, 2, (hipStream_t)stream, db, &bias_done, 0,
                       tcount, x_step_bytes, 0, dy_step_bytes, part, nullptr, nullptr, 0, 0, overwrite ? 1 : 0);
    if (rc != PIVP_OK) return rc;
    if (!bias_done)
        for (int j = 0; j < tcount; ++j) {
            rc = bias_grad(reinterpret_cast<const float*>(reinterpret_cast<const char*>(dy) + j * dy_step_bytes), ldy, cout, B * Hout * Wout, db, (hipStream_t)stream);
            if (rc != PIVP_OK) return rc;
        }
    return PIVP_OK;
}
extern "C" int pivp_conv_wgrad_partial_reduce(int mode, int cin, int cout, float* part, float* dW, float* db, int B, int Hin, int Win, void* stream) {
    if (!part || !dW || !db || mode < 0 || mode > 1) return PIVP_ERR_BADARG;
    const int Hout = mode ? 2 * Hin : Hin / 2, Wout = mode ? 2 * Win : Win / 2;
    WgradDesc d;
    memset(&d, 0, sizeof(d));
    d.c0 = cin; d.ld0 = cin; d.cin = cin; d.wcin = cin; d.N = cout; d.ldy = cout; d.B = B; d.Hx = Hin; d.Wx = Win; d.Hy = Hout; d.Wy = Wout;
    d.deconv = mode; d.ksize = 3; d.pad = 1; d.stride = 2;
    d.Hg = mode ? Hin : Hout; d.Wg = mode ? Win : Wout; d.M = B * d.Hg * d.Wg;
    d.part = part; d.dw = dW; d.db = db;      // (the nine-tap kernel leaves its column sums in the planes: the reduction adds them into db)
    return igemm_wgrad_reduce(d, (hipStream_t)stream);
}
// The fp32 ConvLSTM weight gradient on its own (the sweep runs it on the side stream): a batch of `tcount` timesteps per launch as pivp_wgrad5x5_bf16_batch.
// part == NULL: the round-2 kernel (atomics straight into dW / db).  part != NULL (pivp_wgrad5x5_f32_part_floats floats): the round-6 kernel adds -- overwrite
// != 0: stores -- its segments into the partial slots; pivp_wgrad5x5_f32_reduce then adds the slots' sum into dW and db (fixed order: bit-reproducible).
extern "C" long long pivp_wgrad5x5_f32_part_floats(int cx, int C, int B, int H, int W, int form) {
    if (cx <= 0 || C <= 0 || B <= 0 || H <= 0 || W <= 0 || form < 0 || form > 2) return PIVP_ERR_BADARG;
    return lstm_wgrad_part_floats(cx, C, B, H, W, form);
}
extern "C" int pivp_wgrad5x5_f32_batch(const float* x, int cx, int ldx, const float* h_prev, int C, const float* dG, float* part, int overwrite,
                                       float* dW, float* db, int B, int H, int W, int tcount, long long ts_x, long long ts_h, long long ts_dG, int form,
                                       void* stream) {
    if (!x || !dG || !dW || tcount < 1 || form < 0 || form > 2 || (part && lstm_wgrad_part_floats(cx, C, B, H, W, form) <= 0)) return PIVP_ERR_BADARG;
    int bias_done = 0;
    int rc = run_wgrad(0, x, cx, ldx, h_prev, C, C, cx + C, dG, 4 * C, 4 * C, dW, B, H, W, H, W, 5, 2, 1, (hipStream_t)stream, db, &bias_done, 0,
                       tcount, ts_x, ts_h, ts_dG, part, nullptr, nullptr, 0, form, overwrite ? 1 : 0);
    if (rc != PIVP_OK) return rc;
    if (db && !bias_done)
        for (int j = 0; j < tcount; ++j) {
            rc = bias_grad(reinterpret_cast<const float*>(reinterpret_cast<const char*>(dG) + j * ts_dG), 4 * C, 4 * C, B * H * W, db, (hipStream_t)stream);
            if (rc != PIVP_OK) return rc;
        }
    return PIVP_OK;
}
extern "C" int pivp_wgrad5x5_f32_reduce(int cx, int C, int has_h, float* part, float* dW, float* db, int B, int H, int W, int form, void* stream) {
    if (!part || !dW || form < 0 || form > 2 || lstm_wgrad_part_floats(cx, C, B, H, W, form) <= 0) return PIVP_ERR_BADARG;
    return lstm_wgrad_reduce(cx, C, has_h, part, dW, db, B, H, W, (hipStream_t)stream, form);
}
// The slot kernel's partition walked on the host (no GPU work): geom8 = {blocks per XCD, pixel parts, tile parts, 32-column tiles per wave, tiles, 16-pixel chunks per
// timestep, slots per block, floats per slot}; segs: (block, segment, tile, first chunk, end chunk) per segment in kernel order; slots: (tile, slot) pairs in the
// reduction's order.  The caps bound what is written; the counts are always returned.  tests/test_host.py checks that every (tile, chunk) is covered exactly once and
// that the reduction reads exactly the slots the kernel writes.
extern "C" int pivp_wgrad5x5_f32_partition(int cx, int C, int has_h, int B, int H, int W, int form, int* geom8, int* segs, int seg_cap, int* nsegs,
                                           int* slots, int slot_cap, int* nslots) {
    if (cx <= 0 || C <= 0 || B <= 0 || H <= 0 || W <= 0 || form < 0 || form > 2 || !geom8 || !nsegs || !nslots) return PIVP_ERR_BADARG;
    WgradDesc d;
    lstm_wgrad_geom(d, cx, C, B, H, W);
    d.form = form;
    if (!has_h) { d.c1 = 0; d.cin = cx; }
    if (!wgrad5x5p_ok(d)) return PIVP_ERR_BADARG;
    return wgrad5x5p_partition(d, geom8, segs, seg_cap, nsegs, slots, slot_cap, nslots);
}
// The fp32 ConvLSTM data gradient on its own: a plain 5x5 stride-1 pad-2 convolution of x [B*H*W][cin] with wt (packed [25][cin/32][cout][32]: for the data
// gradient the flipped, transposed weight) into out [B*H*W][cout] (contiguous); tile and K split as the sweep chooses them (out is cleared first when K is split).
extern "C" int pivp_conv5x5_f32(const float* x, int cin, int ldx, const float* wt, float* out, int cout, int B, int H, int W, void* stream) {
    if (!x || !wt || !out) return PIVP_ERR_BADARG;
    return run_conv_s1(x, cin, ldx, wt, out, cout, cout, 5, B, H, W, (hipStream_t)stream, 0, 0, 0);
}
// pivp_convlstm_backward for the sweep's LAST timestep (t = 0): nobody reads d h_{-1}, so only the cx columns of d_in are computed
// (the data gradient runs on the first cx columns of the transposed weight pack) and the h columns of d_in are not computed (left as they are, or cleared with the rest of d_in for a K-split data gradient).
extern "C" int pivp_convlstm_backward_dx_only(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* gates,
                                              const float* c_old, const float* c_new, const float* dh_a, int lda, const float* dh_b, int ldb,
                                              float* dc, int dc_valid, float* dG, float* wt, float* d_in, float* dW, float* db,
                                              int B, int H, int W, void* stream) {
    if (!x || !w || !gates || !c_old || !c_new || !dc || !dG || !wt || !d_in || !dW || !db) return PIVP_ERR_BADARG;
    return run_convlstm_backward(x, cx, ldx, h_prev, C, w, gates, c_old, c_new, dh_a, lda, dh_b, ldb, dc, dc_valid, dG, wt, d_in,
                                 dW, db, B, H, W, (hipStream_t)stream, 0, nullptr, 1, nullptr, nullptr, 1);
}
extern "C" int pivp_layernorm_train(const float* x, const float* gamma, const float* beta, float* out, float* partials, float* stat,
                                    int B, int n, int C, int ldo, float eps, int relu, void* stream) {
    if (!stat) return PIVP_ERR_BADARG;
    return run_layernorm(x, gamma, beta, out, partials, B, n, C, ldo, eps, relu, (hipStream_t)stream, stat);
}
// bf16-operand ConvLSTM (BASELINE.json config 3): weights re-packed to bf16 once, operands rounded to bf16 on the way into LDS,
// fp32 accumulation / gates / state.  nch: 0 automatic, 16 or 32 channels per block.
// Split mode (three bf16 MFMAs per product, 16 bits of product mantissa): weights packed as hi / lo planes, twice the elements.
extern "C" int pivp_pack_lstm_bf16x3(const float* w, void* w_bf16, int cin_total, int C, void* stream) {
    if (C <= 0) return PIVP_ERR_BADARG;
    return pack_lstm_bf16(w, (unsigned short*)w_bf16, cin_total, 4 * C, (hipStream_t)stream, 0, 2);
}
extern "C" int pivp_convlstm_bf16x3(const float* x, int cx, int ldx, const float* h_prev, int C, const void* w_bf16, const float* bias,
                                    const float* c_in, float* c_out, float* h_out, float* gates_out, float* ln_part, int ln_cap,
                                    int* ln_nparts, int B, int H, int W, int nch, void* stream) {
    if (!x || !w_bf16 || !bias || !c_in || !c_out || !h_out) return PIVP_ERR_BADARG;
    return run_convlstm(x, cx, ldx, h_prev, C, nullptr, bias, c_in, c_out, h_out, B, H, W, (hipStream_t)stream, nch, gates_out,
                        ln_part, ln_cap, ln_nparts, (const unsigned short*)w_bf16, 2);
}
extern "C" int pivp_pack_lstm_bf16x6(const float* w, void* w_bf16, int cin_total, int C, void* stream) {
    if (C <= 0) return PIVP_ERR_BADARG;
    return pack_lstm_bf16(w, (unsigned short*)w_bf16, cin_total, 4 * C, (hipStream_t)stream, 0, 3);
}
extern "C" int pivp_convlstm_bf16x6(const float* x, int cx, int ldx, const float* h_prev, int C, const void* w_bf16, const float* bias,
                                    const float* c_in, float* c_out, float* h_out, float* gates_out, float* ln_part, int ln_cap,
                                    int* ln_nparts, int B, int H, int W, int nch, void* stream) {
    if (!x || !w_bf16 || !bias || !c_in || !c_out || !h_out || (nch != 0 && nch != 16 && nch != 32)) return PIVP_ERR_BADARG;
    return run_convlstm(x, cx, ldx, h_prev, C, nullptr, bias, c_in, c_out, h_out, B, H, W, (hipStream_t)stream, nch, gates_out,
                        ln_part, ln_cap, ln_nparts, (const unsigned short*)w_bf16, 3);
}
extern "C" int pivp_pack_lstm_fp16x3(const float* w, void* w_bf16, int cin_total, int C, int map_width, void* stream) {
    if (C <= 0 || map_width <= 0) return PIVP_ERR_BADARG;
    // (fragment-major for every map width; map_width is kept in the signature for ABI stability)
    return pack_lstm_bf16(w, (unsigned short*)w_bf16, cin_total, 4 * C, (hipStream_t)stream, 0, -2, 0);
}
extern "C" int pivp_convlstm_fp16x3(const float* x, int cx, int ldx, const float* h_prev, int C, const void* w_bf16, const float* bias,
                                    const float* c_in, float* c_out, float* h_out, float* gates_out, float* ln_part, int ln_cap,
                                    int* ln_nparts, int B, int H, int W, int nch, void* stream) {
    if (!x || !w_bf16 || !bias || !c_in || !c_out || !h_out || (nch != 0 && nch != 16 && nch != 32 && nch != 256)) return PIVP_ERR_BADARG;
    return run_convlstm(x, cx, ldx, h_prev, C, nullptr, bias, c_in, c_out, h_out, B, H, W, (hipStream_t)stream, nch, gates_out,
                        ln_part, ln_cap, ln_nparts, (const unsigned short*)w_bf16, -2);
}
extern "C" long long pivp_lstm_bf16_weight_elems(int cin_total, int C) {
    if (cin_total <= 0 || cin_total % 32 || C <= 0) return PIVP_ERR_BADARG;
    return (long long)lstm_bf16_weight_elems(cin_total, 4 * C);
}
extern "C" int pivp_pack_lstm_bf16(const float* w, void* w_bf16, int cin_total, int C, void* stream) {
    if (C <= 0) return PIVP_ERR_BADARG;
    return pack_lstm_bf16(w, (unsigned short*)w_bf16, cin_total, 4 * C, (hipStream_t)stream);
}
extern "C" int pivp_convlstm_bf16(const float* x, int cx, int ldx, const float* h_prev, int C, const void* w_bf16, const float* bias,
                                  const float* c_in, float* c_out, float* h_out, float* gates_out, float* ln_part, int ln_cap,
                                  int* ln_nparts, int B, int H, int W, int nch, void* stream) {
    if (!x || !w_bf16 || !bias || !c_in || !c_out || !h_out) return PIVP_ERR_BADARG;
    return run_convlstm(x, cx, ldx, h_prev, C, nullptr, bias, c_in, c_out, h_out, B, H, W, (hipStream_t)stream, nch, gates_out,
                        ln_part, ln_cap, ln_nparts, (const unsigned short*)w_bf16);
}
// Plain 5x5 stride-1 "same" convolution with bf16 operands (the ConvLSTM data gradient of the bf16 mode): w fp32 K-inner packed
// [25][cin/32][cout][32]; w_bf16: scratch of pivp_conv5x5_bf16_weight_elems(cin, cout) 2-byte elements, (re)built by this call.
extern "C" long long pivp_conv5x5_bf16_weight_elems(int cin, int cout) {
    if (cin <= 0 || cin % 32 || cout <= 0) return PIVP_ERR_BADARG;
    return (long long)lstm_bf16_weight_elems(cin, conv5x5_bf16_rows(cout));
}
extern "C" int pivp_conv5x5_bf16(const float* x, int cin, int ldx, const float* w, void* w_bf16, float* out, int cout, int ldo, int accum,
                                 int B, int H, int W, void* stream) {
    if (!x || !w || !w_bf16 || !out || cin <= 0 || cout <= 0) return PIVP_ERR_BADARG;
    int rc = pack_lstm_bf16(w, (unsigned short*)w_bf16, cin, cout, (hipStream_t)stream, conv5x5_bf16_rows(cout));
    if (rc != PIVP_OK) return rc;
    return run_conv5x5_bf16(x, cin, ldx, (const unsigned short*)w_bf16, out, cout, ldo, accum, B, H, W, (hipStream_t)stream);
}
// split form (two bf16 pieces per operand): w_bf16 holds 2 * pivp_conv5x5_bf16_weight_elems(cin, cout) elements
extern "C" int pivp_conv5x5_bf16x3(const float* x, int cin, int ldx, const float* w, void* w_bf16, float* out, int cout, int ldo, int accum,
                                   int B, int H, int W, void* stream) {
    if (!x || !w || !w_bf16 || !out || cin <= 0 || cout <= 0) return PIVP_ERR_BADARG;
    int rc = pack_lstm_bf16(w, (unsigned short*)w_bf16, cin, cout, (hipStream_t)stream, conv5x5_bf16_rows(cout), 2);
    if (rc != PIVP_OK) return rc;
    return run_conv5x5_bf16(x, cin, ldx, (const unsigned short*)w_bf16, out, cout, ldo, accum, B, H, W, (hipStream_t)stream, 2);
}
// three-piece form (six MFMAs per product, fp32-grade): w_bf16 holds 3 * pivp_conv5x5_bf16_weight_elems(cin, cout) elements; W % 16 == 0
extern "C" int pivp_conv5x5_bf16x6(const float* x, int cin, int ldx, const float* w, void* w_bf16, float* out, int cout, int ldo, int accum,
                                   int B, int H, int W, void* stream) {
    if (!x || !w || !w_bf16 || !out || cin <= 0 || cout <= 0 || (W % 16 && (W % 8 || B % 2))) return PIVP_ERR_BADARG;
    int rc = pack_lstm_bf16(w, (unsigned short*)w_bf16, cin, cout, (hipStream_t)stream, conv5x5_bf16_rows(cout), 3, 1);
    if (rc != PIVP_OK) return rc;
    return run_conv5x5_bf16(x, cin, ldx, (const unsigned short*)w_bf16, out, cout, ldo, accum, B, H, W, (hipStream_t)stream, 3);
}
// two-fp16-piece form (three MFMAs per product, fp32-grade; the fp16x3 mode's data gradients): x is staged times the power of two that puts its largest
// |value| into [2^14, 2^15) (gradients lie far below fp16's normal range), w as in pivp_pack_lstm_fp16x3; w_bf16 holds 2 * pivp_conv5x5_bf16_weight_elems
// + 256 elements, scratch 66 floats (x's partial maxima); x contiguous (ldx == cin), W % 16 == 0 or W % 8 == 0 with an even batch
extern "C" int pivp_conv5x5_fp16x3(const float* x, int cin, int ldx, const float* w, void* w_bf16, float* out, int cout, int ldo, int accum,
                                   int B, int H, int W, float* scratch, void* stream) {
    if (!x || !w || !w_bf16 || !out || !scratch || cin <= 0 || cout <= 0 || (W % 16 && (W % 8 || B % 2)) || ldx != cin || B <= 0 || H <= 0) return PIVP_ERR_BADARG;
    int rc = pack_lstm_bf16(w, (unsigned short*)w_bf16, cin, cout, (hipStream_t)stream, conv5x5_bf16_rows(cout), -2, 1);
    if (rc != PIVP_OK) return rc;
    rc = absmax_partials(x, (long)B * H * W * cin, scratch, (hipStream_t)stream);
    if (rc != PIVP_OK) return rc;
    return run_conv5x5_bf16(x, cin, ldx, (const unsigned short*)w_bf16, out, cout, ldo, accum, B, H, W, (hipStream_t)stream, -2, 0, scratch);
}
// ConvLSTM weight gradient with bf16 operands: dW (K-inner packed like the weight, [25][(cx+C)/32][4C][32]) += x|h^T . dG per tap.
extern "C" int pivp_wgrad5x5_bf16(const float* x, int cx, int ldx, const float* h_prev, int C, const float* dG, float* dW, float* db,
                                  int B, int H, int W, void* stream) {
    if (!x || !dG || !dW || C <= 0 || cx <= 0) return PIVP_ERR_BADARG;
    return run_wgrad(0, x, cx, ldx, h_prev, C, C, cx + C, dG, 4 * C, 4 * C, dW, B, H, W, H, W, 5, 2, 1, (hipStream_t)stream, db,
                     nullptr, 1);
}
// ... with two fp16 pieces per operand and three MFMAs per product (fp32-grade; the fp16x3 mode's weight gradient), a batch of timesteps as below:
// scratch: 72 * tcount floats (the partial maxima of every timestep's dG: it is staged times a power of two from the largest of the batch)
extern "C" int pivp_wgrad5x5_fp16x3_batch(const float* x, int cx, int ldx, const float* h_prev, int C, const float* dG, float* dW, float* db,
                                          int B, int H, int W, int tcount, long long ts_x, long long ts_h, long long ts_dG, float* scratch, void* stream) {
    if (!x || !dG || !dW || !scratch || C <= 0 || cx <= 0 || tcount < 1 || B <= 0 || H <= 0 || W <= 0) return PIVP_ERR_BADARG;
    for (int j = 0; j < tcount; ++j) {
        const int rc = absmax_partials(reinterpret_cast<const float*>(reinterpret_cast<const char*>(dG) + j * ts_dG), (long)B * H * W * 4 * C,
                                       scratch + (size_t)j * 72, (hipStream_t)stream);
        if (rc != PIVP_OK) return rc;
    }
    return run_wgrad(0, x, cx, ldx, h_prev, C, C, cx + C, dG, 4 * C, 4 * C, dW, B, H, W, H, W, 5, 2, 1, (hipStream_t)stream, db,
                     nullptr, 0, tcount, ts_x, ts_h, ts_dG, nullptr, nullptr, scratch, 72);
}
// ... with three bf16 pieces per operand and six MFMAs per product (fp32-grade, fp32's exponent range; the bf16x6 mode's weight gradient)
extern "C" int pivp_wgrad5x5_bf16x6_batch(const float* x, int cx, int ldx, const float* h_prev, int C, const float* dG, float* dW, float* db,
                                          int B, int H, int W, int tcount, long long ts_x, long long ts_h, long long ts_dG, void* stream) {
    if (!x || !dG || !dW || C <= 0 || cx <= 0 || tcount < 1) return PIVP_ERR_BADARG;
    return run_wgrad(0, x, cx, ldx, h_prev, C, C, cx + C, dG, 4 * C, 4 * C, dW, B, H, W, H, W, 5, 2, 1, (hipStream_t)stream, db,
                     nullptr, 3, tcount, ts_x, ts_h, ts_dG);
}
// ... of a BATCH of timesteps in one launch (the sum over pixels runs over timesteps too): timestep j reads x + j * ts_x, h_prev + j * ts_h,
// dG + j * ts_dG (byte strides, multiples of 16, may be negative: the backward sweep walks time downwards)
extern "C" int pivp_wgrad5x5_bf16_batch(const float* x, int cx, int ldx, const float* h_prev, int C, const float* dG, float* dW, float* db,
                                        int B, int H, int W, int tcount, long long ts_x, long long ts_h, long long ts_dG, void* stream) {
    if (!x || !dG || !dW || C <= 0 || cx <= 0 || tcount < 1) return PIVP_ERR_BADARG;
    return run_wgrad(0, x, cx, ldx, h_prev, C, C, cx + C, dG, 4 * C, 4 * C, dW, B, H, W, H, W, 5, 2, 1, (hipStream_t)stream, db,
                     nullptr, 1, tcount, ts_x, ts_h, ts_dG);
}
// ... with the block form chosen by the caller: 1 = four-wave blocks (32 channels x 32 columns, about one per CU: they co-reside with the backward sweep's
// small kernels), 2 = eight-wave blocks (32 x 64: half the patch traffic per multiply-add), 0 = by size as pivp_wgrad5x5_bf16_batch does
extern "C" int pivp_wgrad5x5_bf16_batch_form(const float* x, int cx, int ldx, const float* h_prev, int C, const float* dG, float* dW, float* db,
                                             int B, int H, int W, int tcount, long long ts_x, long long ts_h, long long ts_dG, int form, void* stream) {
    if (!x || !dG || !dW || C <= 0 || cx <= 0 || tcount < 1 || form < 0 || form > 2) return PIVP_ERR_BADARG;
    return run_wgrad(0, x, cx, ldx, h_prev, C, C, cx + C, dG, 4 * C, 4 * C, dW, B, H, W, H, W, 5, 2, 1, (hipStream_t)stream, db,
                     nullptr, 1, tcount, ts_x, ts_h, ts_dG, nullptr, nullptr, nullptr, 0, form);
}
static int convlstm_ln_cap(int H, int W, int C) {
    const int tiles = ((H * W + 31) / 32) * (C / 32), slices = ln_stats_slices(H * W * C);
    return tiles > slices ? tiles : slices;
}
extern "C" long long pivp_convlstm_ln_scratch_floats(int B, int H, int W, int C) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 32) return PIVP_ERR_BADARG;
    return (long long)B * convlstm_ln_cap(H, W, C) * 4;
}
extern "C" int pivp_convlstm_ln(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* bias,
                                const float* c_in, float* c_out, float* h_out, const float* gamma, const float* beta,
                                float* ln_out, int ldo, float* partials, float eps, int B, int H, int W, int variant, int* fused,
                                void* stream) {
    if (!x || !w || !bias || !c_in || !c_out || !h_out || !gamma || !beta || !ln_out || !partials || variant < 0 || variant > 4)
        return PIVP_ERR_BADARG;
    if (C <= 0 || C % 32) return PIVP_ERR_BADARG;
    int np = 0;
    int rc = run_convlstm(x, cx, ldx, h_prev, C, w, bias, c_in, c_out, h_out, B, H, W, (hipStream_t)stream, variant, nullptr,
                          partials, convlstm_ln_cap(H, W, C), &np);
    if (rc != PIVP_OK) return rc;
    if (fused) *fused = np > 0;
    return run_layernorm(h_out, gamma, beta, ln_out, partials, B, H * W * C, C, ldo, eps, 0, (hipStream_t)stream, nullptr, np);
}
extern "C" long long pivp_layernorm_backward_scratch_floats(int B, int n) {
    if (B <= 0 || n <= 0) return PIVP_ERR_BADARG;
    return (long long)B * ln_bwd_slices(n) * 2;
}
extern "C" int pivp_layernorm_backward(const float* dy, int lddy, const float* y, int ldy, const float* x, const float* stat,
                                       const float* gamma, float* partials, float* dx, float* dgamma, float* dbeta,
                                       int B, int n, int C, int relu, void* stream) {
    return ln_backward(dy, lddy, y, ldy, x, stat, gamma, partials, dx, dgamma, dbeta, B, n, C, relu, (hipStream_t)stream);
}
// LayerNorm backward of the norm behind a ConvLSTM + the cell's gate backward, the way the BPTT sweep runs the pair (pivp_plan.hip:
// lnb_cell + lstmb): sums + parameter planes in one launch (ln_bwd_sums_params_kernel), then the gate kernel forms the norm's dx from the sums on the fly.
extern "C" long long pivp_gates_backward_ln_scratch_floats(int B, int n) {
    if (B <= 0 || n <= 0) return PIVP_ERR_BADARG;
    return (long long)B * ln_bwd_slices(n) * 2 + ln_bwd_param_part_floats(n);
}
extern "C" int pivp_gates_backward_ln(const float* gates, const float* c_old, const float* c_new, const float* dy, int lddy,
                                      const float* gamma, const float* stat, const float* h, const float* dh_b, int ldb, float* dc,
                                      int dc_valid, float* dG, float* dgamma, float* dbeta, float* scratch, int B, int npix, int C,
                                      void* stream) {
    if (!gates || !c_old || !c_new || !dy || !gamma || !stat || !h || !dc || !dG || !dgamma || !dbeta || !scratch) return PIVP_ERR_BADARG;
    if (B <= 0 || npix <= 0 || C <= 0 || C % 4 || lddy < C || lddy % 4) return PIVP_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int n = npix * C;
    float* partials = scratch;
    float* part = scratch + (size_t)B * ln_bwd_slices(n) * 2;
    if (hipMemsetAsync(part, 0, (size_t)ln_bwd_param_part_floats(n) * 4, s) != hipSuccess) return PIVP_ERR_LAUNCH;
    LnFuse lf;
    memset(&lf, 0, sizeof(lf));
    lf.dy = dy; lf.lddy = lddy; lf.gamma = gamma; lf.stat = stat; lf.h = h;
    lf.partials = partials; lf.S = ln_bwd_slices(n);
    int rc = ln_backward(dy, lddy, nullptr, 0, h, stat, gamma, partials, nullptr, dgamma, dbeta, B, n, C, 0, s, part);
    if (rc != PIVP_OK) return rc;
    rc = lstm_gates_bwd(gates, c_old, c_new, nullptr, 0, dh_b, ldb, dc, dc_valid, dG, B * npix, C, s, B, &lf);
    if (rc != PIVP_OK) return rc;
    return ln_bwd_params_reduce(part, dgamma, dbeta, B, n, s);
}
extern "C" int pivp_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr_t, double beta1, double beta2,
                              double eps, double gscale, void* stream) {
    return adam_step(p, g, m, v, (long)n, lr_t, beta1, beta2, eps, gscale, (hipStream_t)stream);
}
extern "C" int pivp_grad_pack_bf16(const float* src, void* dst_bf16, long long n, void* stream) {
    return grad_pack_bf16(src, dst_bf16, (long)n, (hipStream_t)stream);
}
extern "C" int pivp_grad_sum_shards(const void* src, int src_bf16, int nshards, long long shard_len, void* dst, int dst_bf16, void* stream) {
    return grad_sum_shards(src, src_bf16, nshards, (long)shard_len, dst, dst_bf16, (hipStream_t)stream);
}
extern "C" int pivp_grad_unpack_bf16(const void* src_bf16, float* dst, long long n, void* stream) {
    return grad_unpack_bf16(src_bf16, dst, (long)n, (hipStream_t)stream);
}
extern "C" int pivp_conv3x3s2(const float* x, int cin, int ldx, const float* w, const float* bias, float* out, int cout,
                              int ldo, int relu, int B, int Hin, int Win, void* stream) {
    if (!x || !w || !out) return PIVP_ERR_BADARG;
    return run_conv3x3s2(x, cin, ldx, w, bias, out, cout, ldo, relu, B, Hin, Win, (hipStream_t)stream);
}
extern "C" int pivp_deconv3x3s2(const float* x, int cin, int ldx, const float* w, const float* bias, float* out, int cout,
                                int ldo, int relu, int B, int Hin, int Win, void* stream) {
    if (!x || !w || !out) return PIVP_ERR_BADARG;
    return run_deconv3x3s2(x, cin, ldx, w, bias, out, cout, ldo, relu, B, Hin, Win, (hipStream_t)stream);
}
// deconv3x3s2 of concat(LayerNorm(h_raw), x1) in ONE conv launch (the norm applied while the tile kernel stages its patch): what inference
// rollouts run for enc5 / enc6.  The statistics are taken here by ln_stats into `partials` (pivp_layernorm_scratch_floats(B, Hin*Win*c_ln));
// in the rollout they come from the ConvLSTM's epilogue.  precision: 0 fp32, 1 bf16 operands, 2 split.  PIVP_ERR_BADARG for geometries
// the tile kernel does not take (pivp_deconv3x3s2_ln_fits).
extern "C" int pivp_deconv3x3s2_ln_fits(int c_ln, int c1, int cout, int B, int Hin, int Win) {
    return deconv3x3s2_ln_ok(c_ln, c1, cout, B, Hin, Win) ? 1 : 0;
}
extern "C" int pivp_deconv3x3s2_ln(const float* h_raw, int c_ln, const float* x1, int c1, int ld1, const float* w, const float* bias,
                                   const float* gamma, const float* beta, float eps, float* partials, float* out, int cout, int ldo, int relu,
                                   int B, int Hin, int Win, int precision, void* stream) {
    if (!h_raw || !w || !out || !gamma || !beta || !partials || precision < 0 || precision > 2) return PIVP_ERR_BADARG;
    if (!deconv3x3s2_ln_ok(c_ln, x1 ? c1 : 0, cout, B, Hin, Win)) return PIVP_ERR_BADARG;
    const int n = Hin * Win * c_ln;
    int rc = ln_stats(h_raw, partials, B, n, (hipStream_t)stream);
    if (rc != PIVP_OK) return rc;
    return run_deconv3x3s2_ln(h_raw, c_ln, x1, c1, ld1, w, bias, out, cout, ldo, relu, B, Hin, Win, (hipStream_t)stream, gamma, beta, partials,
                              ln_stats_slices(n), eps, nullptr, 0, nullptr, precision);
}
// bf16-operand form (precision mode bf16): x and w rounded to bf16 on the way into LDS, fp32 accumulation / bias / ReLU.  Only maps that the
// all-parities tile kernel takes (Hin % 8 == 0, Win % 16 == 0, at least 16 blocks) run in bf16; others fall to the fp32 kernels.
extern "C" int pivp_deconv3x3s2_bf16(const float* x, int cin, int ldx, const float* w, const float* bias, float* out, int cout,
                                     int ldo, int relu, int B, int Hin, int Win, void* stream) {
    if (!x || !w || !out) return PIVP_ERR_BADARG;
    return run_deconv3x3s2(x, cin, ldx, w, bias, out, cout, ldo, relu, B, Hin, Win, (hipStream_t)stream, 0, nullptr, 0, nullptr, 1);
}
extern "C" int pivp_deconv3x3s2_bf16x3(const float* x, int cin, int ldx, const float* w, const float* bias, float* out, int cout,
                                       int ldo, int relu, int B, int Hin, int Win, void* stream) {   // split mode: two bf16 pieces per operand
    if (!x || !w || !out) return PIVP_ERR_BADARG;
    return run_deconv3x3s2(x, cin, ldx, w, bias, out, cout, ldo, relu, B, Hin, Win, (hipStream_t)stream, 0, nullptr, 0, nullptr, 2);
}
// two fp16 pieces per operand, three MFMAs per product (precision mode PIVP_PRECISION_FP16X3): scratch = 66 floats (the weights' partial maxima)
extern "C" int pivp_deconv3x3s2_fp16x3(const float* x, int cin, int ldx, const float* w, const float* bias, float* out, int cout,
                                       int ldo, int relu, int B, int Hin, int Win, float* scratch, void* stream) {
    if (!x || !w || !out || !scratch || cin <= 0 || cout <= 0) return PIVP_ERR_BADARG;
    int rc = absmax_partials(w, 9L * cin * cout, scratch, (hipStream_t)stream);
    if (rc != PIVP_OK) return rc;
    return run_deconv3x3s2(x, cin, ldx, w, bias, out, cout, ldo, relu, B, Hin, Win, (hipStream_t)stream, 0, nullptr, 0, nullptr, 3, scratch);
}
extern "C" int pivp_conv_enc0(const float* img, const float* w, const float* bias, float* out, int B, int H, int W, void* stream) {
    return conv_enc0(img, w, bias, out, B, H, W, (hipStream_t)stream);
}
extern "C" long long pivp_layernorm_scratch_floats(int B, int n) {
    if (B <= 0 || n <= 0) return PIVP_ERR_BADARG;
    return (long long)B * ln_stats_slices(n) * 4;
}
extern "C" int pivp_layernorm(const float* x, const float* gamma, const float* beta, float* out, float* partials,
                              int B, int n, int C, int ldo, float eps, int relu, void* stream) {
    return run_layernorm(x, gamma, beta, out, partials, B, n, C, ldo, eps, relu, (hipStream_t)stream);
}
extern "C" int pivp_enc3_state(const float* e2, const float* action, const float* state, const float* w3, const float* b3,
                               const float* wcs, const float* bcs, float* e3, float* state_out, int B, int HW8, int use_state,
                               void* stream) {
    return enc3_state(e2, action, state, w3, b3, wcs, bcs, e3, state_out, B, HW8, use_state, (hipStream_t)stream);
}
extern "C" int pivp_heads(const float* e6, const float* wm, const float* bm, const float* we, const float* be,
                          float* mask_logits, float* enc7, float* layer0, int B, int HW, int num_masks, int model_type,
                          void* stream) {
    if (model_type < 0 || model_type > 2) return PIVP_ERR_BADARG;
    return heads_1x1(e6, wm, bm, we, be, mask_logits, enc7, layer0, B, HW, num_masks + 1,
                     model_type == PIVP_MODEL_DNA ? 25 : 3, model_type, (hipStream_t)stream);
}
extern "C" long long pivp_linear_scratch_floats(int B, int K) {
    if (B <= 0 || K <= 0) return PIVP_ERR_BADARG;
    return motion_partials_floats(B, K);      // [B][K slices][256] + the tail pivp_frame_head's finisher may read (never summed)
}
extern "C" int pivp_cdna_kernels(const float* hidden5, const float* wt, const float* bias, float* partials, float* kerns,
                                 int B, int K, int num_masks, void* stream) {
    return cdna_kernels(hidden5, wt, bias, partials, kerns, B, K, num_masks, (hipStream_t)stream);
}
extern "C" int pivp_stp_params(const float* hidden5, const float* wt1, const float* b1, const float* w2, const float* b2,
                               float* partials, float* theta, int B, int K, void* stream) {
    return stp_params(hidden5, wt1, b1, w2, b2, partials, theta, B, K, (hipStream_t)stream);
}
extern "C" int pivp_frame_head_fits(int model_type, int B, int H, int W, int num_masks, int K) {
    if (!frame_head_ok(model_type, B, H, W, num_masks)) return 0;
    return (model_type == PIVP_MODEL_DNA || frame_head_finishes(K)) ? 1 : 2;
}
extern "C" int pivp_motion_partials(const float* hidden5, const float* wt, float* partials, int B, int K, int fp64_accumulate, void* stream) {
    return motion_partials(hidden5, wt, partials, B, K, fp64_accumulate, (hipStream_t)stream);
}
extern "C" int pivp_frame_head(const pivp_frame_head_args_t* g, void* stream) {
    if (!g) return PIVP_ERR_BADARG;
    FrameHeadArgs a;
    a.e6raw = g->e6raw; a.ln_part = g->ln_part; a.ln_nparts = g->ln_nparts; a.gamma = g->gamma; a.beta = g->beta; a.eps = g->ln_eps;
    a.wm = g->masks_w; a.bm = g->masks_b; a.we = g->enc7_w; a.be = g->enc7_b; a.prev = g->prev;
    a.partials = g->partials; a.KS = g->kslices; a.hbias = g->head_bias; a.w2 = g->w2; a.b2 = g->b2; a.aux = g->aux;
    a.out = g->out; a.masks_out = g->masks_out; a.enc7 = g->enc7;
    a.logits_out = g->logits_out; a.layer0_out = g->layer0_out; a.y_out = g->enc6_out; a.stat_out = g->stat_out;
    a.kerns_out = g->kerns_out; a.vpre_out = g->vpre_out;
    a.B = g->B; a.H = g->H; a.W = g->W; a.NM = g->num_masks; a.stp_zero = g->stp_zero_border;
    return frame_head(a, g->model_type, (hipStream_t)stream);
}
extern "C" int pivp_composite(const float* prev, const float* mask_logits, const float* layer0, const float* aux, float* out,
                              float* masks_out, int B, int H, int W, int num_masks, int model_type, int stp_zero_border,
                              void* stream) {
    return composite(prev, mask_logits, layer0, aux, out, masks_out, B, H, W, num_masks, model_type, stp_zero_border,
                     (hipStream_t)stream);
}
extern "C" int pivp_resize_images(const float* in, float* out, int planes, int Hin, int Win, int Hout, int Wout, float scale, void* stream) {
    return resize_bilinear(in, out, planes, Hin, Win, Hout, Wout, scale, (hipStream_t)stream);
}
extern "C" int pivp_select_frames(const float* ground_truth, const float* generated, const unsigned char* take_gt, float* out,
                                  int B, int frame_numel, void* stream) {
    if (!ground_truth || !generated || !take_gt || !out || B <= 0 || frame_numel <= 0 || frame_numel % 4) return PIVP_ERR_BADARG;
    return run_select_frames(ground_truth, generated, take_gt, out, B, frame_numel, (hipStream_t)stream);
}
