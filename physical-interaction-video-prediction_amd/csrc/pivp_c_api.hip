// C-ABI layer: the plan (Model.__init__ / reset_state / __call__ of the reference, TM:484-764) and
// the per-op entry points declared in include/pivp_hip.h.  The whole per-timestep op program is
// sequenced here in native code on one HIP stream, so the Python host only passes pointers.
#include <string.h>
#include <string>
#include <vector>

#include "../../include/pivp_hip.h"
#include "pivp_kernels.h"

using namespace pivp;

namespace {

struct LstmSpec { const char* name; int cx; int C; int level; };  // level: 2 -> H/2, 4 -> H/4, 8 -> H/8
const LstmSpec kLstm[7] = {
    {"lstm1", 32, 32, 2}, {"lstm2", 32, 32, 2}, {"lstm3", 32, 64, 4}, {"lstm4", 64, 64, 4},
    {"lstm5", 64, 128, 8}, {"lstm6", 128, 64, 4}, {"lstm7", 96, 32, 2}};

struct ParamInfo { std::string name; long long numel; const float* ptr; };

struct Slab {                 // per-timestep activations, offsets in floats from the workspace base
    size_t cat7, n1, n2, cat6, n3, n4, e2, e3, n5, e4, e5, e6;
    size_t h[7], c[7];
};

// extent in bytes of an NHWC view with pixel stride ld (for the kernels' buffer descriptors)
long long view_bytes(int B, int H, int W, int ld) { return (long long)B * H * W * ld * 4; }
bool fits31(long long v) { return v > 0 && v < (1LL << 31); }

int run_convlstm(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* bias,
                 const float* c_in, float* c_out, float* h_out, int B, int H, int W, hipStream_t s, int variant = 0,
                 float* gates_out = nullptr) {
    IgemmDesc d;
    memset(&d, 0, sizeof(d));
    // h_prev == nullptr: the recurrent input is identically zero (first timestep after reset_state, TM:254-257);
    // its K range contributes exactly 0 and is skipped
    d.x0 = x; d.c0 = cx; d.ld0 = ldx; d.x1 = h_prev; d.c1 = h_prev ? C : 0; d.ld1 = C; d.wcin = cx + C;
    d.w = w; d.bias = bias;
    d.B = B; d.Hin = H; d.Win = W; d.Hg = H; d.Wg = W; d.in_step = 1;
    d.N = 4 * C; d.M = B * H * W;
    d.nphase = 1; d.deconv = 0; d.ksize = 5; d.pad = 2;
    const long long b0 = view_bytes(B, H, W, ldx), b1 = view_bytes(B, H, W, C), bw = 25LL * (cx + C) * 4 * C * 4;
    if (!fits31(b0) || !fits31(b1) || !fits31(bw)) return PIVP_ERR_BADARG;
    d.bytes0 = (int)b0; d.bytes1 = (int)b1; d.bytesw = (int)bw;
    d.out_step = 1; d.Hout = H; d.Wout = W;
    d.cstate_in = c_in; d.cstate_out = c_out; d.hout = h_out; d.C = C; d.gates_out = gates_out;
    return igemm_lstm(d, s, variant);
}

int run_conv3x3s2(const float* x, int cin, int ldx, const float* w, const float* bias, float* out, int cout,
                  int ldo, int relu, int B, int Hin, int Win, hipStream_t s, int accum = 0) {
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

int run_deconv3x3s2(const float* x, int cin, int ldx, const float* w, const float* bias, float* out, int cout,
                    int ldo, int relu, int B, int Hin, int Win, hipStream_t s, int accum = 0) {
    IgemmDesc d;
    memset(&d, 0, sizeof(d));
    d.x0 = x; d.c0 = cin; d.ld0 = ldx; d.wcin = cin; d.w = w; d.bias = bias;
    d.B = B; d.Hin = Hin; d.Win = Win; d.Hg = Hin; d.Wg = Win; d.in_step = 1;
    d.N = cout; d.M = B * Hin * Win;
    d.nphase = 4; d.deconv = 1; d.ksize = 3; d.pad = 1;
    const long long b0 = view_bytes(B, Hin, Win, ldx), bw = 9LL * cin * cout * 4;
    if (!fits31(b0) || !fits31(bw)) return PIVP_ERR_BADARG;
    d.bytes0 = (int)b0; d.bytesw = (int)bw;
    d.out_step = 2; d.Hout = 2 * Hin; d.Wout = 2 * Win; d.out = out; d.ldo = ldo; d.relu = relu; d.accum = accum;
    return igemm_conv(d, s);
}

// stride-1 K x K "same" convolution through the generic kernel (used as the ConvLSTM data gradient)
int run_conv_s1(const float* x, int cin, int ldx, const float* w, float* out, int cout, int ldo, int ksize, int B, int H, int W,
                hipStream_t s, int accum = 0) {
    IgemmDesc d;
    memset(&d, 0, sizeof(d));
    d.x0 = x; d.c0 = cin; d.ld0 = ldx; d.wcin = cin; d.w = w; d.bias = nullptr;
    d.B = B; d.Hin = H; d.Win = W; d.Hg = H; d.Wg = W; d.in_step = 1;
    d.N = cout; d.M = B * H * W;
    d.nphase = 1; d.deconv = 0; d.ksize = ksize; d.pad = ksize / 2;
    const long long b0 = view_bytes(B, H, W, ldx), bw = (long long)ksize * ksize * cin * cout * 4;
    if (!fits31(b0) || !fits31(bw)) return PIVP_ERR_BADARG;
    d.bytes0 = (int)b0; d.bytesw = (int)bw;
    d.out_step = 1; d.Hout = H; d.Wout = W; d.out = out; d.ldo = ldo; d.relu = 0; d.accum = accum;
    return igemm_conv(d, s);
}

// weight gradient of: mode 0 = conv K x K stride `stride` pad `pad`; mode 1 = transposed 3x3 s2 p1
int run_wgrad(int mode, const float* x0, int c0, int ld0, const float* x1, int c1, int ld1, int wcin, const float* dy, int ldy, int N,
              float* dw, int B, int Hx, int Wx, int Hy, int Wy, int ksize, int pad, int stride, hipStream_t s) {
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
    return igemm_wgrad(d, s);
}

// ConvLSTM cell backward (TM:262-272): gate math, data gradient d[x,h_prev], weight and bias gradients.
//   d_in [M][cx+C] receives d x (first cx channels) and d h_{t-1} (last C); dc is updated in place to d c_{t-1}.
int run_convlstm_backward(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* gates,
                          const float* c_old, const float* c_new, const float* dh_a, int lda, const float* dh_b, int ldb,
                          float* dc, int dc_valid, float* dG, float* wt, float* d_in, float* dW, float* db,
                          int B, int H, int W, hipStream_t s) {
    const int M = B * H * W, cin = cx + C, N = 4 * C;
    int rc = lstm_gates_bwd(gates, c_old, c_new, dh_a, lda, dh_b, ldb, dc, dc_valid, dG, M, C, s);
    if (rc != PIVP_OK) return rc;
    rc = repack_transpose(w, wt, 25, cin, N, 1, s);                       // [25][cin/32][4C][32] -> flipped [25][4C/32][cin][32]
    if (rc != PIVP_OK) return rc;
    rc = run_conv_s1(dG, N, N, wt, d_in, cin, cin, 5, B, H, W, s);        // d[x,h] = conv5x5(dG, W^T flipped)
    if (rc != PIVP_OK) return rc;
    rc = run_wgrad(0, x, cx, ldx, h_prev, C, C, cin, dG, N, N, dW, B, H, W, H, W, 5, 2, 1, s);
    if (rc != PIVP_OK) return rc;
    return bias_grad(dG, N, N, M, db, s);
}

int run_layernorm(const float* x, const float* g, const float* b, float* out, float* partials, int B, int n, int C,
                  int ldo, float eps, int relu, hipStream_t s, float* stat_out = nullptr) {
    int rc = ln_stats(x, partials, B, n, s);
    if (rc != PIVP_OK) return rc;
    return ln_apply(x, partials, g, b, out, B, n, C, ldo, eps, relu, s, stat_out);
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

}  // namespace

struct pivp_plan {
    pivp_config_t cfg;
    std::vector<ParamInfo> params;
    // parameter indices
    int i_enc_w[7], i_enc_b[7], i_lstm_w[7], i_lstm_b[7];
    int i_ln_g[9], i_ln_b[9];   // order: norm_enc0, hidden1..hidden7, norm_enc6
    int i_masks_w, i_masks_b, i_cs_w, i_cs_b, i_enc7_w, i_enc7_b;
    int i_head_w, i_head_b, i_head2_w, i_head2_b;  // cdna_kerns | stp_input ; identity_params
    // geometry
    int H2, W2, H4, W4, H8, W8, NP, NE, K5;
    // workspace
    float* ws; long long ws_floats;
    int nslabs;
    std::vector<Slab> slabs;
    size_t o_zero, o_lnpart, o_linpart, o_kerns, o_theta, o_logits, o_enc7, o_layer0, o_masks, o_prevsel,
           o_e0raw, o_e6raw, o_losspart;
    int loss_nparts;
    int last_steps;  // timesteps run by the last rollout
    // optional event timing of the dominant kernel (ConvLSTM gate conv), per layer
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;   // pairs
    std::vector<int> prof_layer;
    size_t prof_used = 0;
    ~pivp_plan() { for (hipEvent_t e : prof_ev) (void)hipEventDestroy(e); }
};

static const float* P(const pivp_plan* p, int idx) { return p->params[idx].ptr; }

extern "C" int pivp_abi_version(void) { return 1; }

extern "C" int pivp_plan_create(const pivp_config_t* cfg, pivp_plan_t** out) {
    if (!cfg || !out) return PIVP_ERR_BADARG;
    if (cfg->batch <= 0 || cfg->seq_len < 2 || cfg->height < 16 || cfg->width < 16) return PIVP_ERR_BADARG;
    if (cfg->height % 8 || cfg->width % 8) return PIVP_ERR_BADARG;
    if (cfg->model_type < 0 || cfg->model_type > 2) return PIVP_ERR_BADARG;
    if (cfg->num_masks < 1 || cfg->num_masks > 11) return PIVP_ERR_BADARG;
    if (cfg->model_type == PIVP_MODEL_DNA && cfg->num_masks != 1) return PIVP_ERR_BADARG;  // TM:389-390
    if (cfg->context_frames < 1 || cfg->context_frames >= cfg->seq_len) return PIVP_ERR_BADARG;
    pivp_plan* p = new pivp_plan();
    p->cfg = *cfg;
    const int H = cfg->height, W = cfg->width, B = cfg->batch;
    p->H2 = H / 2; p->W2 = W / 2; p->H4 = H / 4; p->W4 = W / 4; p->H8 = H / 8; p->W8 = W / 8;
    p->NP = cfg->num_masks + 1;
    p->NE = cfg->model_type == PIVP_MODEL_DNA ? 25 : 3;
    p->K5 = 128 * p->H8 * p->W8;
    p->ws = nullptr; p->ws_floats = 0; p->last_steps = 0;

    auto add = [&](const std::string& name, long long n) { p->params.push_back({name, n, nullptr}); return (int)p->params.size() - 1; };
    const int cin3 = 64 + (cfg->use_state ? 10 : 0);
    const long long encw[7] = {75 * 32, 9 * 32 * 32, 9 * 64 * 64, (long long)cin3 * 64, 9 * 128 * 128, 9 * 96 * 96, 9 * 64 * 64};
    const int encb[7] = {32, 32, 64, 64, 128, 96, 64};
    for (int i = 0; i < 7; ++i) {
        p->i_enc_w[i] = add("enc" + std::to_string(i) + "/W", encw[i]);
        p->i_enc_b[i] = add("enc" + std::to_string(i) + "/b", encb[i]);
    }
    for (int i = 0; i < 7; ++i) {
        const LstmSpec& L = kLstm[i];
        p->i_lstm_w[i] = add(std::string(L.name) + "/conv/W", 25LL * (L.cx + L.C) * 4 * L.C);
        p->i_lstm_b[i] = add(std::string(L.name) + "/conv/b", 4 * L.C);
    }
    const char* lnn[9] = {"norm_enc0", "hidden1", "hidden2", "hidden3", "hidden4", "hidden5", "hidden6", "hidden7", "norm_enc6"};
    const long long lnsz[9] = {32LL * p->H2 * p->W2, 32LL * p->H2 * p->W2, 32LL * p->H2 * p->W2, 64LL * p->H4 * p->W4,
                               64LL * p->H4 * p->W4, 128LL * p->H8 * p->W8, 64LL * p->H4 * p->W4, 32LL * p->H2 * p->W2,
                               64LL * H * W};
    for (int i = 0; i < 9; ++i) {
        p->i_ln_g[i] = add(std::string(lnn[i]) + "/norm/gamma", lnsz[i]);
        p->i_ln_b[i] = add(std::string(lnn[i]) + "/norm/beta", lnsz[i]);
    }
    p->i_masks_w = add("masks/W", 64LL * p->NP);
    p->i_masks_b = add("masks/b", p->NP);
    p->i_cs_w = add("current_state/W", 50);
    p->i_cs_b = add("current_state/b", 5);
    p->i_enc7_w = add("model/enc7/W", 64LL * p->NE);
    p->i_enc7_b = add("model/enc7/b", p->NE);
    p->i_head_w = p->i_head_b = p->i_head2_w = p->i_head2_b = -1;
    if (cfg->model_type == PIVP_MODEL_CDNA) {
        p->i_head_w = add("model/cdna_kerns/W", (long long)p->K5 * 256);
        p->i_head_b = add("model/cdna_kerns/b", 25LL * cfg->num_masks);
    } else if (cfg->model_type == PIVP_MODEL_STP) {
        p->i_head_w = add("model/stp_input/W", (long long)p->K5 * 256);
        p->i_head_b = add("model/stp_input/b", 100);
        p->i_head2_w = add("model/identity_params/W", 600);
        p->i_head2_b = add("model/identity_params/b", 6);
    }

    // ---- workspace carve (offsets in floats, 256-B aligned) ----
    size_t off = 0;
    auto carve = [&](size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return o; };
    const size_t HW = (size_t)H * W, HW2 = (size_t)p->H2 * p->W2, HW4 = (size_t)p->H4 * p->W4, HW8 = (size_t)p->H8 * p->W8;
    p->o_zero = carve((size_t)B * HW2 * 32);
    p->o_lnpart = carve((size_t)B * ln_stats_slices((int)(64 * HW)) * 4);
    p->o_linpart = carve((size_t)cdna_kernel_partials_slices(p->K5) * B * 256);
    p->o_kerns = carve((size_t)B * 25 * cfg->num_masks);
    p->o_theta = carve((size_t)B * 6);
    p->o_logits = carve((size_t)B * p->NP * HW);
    p->o_enc7 = carve((size_t)B * p->NE * HW);
    p->o_layer0 = carve((size_t)B * 3 * HW);
    p->o_masks = carve((size_t)B * p->NP * HW);
    p->o_prevsel = carve((size_t)B * 3 * HW);
    p->o_e0raw = carve((size_t)B * HW2 * 32);
    p->o_e6raw = carve((size_t)B * HW * 64);
    p->loss_nparts = loss_partials_count((int)(B * 3 * HW));
    p->o_losspart = carve((size_t)(cfg->seq_len) * p->loss_nparts);
    p->nslabs = cfg->keep_activations ? cfg->seq_len - 1 : 2;
    p->slabs.resize(p->nslabs);
    const size_t hsz[7] = {HW2 * 32, HW2 * 32, HW4 * 64, HW4 * 64, HW8 * 128, HW4 * 64, HW2 * 32};
    for (int s = 0; s < p->nslabs; ++s) {
        Slab& S = p->slabs[s];
        S.cat7 = carve(B * HW2 * 64); S.n1 = carve(B * HW2 * 32); S.n2 = carve(B * HW2 * 32);
        S.cat6 = carve(B * HW4 * 96); S.n3 = carve(B * HW4 * 64); S.n4 = carve(B * HW4 * 64);
        S.e2 = carve(B * HW8 * 64); S.e3 = carve(B * HW8 * 64); S.n5 = carve(B * HW8 * 128);
        S.e4 = carve(B * HW4 * 128); S.e5 = carve(B * HW2 * 96); S.e6 = carve(B * HW * 64);
        for (int i = 0; i < 7; ++i) { S.h[i] = carve(B * hsz[i]); S.c[i] = carve(B * hsz[i]); }
    }
    p->ws_floats = (long long)off;
    *out = p;
    return PIVP_OK;
}

extern "C" void pivp_plan_destroy(pivp_plan_t* plan) { delete plan; }

extern "C" int pivp_param_count(const pivp_plan_t* plan) { return plan ? (int)plan->params.size() : PIVP_ERR_BADARG; }
extern "C" const char* pivp_param_name(const pivp_plan_t* plan, int idx) {
    if (!plan || idx < 0 || idx >= (int)plan->params.size()) return nullptr;
    return plan->params[idx].name.c_str();
}
extern "C" long long pivp_param_numel(const pivp_plan_t* plan, int idx) {
    if (!plan || idx < 0 || idx >= (int)plan->params.size()) return PIVP_ERR_BADARG;
    return plan->params[idx].numel;
}
extern "C" int pivp_plan_set_param(pivp_plan_t* plan, int idx, const float* dptr) {
    if (!plan || idx < 0 || idx >= (int)plan->params.size() || !dptr) return PIVP_ERR_BADARG;
    plan->params[idx].ptr = dptr;
    return PIVP_OK;
}
extern "C" long long pivp_plan_workspace_bytes(const pivp_plan_t* plan) { return plan ? plan->ws_floats * 4 : PIVP_ERR_BADARG; }
extern "C" int pivp_plan_set_workspace(pivp_plan_t* plan, void* dptr, long long bytes) {
    if (!plan || !dptr || bytes < plan->ws_floats * 4 || ((uintptr_t)dptr & 255)) return PIVP_ERR_BADARG;
    plan->ws = (float*)dptr;
    return PIVP_OK;
}

extern "C" int pivp_reset_state(pivp_plan_t* plan, void* stream) {
    if (!plan || !plan->ws) return PIVP_ERR_STATE;
    const size_t n = (size_t)plan->cfg.batch * plan->H2 * plan->W2 * 32;
    if (hipMemsetAsync(plan->ws + plan->o_zero, 0, n * 4, (hipStream_t)stream) != hipSuccess) return PIVP_ERR_LAUNCH;
    return PIVP_OK;
}

#define RC(call) do { int rc_ = (call); if (rc_ != PIVP_OK) return rc_; } while (0)

static int run_step(pivp_plan* p, int t, const float* prev, const float* action, const float* state_prev,
                    float* gen_out, float* state_out, hipStream_t s) {
    const pivp_config_t& c = p->cfg;
    const int B = c.batch, H = c.height, W = c.width;
    float* ws = p->ws;
    const Slab& S = p->slabs[t % p->nslabs];
    const Slab* Sp = t > 0 ? &p->slabs[(t - 1) % p->nslabs] : nullptr;
    float* lnp = ws + p->o_lnpart;
    const float eps = c.ln_eps;
    auto hp = [&](int i) -> const float* { return Sp ? ws + Sp->h[i] : nullptr; };   // t = 0: h == 0, skipped
    auto cp = [&](int i) { return Sp ? ws + Sp->c[i] : ws + p->o_zero; };
    auto lstm = [&](int i, const float* x, int ldx, int hh, int wwid) {
        const bool prof = p->prof_on && p->prof_used + 2 <= p->prof_ev.size();
        if (prof) (void)hipEventRecord(p->prof_ev[p->prof_used], s);
        int rc = run_convlstm(x, kLstm[i].cx, ldx, hp(i), kLstm[i].C, P(p, p->i_lstm_w[i]), P(p, p->i_lstm_b[i]),
                              cp(i), ws + S.c[i], ws + S.h[i], B, hh, wwid, s);
        if (prof) {
            (void)hipEventRecord(p->prof_ev[p->prof_used + 1], s);
            p->prof_layer[p->prof_used / 2] = i + (Sp ? 0 : 8);   // +8: first-step launch without the h half of K
            p->prof_used += 2;
        }
        return rc;
    };
    auto ln = [&](int j, const float* x, float* out, int n, int C, int ldo, int relu) {
        return run_layernorm(x, P(p, p->i_ln_g[j]), P(p, p->i_ln_b[j]), out, lnp, B, n, C, ldo, eps, relu, s);
    };
    const int n2 = 32 * p->H2 * p->W2, n4 = 64 * p->H4 * p->W4, n8 = 128 * p->H8 * p->W8;

    // group 0 (TM:595): enc0 -> norm_enc0 -> relu   => cat7[:, 32:64]
    RC(conv_enc0(prev, P(p, p->i_enc_w[0]), P(p, p->i_enc_b[0]), ws + p->o_e0raw, B, H, W, s));
    RC(ln(0, ws + p->o_e0raw, ws + S.cat7 + 32, n2, 32, 64, 1));
    // group 1 (TM:596): lstm1 -> hidden1 -> lstm2 -> hidden2 -> enc1 -> relu  => cat6[:, 64:96]
    RC(lstm(0, ws + S.cat7 + 32, 64, p->H2, p->W2));
    RC(ln(1, ws + S.h[0], ws + S.n1, n2, 32, 32, 0));
    RC(lstm(1, ws + S.n1, 32, p->H2, p->W2));
    RC(ln(2, ws + S.h[1], ws + S.n2, n2, 32, 32, 0));
    RC(run_conv3x3s2(ws + S.n2, 32, 32, P(p, p->i_enc_w[1]), P(p, p->i_enc_b[1]), ws + S.cat6 + 64, 32, 96, 1, B, p->H2, p->W2, s));
    // group 2 (TM:597)
    RC(lstm(2, ws + S.cat6 + 64, 96, p->H4, p->W4));
    RC(ln(3, ws + S.h[2], ws + S.n3, n4, 64, 64, 0));
    RC(lstm(3, ws + S.n3, 64, p->H4, p->W4));
    RC(ln(4, ws + S.h[3], ws + S.n4, n4, 64, 64, 0));
    RC(run_conv3x3s2(ws + S.n4, 64, 64, P(p, p->i_enc_w[2]), P(p, p->i_enc_b[2]), ws + S.e2, 64, 64, 1, B, p->H4, p->W4, s));
    // group 3 (TM:598) + state predictor (TM:730)
    RC(enc3_state(ws + S.e2, action, state_prev, P(p, p->i_enc_w[3]), P(p, p->i_enc_b[3]), P(p, p->i_cs_w), P(p, p->i_cs_b),
                  ws + S.e3, state_out, B, p->H8 * p->W8, c.use_state, s));
    // group 4 (TM:599)
    RC(lstm(4, ws + S.e3, 64, p->H8, p->W8));
    RC(ln(5, ws + S.h[4], ws + S.n5, n8, 128, 128, 0));
    RC(run_deconv3x3s2(ws + S.n5, 128, 128, P(p, p->i_enc_w[4]), P(p, p->i_enc_b[4]), ws + S.e4, 128, 128, 1, B, p->H8, p->W8, s));
    // group 5 (TM:600): lstm6 -> hidden6 -> concat(., enc1) -> enc5 -> relu
    RC(lstm(5, ws + S.e4, 128, p->H4, p->W4));
    RC(ln(6, ws + S.h[5], ws + S.cat6, n4, 64, 96, 0));
    RC(run_deconv3x3s2(ws + S.cat6, 96, 96, P(p, p->i_enc_w[5]), P(p, p->i_enc_b[5]), ws + S.e5, 96, 96, 1, B, p->H4, p->W4, s));
    // group 6 (TM:601): lstm7 -> hidden7 -> concat(., enc0) -> enc6 -> norm_enc6 -> relu
    RC(lstm(6, ws + S.e5, 96, p->H2, p->W2));
    RC(ln(7, ws + S.h[6], ws + S.cat7, n2, 32, 64, 0));
    RC(run_deconv3x3s2(ws + S.cat7, 64, 64, P(p, p->i_enc_w[6]), P(p, p->i_enc_b[6]), ws + p->o_e6raw, 64, 64, 0, B, p->H2, p->W2, s));
    RC(ln(8, ws + p->o_e6raw, ws + S.e6, 64 * H * W, 64, 64, 1));
    // heads (TM:711-728)
    RC(heads_1x1(ws + S.e6, P(p, p->i_masks_w), P(p, p->i_masks_b), P(p, p->i_enc7_w), P(p, p->i_enc7_b),
                 ws + p->o_logits, ws + p->o_enc7, ws + p->o_layer0, B, H * W, p->NP, p->NE, c.model_type, s));
    const float* aux = nullptr;
    if (c.model_type == PIVP_MODEL_CDNA) {
        RC(cdna_kernels(ws + S.n5, P(p, p->i_head_w), P(p, p->i_head_b), ws + p->o_linpart, ws + p->o_kerns, B, p->K5, c.num_masks, s));
        aux = ws + p->o_kerns;
    } else if (c.model_type == PIVP_MODEL_STP) {
        RC(stp_params(ws + S.n5, P(p, p->i_head_w), P(p, p->i_head_b), P(p, p->i_head2_w), P(p, p->i_head2_b),
                      ws + p->o_linpart, ws + p->o_theta, B, p->K5, s));
        aux = ws + p->o_theta;
    } else {
        aux = ws + p->o_enc7;
    }
    RC(composite(prev, ws + p->o_logits, ws + p->o_layer0, aux, gen_out, ws + p->o_masks, B, H, W, c.num_masks,
                 c.model_type, c.stp_zero_border, s));
    return PIVP_OK;
}

extern "C" int pivp_rollout_forward(pivp_plan_t* plan, const float* images, const float* actions, const float* states,
                                    const unsigned char* gt_select, float* gen_images, float* gen_states, float* results,
                                    void* stream) {
    if (!plan || !images || !actions || !states || !gen_images || !gen_states || !results) return PIVP_ERR_BADARG;
    if (!plan->ws) return PIVP_ERR_STATE;
    for (const ParamInfo& pi : plan->params) if (!pi.ptr) return PIVP_ERR_STATE;
    hipStream_t s = (hipStream_t)stream;
    const pivp_config_t& c = plan->cfg;
    const int B = c.batch, T = c.seq_len, ctx = c.context_frames;
    const size_t fr = (size_t)B * 3 * c.height * c.width;
    for (int t = 0; t < T - 1; ++t) {
        const float* prev;
        if (t < ctx) {
            prev = images + t * fr;                                    // TM:672-673
        } else if (!gt_select) {
            prev = gen_images + (t - 1) * fr;                          // TM:664-666
        } else {                                                       // TM:667-670
            float* sel = plan->ws + plan->o_prevsel;
            hipLaunchKernelGGL(select_frames_kernel, dim3(8, B), dim3(256), 0, s, images + t * fr, gen_images + (t - 1) * fr,
                               gt_select + (size_t)t * B, sel, (int)(fr / B));
            prev = sel;
        }
        const float* st_prev = t == 0 ? states : gen_states + (size_t)(t - 1) * B * 5;   // TM:646, TM:730
        RC(run_step(plan, t, prev, actions + (size_t)t * B * 5, st_prev, gen_images + t * fr,
                    gen_states + (size_t)t * B * 5, s));
    }
    plan->last_steps = T - 1;
    // loss (TM:737-759): frames ctx..T-1 vs gen[ctx-1..T-2]
    const int nf = T - ctx;
    float* lp = plan->ws + plan->o_losspart;
    for (int i = 0; i < nf; ++i)
        RC(frame_sqerr_partials(images + (size_t)(ctx + i) * fr, gen_images + (size_t)(ctx - 1 + i) * fr,
                                lp + (size_t)i * plan->loss_nparts, (int)fr, s));
    RC(loss_finalize(lp, plan->loss_nparts, nf, (int)fr, states + (size_t)ctx * B * 5, gen_states + (size_t)(ctx - 1) * B * 5,
                     B * 5, (float)(T - ctx), results, s));
    return PIVP_OK;
}

extern "C" int pivp_plan_set_profiling(pivp_plan_t* plan, int enable) {
    if (!plan) return PIVP_ERR_BADARG;
    if (enable && plan->prof_ev.empty()) {
        const size_t n = (size_t)2 * 7 * (plan->cfg.seq_len - 1);
        plan->prof_ev.resize(n);
        plan->prof_layer.assign(n / 2, 0);
        for (size_t i = 0; i < n; ++i)
            if (hipEventCreate(&plan->prof_ev[i]) != hipSuccess) { plan->prof_ev.resize(i); return PIVP_ERR_LAUNCH; }
    }
    plan->prof_on = enable != 0;
    plan->prof_used = 0;
    return PIVP_OK;
}

extern "C" int pivp_plan_profile_read(pivp_plan_t* plan, double* ms_per_layer, int* launches_per_layer, double* flops_per_layer) {
    if (!plan || !ms_per_layer || !launches_per_layer || !flops_per_layer) return PIVP_ERR_BADARG;
    const pivp_config_t& c = plan->cfg;
    double fl_full[7], fl_first[7];
    for (int i = 0; i < 7; ++i) {
        ms_per_layer[i] = 0.0; launches_per_layer[i] = 0; flops_per_layer[i] = 0.0;
        const int lv = kLstm[i].level;
        const double M = (double)c.batch * (c.height / lv) * (c.width / lv);
        fl_full[i] = 2.0 * M * 4.0 * kLstm[i].C * 25.0 * (kLstm[i].cx + kLstm[i].C);
        fl_first[i] = 2.0 * M * 4.0 * kLstm[i].C * 25.0 * kLstm[i].cx;     // executed flops when h == 0 is skipped
    }
    for (size_t k = 0; k + 1 < plan->prof_used; k += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, plan->prof_ev[k], plan->prof_ev[k + 1]) != hipSuccess) return PIVP_ERR_STATE;
        const int tag = plan->prof_layer[k / 2];
        const int L = tag & 7;
        ms_per_layer[L] += ms; launches_per_layer[L] += 1;
        flops_per_layer[L] += (tag & 8) ? fl_first[L] : fl_full[L];   // SUM of executed flops over the launches
    }
    plan->prof_used = 0;
    return PIVP_OK;
}

extern "C" long long pivp_get_tap(pivp_plan_t* plan, const char* name, int step, float* out, void* stream) {
    if (!plan || !name || !out || !plan->ws) return PIVP_ERR_BADARG;
    if (step < 0 || step >= plan->last_steps) return PIVP_ERR_BADARG;
    if (!plan->cfg.keep_activations && step < plan->last_steps - 2) return PIVP_ERR_STATE;
    hipStream_t s = (hipStream_t)stream;
    const pivp_config_t& c = plan->cfg;
    const int B = c.batch;
    const Slab& S = plan->slabs[step % plan->nslabs];
    float* ws = plan->ws;
    const int HW = c.height * c.width, HW2 = plan->H2 * plan->W2, HW4 = plan->H4 * plan->W4, HW8 = plan->H8 * plan->W8;
    struct T { const char* n; size_t off; int C, hw, ld; };
    const T taps[] = {
        {"enc0", S.cat7 + 32, 32, HW2, 64}, {"enc1", S.cat6 + 64, 32, HW4, 96}, {"enc2", S.e2, 64, HW8, 64},
        {"enc3", S.e3, 64, HW8, 64}, {"enc4", S.e4, 128, HW4, 128}, {"enc5", S.e5, 96, HW2, 96}, {"enc6", S.e6, 64, HW, 64},
        {"hidden1", S.n1, 32, HW2, 32}, {"hidden2", S.n2, 32, HW2, 32}, {"hidden3", S.n3, 64, HW4, 64},
        {"hidden4", S.n4, 64, HW4, 64}, {"hidden5", S.n5, 128, HW8, 128}, {"hidden6", S.cat6, 64, HW4, 96},
        {"hidden7", S.cat7, 32, HW2, 64},
        {"lstm1_h", S.h[0], 32, HW2, 32}, {"lstm2_h", S.h[1], 32, HW2, 32}, {"lstm3_h", S.h[2], 64, HW4, 64},
        {"lstm4_h", S.h[3], 64, HW4, 64}, {"lstm5_h", S.h[4], 128, HW8, 128}, {"lstm6_h", S.h[5], 64, HW4, 64},
        {"lstm7_h", S.h[6], 32, HW2, 32},
        {"lstm1_c", S.c[0], 32, HW2, 32}, {"lstm2_c", S.c[1], 32, HW2, 32}, {"lstm3_c", S.c[2], 64, HW4, 64},
        {"lstm4_c", S.c[3], 64, HW4, 64}, {"lstm5_c", S.c[4], 128, HW8, 128}, {"lstm6_c", S.c[5], 64, HW4, 64},
        {"lstm7_c", S.c[6], 32, HW2, 32}};
    for (const T& t : taps) {
        if (strcmp(t.n, name) == 0) {
            int rc = nhwc_to_nchw(ws + t.off, out, B, t.C, t.hw, t.ld, s);
            return rc == PIVP_OK ? (long long)B * t.C * t.hw : rc;
        }
    }
    // planar buffers exist only for the most recent step
    size_t off = 0, n = 0;
    if (strcmp(name, "enc7") == 0) { off = plan->o_enc7; n = (size_t)B * plan->NE * HW; }
    else if (strcmp(name, "masks") == 0) { off = plan->o_masks; n = (size_t)B * plan->NP * HW; }
    else if (strcmp(name, "cdna_kerns") == 0 && c.model_type == PIVP_MODEL_CDNA) { off = plan->o_kerns; n = (size_t)B * 25 * c.num_masks; }
    else if (strcmp(name, "stp_theta") == 0 && c.model_type == PIVP_MODEL_STP) { off = plan->o_theta; n = (size_t)B * 6; }
    else return PIVP_ERR_BADARG;
    if (step != plan->last_steps - 1) return PIVP_ERR_STATE;
    if (hipMemcpyAsync(out, ws + off, n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return PIVP_ERR_LAUNCH;
    return (long long)n;
}

// ---------------------------------------------------------------------------------------------
// per-op entry points
// ---------------------------------------------------------------------------------------------
extern "C" int pivp_convlstm(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* bias,
                             const float* c_in, float* c_out, float* h_out, int B, int H, int W, void* stream) {
    if (!x || !w || !bias || !c_in || !c_out || !h_out) return PIVP_ERR_BADARG;
    return run_convlstm(x, cx, ldx, h_prev, C, w, bias, c_in, c_out, h_out, B, H, W, (hipStream_t)stream);
}
extern "C" int pivp_convlstm_v(const float* x, int cx, int ldx, const float* h_prev, int C, const float* w, const float* bias,
                               const float* c_in, float* c_out, float* h_out, int B, int H, int W, int variant, void* stream) {
    if (!x || !w || !bias || !c_in || !c_out || !h_out || variant < 0 || variant > 13) return PIVP_ERR_BADARG;
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
    hipStream_t s = (hipStream_t)stream;
    const int Hout = mode ? 2 * Hin : Hin / 2, Wout = mode ? 2 * Win : Win / 2;
    int rc = PIVP_OK;
    if (dx) {
        rc = repack_transpose(w, wt, 9, cin, cout, 0, s);
        if (rc != PIVP_OK) return rc;
        rc = mode ? run_conv3x3s2(dy, cout, ldy, wt, nullptr, dx, cin, lddx, 0, B, Hout, Wout, s, accum_dx)
                  : run_deconv3x3s2(dy, cout, ldy, wt, nullptr, dx, cin, lddx, 0, B, Hout, Wout, s, accum_dx);
        if (rc != PIVP_OK) return rc;
    }
    rc = run_wgrad(mode, x, cin, ldx, nullptr, 0, 0, cin, dy, ldy, cout, dW, B, Hin, Win, Hout, Wout, 3, 1, 2, s);
    if (rc != PIVP_OK) return rc;
    return bias_grad(dy, ldy, cout, B * Hout * Wout, db, s);
}
extern "C" int pivp_layernorm_train(const float* x, const float* gamma, const float* beta, float* out, float* partials, float* stat,
                                    int B, int n, int C, int ldo, float eps, int relu, void* stream) {
    if (!stat) return PIVP_ERR_BADARG;
    return run_layernorm(x, gamma, beta, out, partials, B, n, C, ldo, eps, relu, (hipStream_t)stream, stat);
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
extern "C" int pivp_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr_t, double beta1, double beta2,
                              double eps, double gscale, void* stream) {
    return adam_step(p, g, m, v, (long)n, lr_t, beta1, beta2, eps, gscale, (hipStream_t)stream);
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
    return (long long)cdna_kernel_partials_slices(K) * B * 256;
}
extern "C" int pivp_cdna_kernels(const float* hidden5, const float* wt, const float* bias, float* partials, float* kerns,
                                 int B, int K, int num_masks, void* stream) {
    return cdna_kernels(hidden5, wt, bias, partials, kerns, B, K, num_masks, (hipStream_t)stream);
}
extern "C" int pivp_stp_params(const float* hidden5, const float* wt1, const float* b1, const float* w2, const float* b2,
                               float* partials, float* theta, int B, int K, void* stream) {
    return stp_params(hidden5, wt1, b1, w2, b2, partials, theta, B, K, (hipStream_t)stream);
}
extern "C" int pivp_composite(const float* prev, const float* mask_logits, const float* layer0, const float* aux, float* out,
                              float* masks_out, int B, int H, int W, int num_masks, int model_type, int stp_zero_border,
                              void* stream) {
    return composite(prev, mask_logits, layer0, aux, out, masks_out, B, H, W, num_masks, model_type, stp_zero_border,
                     (hipStream_t)stream);
}
extern "C" int pivp_select_frames(const float* ground_truth, const float* generated, const unsigned char* take_gt, float* out,
                                  int B, int frame_numel, void* stream) {
    if (!ground_truth || !generated || !take_gt || !out || B <= 0 || frame_numel <= 0 || frame_numel % 4) return PIVP_ERR_BADARG;
    hipLaunchKernelGGL(select_frames_kernel, dim3(8, B), dim3(256), 0, (hipStream_t)stream, ground_truth, generated, take_gt,
                       out, frame_numel);
    return PIVP_LAUNCH_STATUS();
}
