// The plan: Model.__init__ / reset_state / __call__ of the reference (TM:484-764) and the backward pass that
// Chainer's autograd performs under optimizer.update (TM:950), sequenced in native code on one HIP stream.
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pivp_hip.h"
#include "pivp_host.h"

using namespace pivp;

namespace {

struct LstmSpec { const char* name; int cx; int C; int level; };  // level: 2 -> H/2, 4 -> H/4, 8 -> H/8
const LstmSpec kLstm[7] = {
    {"lstm1", 32, 32, 2}, {"lstm2", 32, 32, 2}, {"lstm3", 32, 64, 4}, {"lstm4", 64, 64, 4},
    {"lstm5", 64, 128, 8}, {"lstm6", 128, 64, 4}, {"lstm7", 96, 32, 2}};

struct ParamInfo { std::string name; long long numel; const float* ptr; float* grad; int group; };

// group of a parameter = position of its layer in the backward sweep of one timestep (include/pivp_hip.h, PIVP_GRAD_GROUPS)
static int grad_group_of(const std::string& n) {
    auto starts = [&](const char* p) { return n.rfind(p, 0) == 0; };
    if (starts("masks/") || starts("model/") || starts("norm_enc6/") || starts("enc6/")) return 0;
    if (starts("hidden7/") || starts("lstm7/")) return 1;
    if (starts("enc5/") || starts("hidden6/") || starts("lstm6/")) return 2;
    if (starts("enc4/") || starts("hidden5/") || starts("lstm5/")) return 3;
    if (starts("enc3/") || starts("current_state/") || starts("enc2/") || starts("hidden4/") || starts("lstm4/") ||
        starts("hidden3/") || starts("lstm3/")) return 4;
    return 5;   // enc1, hidden2, lstm2, hidden1, lstm1, norm_enc0, enc0
}

// Per-timestep activations (offsets in floats from the workspace base).  Two rolling slabs for inference, T-1 slabs
// when keep_activations (BPTT needs every step).
// LayerNorm partial slots per sample: the ln_stats slices of the largest map (64 channels at full resolution) or the tiles
// of the producers that write partials themselves (enc6 through igemm_small: HW/4 anchors / 32 x 2 column blocks x 4 parities)
static int ln_partial_cap(int HW) {
    const int a = ln_stats_slices(64 * HW), b = (HW / 4 / 32 + 1) * 2 * 4;
    return a > b ? a : b;
}

// ConvLSTM weight gradients can be taken over a batch of up to this many timesteps in ONE launch (the reduction runs over pixels and
// timesteps alike; WgradDesc::tcount): a quarter of the block epilogues and grid ramps.  Measured at B = 32 it does not pay: alone
// it is worth 0.4 ms of a 35.6 ms train step, and next to the side stream it LOSES 0.7 ms (34.1 vs 33.4 ms) because the weight
// gradients then arrive in bursts instead of filling the gaps of every step.  So the default is one timestep per launch
// (PIVP_WGRAD_BATCH=1), the dG rings then simply double-buffer; the batched path stays for larger per-GPU batches and is tested
// (tests/test_gpu_train.py).  Timestep 0 is always its own batch (its h_{-1} = 0 half is skipped).
constexpr size_t ENC_RING_BYTES_MAX = (size_t)2 << 30;   // the five stride-2 3x3 layers' dY rings together (pivp_plan::eg_cap is cut to fit)
constexpr int WG_BATCH_MAX = 8;   // most timesteps one ConvLSTM weight-gradient launch can take (dG ring slots per ring: pivp_plan::wg_cap <= this)

struct Slab {
    size_t cat7, n1, n2, cat6, n3, n4, e2, e3, n5, e4, e5, e6;   // NHWC feature maps (cat7 = [hidden7|enc0], cat6 = [hidden6|enc1])
    size_t h[7], c[7];                                           // ConvLSTM states
    size_t e0raw, e6raw;                                         // LayerNorm inputs of norm_enc0 / norm_enc6
    size_t gates[7];                                             // gate activations [M][4C] (training)
    size_t lnstat;                                               // [9][B][2] mean, rstd (training)
    size_t logits, enc7, layer0, kerns, vpre, theta;             // head tensors of the step
    size_t prevsel;                                              // scheduled-sampling input frame of the step
};

struct Grads {   // gradient workspace (single copy, reused by every timestep of the backward sweep)
    size_t cat7, n2, n4, n5, e6;
    size_t e0raw[2], dv[2];   // dY tensors the side stream's weight gradients read, by timestep parity: a step rewrites the buffer the weight gradients of
                              // TWO steps ago read, so the main stream never waits for the previous step's (round 5)
    // The dY tensors of the five stride-2 3x3 layers live in RINGS like the ConvLSTMs' dG: 2 rings x eg_cap timesteps, one weight-gradient launch per
    // batch (wgrad3x3s2.hip).  enc6: d e6raw; enc2: d e2; enc1: columns 64.. of d cat6; enc5 / enc4: the x columns of lstm7's / lstm6's input gradient,
    // din[6] / din[5] -- so those two hold ring slots as well (the other cells' din: two buffers by timestep parity).  *_sz: floats per slot.
    size_t cat6, e2, e6raw, cat6_sz, e2_sz, e6raw_sz;
    size_t din[7], din_sz[7], dc[7];
    size_t dG[7], go, dmk, dz, dkpart, dstate, lnpart;   // dG: gate pre-activation gradients per ConvLSTM: 2 rings x wg_cap timesteps (batched weight gradients)
                                                             // go: d loss / d gen[t] for every t ([T-1] frames: the loss terms of all of them come from ONE launch)
    size_t wt_lstm[7], wt_enc[7];   // re-packed (transposed) weights for the data gradients, rebuilt once per backward
    size_t ln_ppart[9], ln_ppart_floats;   // per-norm partial parameter gradients (ln_backward's param_part), one contiguous region
    size_t wg_part[5], wg_part_floats;   // per-block partial weight gradients of enc6, enc5, enc4, enc2, enc1 (WgradDesc::part), one contiguous region
    size_t wtb_lstm[7];             // ... and their bf16 packs (bf16 precision mode)
    size_t dg_absmax;               // fp16-piece data gradients: the partial maxima of the dG in front of the launch (absmax_partials; stream-ordered, one buffer)
};

}  // namespace

struct pivp_plan {
    pivp_config_t cfg;
    std::vector<ParamInfo> params;
    int i_enc_w[7], i_enc_b[7], i_lstm_w[7], i_lstm_b[7];
    int i_ln_g[9], i_ln_b[9];   // order: norm_enc0, hidden1..hidden7, norm_enc6
    int i_masks_w, i_masks_b, i_cs_w, i_cs_b, i_enc7_w, i_enc7_b;
    int i_head_w, i_head_b, i_head2_w, i_head2_b;
    int H2, W2, H4, W4, H8, W8, NP, NE, K5;
    float* ws; long long ws_floats;
    int nslabs;
    std::vector<Slab> slabs;
    Grads g;
    bool has_grads;
    size_t o_zero, o_lnpart, o_lnpart2, o_linpart, o_masks, o_losspart;   // o_lnpart2: enc6 reads hidden7's partials while writing its own
    size_t o_wabs[2];                 // precision mode FP16X3: absmax_partials of the enc5 / enc6 weights (rebuilt at the start of a rollout)
    size_t o_wbf16[7];                // bf16 packs of the ConvLSTM weights (pivp_plan_set_precision), rebuilt at the start of a rollout
    int lstm_bf16 = 0;                // 1: bf16 operands in the ConvLSTM forward (precision modes BF16 and BF16X3)
    int precision = 0;                // PIVP_PRECISION_*
    int pack_cache = 0, packs_valid = 0;   // pivp_plan_set_pack_cache: keep the precision modes' weight packs across rollouts until pivp_plan_params_changed
    int bwd_planes = 1;               // the data gradients' form of lstm_planes
    bool x3_wgrad = false;            // fp16x3 mode: the ConvLSTM weight gradients with two fp16 pieces too (wgrad25_bf16_kernel<.., 2>, batched like the bf16 mode's)
    bool x6_wgrad = false;            // bf16x6 mode: ... with three bf16 pieces (wgrad25_bf16_kernel<.., 3>), same schedule
    int lstm_planes = 1;              // 2: split mode (hi / lo planes, three MFMAs per product); 3: three pieces, six MFMAs (forward gate convs only: the
                                      // backward sweep and every other op of that mode are the fp32 ones)
    int main_prio = -1;               // pivp_plan_set_main_priority: -1 = on unless a gradient listener is registered (data parallelism), 0 / 1 = as said
    int bf16_all = 0;                 // precision mode BF16: also the ConvLSTM gradients and the enc5 / enc6 transposed convs
    pivp_grad_group_cb grad_cb = nullptr; void* grad_cb_user = nullptr;   // gradient-group-final notifications (t = 0 sweep)
    int loss_nparts;
    int last_steps;
    bool last_sched;                  // last forward used scheduled sampling (frames detached, TM:669-670)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    std::vector<int> prof_layer;
    size_t prof_used = 0;
    // Weight gradients run on a second, lower-priority stream next to the backward sweep's critical path (SideFork, pivp_host.h):
    // slots 0..6 = the ConvLSTMs, 7..11 = enc6, enc5, enc4, enc2, enc1.  Created on first use; PIVP_SIDE_STREAM=0 (read when
    // the plan is created) keeps everything on the caller's stream.
    static constexpr int NSLOT = 14;      // 12: enc0's weight gradient, 13: the motion head's Linear (cdna_kernels / stp_input)
    bool use_side = true;
    bool ln_fold_train = true;          // training: the norms of hidden2 / hidden4 applied (and written) by enc1 / enc2's launches (PIVP_LN_FOLD_TRAIN=0: ln_apply launches)
    bool fuse_enc3 = true;              // inference: enc3 + state predictor in enc2's epilogue (PIVP_FUSE_ENC3=0: their own launch)
    bool rider = true;                  // the motion head's finisher rides behind enc5's tiles (PIVP_FINISH_RIDER=0: inside frame_head, as rounds 4-5)
    hipStream_t side = nullptr;
    hipStream_t side_of(int) const { return side; }      // (a second side stream for the odd slots, round 3: fp32 no change, bf16 12.25 -> 12.05 ms, but with two
                                                         // processes on one GPU the step went from 65 ms to 78 SECONDS -- the hardware queues oversubscribe)
    hipEvent_t ev_ready[NSLOT][2] = {}, ev_done[NSLOT][2] = {};      // slots 7..11 (the stride-2 3x3 layers): by dY ring; 12, 13: by timestep parity (Grads::cat6, e0raw)
    hipEvent_t ev_ring_done[7][2] = {};        // ConvLSTM slots: one `done` per dG ring (slots 0..6 of ev_done are unused)
    int wg_cap = 1;                            // dG ring slots per ring = min(T - 2, WG_BATCH_MAX), fixed when the workspace is laid out
    int wg_batch = 1;                          // timesteps per weight-gradient launch (<= wg_cap)
    const float* wg_x[7] = {}; const float* wg_h[7] = {};   // operands of the first timestep of the open batch
    bool group_join = true;                                 // pivp_plan_set_group_join
    bool ln_touched[9] = {};                                // norms whose partial parameter gradients still await their reduction
    WgradDesc enc_desc[5]; bool enc_desc_valid[5] = {};     // enc6, enc5, enc4, enc2, enc1: what this sweep launched (for the reduction of the partial sums)
    int eg_cap = 1;                                         // slots per enc dY ring (Grads::cat6): min(T - 2, WG_BATCH_MAX), cut to ENC_RING_BYTES_MAX
    int enc_cnt[5] = {};                                    // timesteps in each layer's open batch (enc6 only counts the steps a frame gradient reaches)
    const float* enc_x0[5] = {};                            // ... and the forward input of the batch's first timestep
    bool enc_started[5] = {};                               // the layer has launched in this sweep: its partial planes hold sums (before: stored, not added)
    // the side stream(s) and every fork / join event, back to "never created" (the destructor; ensure_side's partial-failure path)
    void destroy_side() {
        for (int i = 0; i < 7; ++i) for (int r = 0; r < 2; ++r) if (ev_ring_done[i][r]) { (void)hipEventDestroy(ev_ring_done[i][r]); ev_ring_done[i][r] = nullptr; }
        for (int i = 0; i < NSLOT; ++i) {
            for (int r = 0; r < 2; ++r) {
                if (ev_ready[i][r]) { (void)hipEventDestroy(ev_ready[i][r]); ev_ready[i][r] = nullptr; }
                if (ev_done[i][r]) { (void)hipEventDestroy(ev_done[i][r]); ev_done[i][r] = nullptr; }
            }
        }
        if (side) { (void)hipStreamDestroy(side); side = nullptr; }
    }
    ~pivp_plan() {
        if (side) (void)hipStreamSynchronize(side);      // a sweep that failed half-way may have left weight-gradient kernels in flight
        for (hipEvent_t e : prof_ev) (void)hipEventDestroy(e);
        destroy_side();
    }
};

static const float* P(const pivp_plan* p, int idx) { return p->params[idx].ptr; }
static float* G(const pivp_plan* p, int idx) { return p->params[idx].grad; }

// ---- workspace carve (offsets in floats, 256-B aligned).  Runs at pivp_plan_create and again from pivp_plan_set_precision while no
// workspace is bound: the ConvLSTM dG rings hold wg_cap timesteps per ring, and only the bf16 mode (or PIVP_WGRAD_BATCH) batches its weight
// gradients -- an fp32 plan double-buffers with ONE slot per ring instead of 8 (2.5 GB less at 128 x 128, B = 32, T = 20). ----
// PIVP_WGRAD_BATCH (tests, measurements): timesteps per ConvLSTM weight-gradient launch, whatever the precision mode; 0 = the mode's own choice
static int wgrad_batch_env() {
    static const int v = [] { const char* e = getenv("PIVP_WGRAD_BATCH"); return e ? atoi(e) : 0; }();
    return v;
}
// dG ring slots per ring: timesteps t = T-2 .. 1 batch (t = 0, no h input, goes alone), as many slots as a launch may take timesteps.
// (fp16-piece / three-piece weight gradients: TWO timesteps per launch on half the CUs -- measured grid, profiles/r04/fp16x3_train_wgrad_batch_slots.txt:
// 8 per launch on every CU, what the bf16 mode does, leaves all of that work to the end of the sweep: 16.85 ms against 16.34)
static int wg_cap_of(const pivp_plan* p) {
    const int T = p->cfg.seq_len, e = wgrad_batch_env();
    const int want = e ? e : (p->bf16_all ? WG_BATCH_MAX : (p->x3_wgrad || p->x6_wgrad) ? 2 : 1);
    int cap = T - 2 < 1 ? 1 : (T - 2 > WG_BATCH_MAX ? WG_BATCH_MAX : T - 2);
    if (want < cap) cap = want < 1 ? 1 : want;
    return cap;
}
static void plan_layout(pivp_plan* p) {
    const pivp_config_t* cfg = &p->cfg;
    const int H = cfg->height, W = cfg->width, B = cfg->batch, T = cfg->seq_len;
    const int cin3 = 64 + (cfg->use_state ? 10 : 0);
    const long long encw[7] = {75 * 32, 9 * 32 * 32, 9 * 64 * 64, (long long)cin3 * 64, 9 * 128 * 128, 9 * 96 * 96, 9 * 64 * 64};
    const long long lnsz[9] = {32LL * p->H2 * p->W2, 32LL * p->H2 * p->W2, 32LL * p->H2 * p->W2, 64LL * p->H4 * p->W4,
                               64LL * p->H4 * p->W4, 128LL * p->H8 * p->W8, 64LL * p->H4 * p->W4, 32LL * p->H2 * p->W2,
                               64LL * H * W};
    size_t off = 0;
    auto carve = [&](size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return o; };
    const size_t HW = (size_t)H * W, HW2 = (size_t)p->H2 * p->W2, HW4 = (size_t)p->H4 * p->W4, HW8 = (size_t)p->H8 * p->W8;
    const size_t hsz[7] = {HW2 * 32, HW2 * 32, HW4 * 64, HW4 * 64, HW8 * 128, HW4 * 64, HW2 * 32};
    const bool train = cfg->keep_activations != 0;
    p->o_zero = carve((size_t)B * HW2 * 32);
    p->o_lnpart = carve((size_t)B * ln_partial_cap((int)HW) * 4);
    p->o_lnpart2 = carve((size_t)B * ln_partial_cap((int)HW) * 4);
    p->o_linpart = carve((size_t)motion_partials_floats(B, p->K5));      // [B][K slices][256] + the tail frame_head_kernel's finisher may read
    p->o_masks = carve((size_t)B * p->NP * HW);
    p->loss_nparts = loss_partials_count((int)(B * 3 * HW));
    p->o_losspart = carve((size_t)T * p->loss_nparts);
    p->o_wabs[0] = carve(128); p->o_wabs[1] = carve(128);
    for (int i = 0; i < 7; ++i)       // 2-byte elements in a float-counted workspace
        p->o_wbf16[i] = carve(lstm_bf16_weight_elems(kLstm[i].cx + kLstm[i].C, 4 * kLstm[i].C) * 3 / 2 + 64);   // room for the three planes of PIVP_PRECISION_BF16X6
    p->nslabs = train ? T - 1 : 2;
    p->wg_cap = wg_cap_of(p);
    p->eg_cap = T - 2 < 1 ? 1 : (T - 2 > WG_BATCH_MAX ? WG_BATCH_MAX : T - 2);      // whatever the precision mode: every mode batches these
    if (train) {
        // ... bounded by bytes: a ring slot holds the dY of all five layers (60 MB at B = 32 on 64 x 64 frames, 240 MB on 128 x 128), and there are
        // 2 x eg_cap of them.  ENC_RING_BYTES_MAX keeps config 2 at 8 timesteps per launch (0.97 GB) and gives config 5 four (1.9 GB instead of
        // 3.9 GB; 4 against 8 per launch measured 11.22 against 11.10 ms on config 3, profiles/r05/NOTES.md).  include/pivp_hip.h documents the growth.
        const size_t r64 = 63;
        const size_t slot_floats = (((size_t)B * HW4 * 96 + r64) & ~r64) + (((size_t)B * HW8 * 64 + r64) & ~r64) + (((size_t)B * HW * 64 + r64) & ~r64) +
                                   (((size_t)B * HW4 * (kLstm[5].cx + kLstm[5].C) + r64) & ~r64) + (((size_t)B * HW2 * (kLstm[6].cx + kLstm[6].C) + r64) & ~r64);
        while (p->eg_cap > 1 && (size_t)2 * p->eg_cap * slot_floats * 4 > ENC_RING_BYTES_MAX) --p->eg_cap;
    }
    p->slabs.resize(p->nslabs);
    for (int s = 0; s < p->nslabs; ++s) {
        Slab& S = p->slabs[s];
        S.cat7 = carve(B * HW2 * 64); S.n1 = carve(B * HW2 * 32); S.n2 = carve(B * HW2 * 32);
        S.cat6 = carve(B * HW4 * 96); S.n3 = carve(B * HW4 * 64); S.n4 = carve(B * HW4 * 64);
        S.e2 = carve(B * HW8 * 64); S.e3 = carve(B * HW8 * 64); S.n5 = carve(B * HW8 * 128);
        S.e4 = carve(B * HW4 * 128); S.e5 = carve(B * HW2 * 96); S.e6 = carve(B * HW * 64);
        for (int i = 0; i < 7; ++i) { S.h[i] = carve(B * hsz[i]); S.c[i] = carve(B * hsz[i]); }
        S.e0raw = carve(B * HW2 * 32); S.e6raw = carve(B * HW * 64);
        for (int i = 0; i < 7; ++i) S.gates[i] = train ? carve(B * hsz[i] * 4) : 0;
        S.lnstat = carve((size_t)9 * B * 2);
        S.logits = carve((size_t)B * p->NP * HW); S.enc7 = carve((size_t)B * p->NE * HW); S.layer0 = carve((size_t)B * 3 * HW);
        S.kerns = carve((size_t)B * 25 * cfg->num_masks); S.vpre = carve((size_t)B * 256); S.theta = carve((size_t)B * 6);
        S.prevsel = carve((size_t)B * 3 * HW);
    }
    p->has_grads = train;
    if (p->has_grads) {
        Grads& g = p->g;
        g.cat7 = carve(B * HW2 * 64); g.n2 = carve(B * HW2 * 32); g.n4 = carve(B * HW4 * 64);
        g.n5 = carve(B * HW8 * 128); g.e6 = carve(B * HW * 64);
        for (int r = 0; r < 2; ++r) { g.e0raw[r] = carve(B * HW2 * 32); g.dv[r] = carve((size_t)B * 256); }
        const size_t nq = (size_t)2 * p->eg_cap;      // slots per enc dY ring pair
        auto slots = [&](size_t n, size_t count, size_t& sz) { sz = (n + 63) / 64 * 64; return carve(sz * count); };
        g.cat6 = slots(B * HW4 * 96, nq, g.cat6_sz); g.e2 = slots(B * HW8 * 64, nq, g.e2_sz); g.e6raw = slots(B * HW * 64, nq, g.e6raw_sz);
        for (int i = 0; i < 7; ++i) {
            const size_t M = hsz[i] / kLstm[i].C * B;
            g.dc[i] = carve(B * hsz[i]);   // (d h of a cell is never materialised: the LayerNorm backward is folded into the gate backward)
            g.din[i] = slots(M * (kLstm[i].cx + kLstm[i].C), i >= 5 ? nq : 2, g.din_sz[i]);
            g.dG[i] = carve(M * 4 * kLstm[i].C * 2 * p->wg_cap);
            g.wt_lstm[i] = carve((size_t)25 * (kLstm[i].cx + kLstm[i].C) * 4 * kLstm[i].C);
            g.wtb_lstm[i] = carve(lstm_bf16_weight_elems(4 * kLstm[i].C, conv5x5_bf16_rows(kLstm[i].cx + kLstm[i].C)) * 3 / 2 + 64);   // up to three planes
            g.wt_enc[i] = (i == 0 || i == 3) ? 0 : carve((size_t)encw[i]);
        }
        g.dg_absmax = carve((size_t)7 * 2 * p->wg_cap * 72);     // dG's partial maxima per (cell, ring, slot): the fp16-piece gradients' scales
        g.go = carve((size_t)(T - 1) * B * 3 * HW);
        g.dmk = carve((size_t)B * p->NP * HW); g.dz = carve((size_t)B * p->NE * HW);
        g.dkpart = carve((size_t)B * composite_bwd_tiles(H, W) * 256);
        g.dstate = carve((size_t)T * B * 5);
        g.lnpart = carve((size_t)B * ln_bwd_slices((int)(64 * HW)) * 2);
        g.ln_ppart_floats = 0;
        for (int j = 0; j < 9; ++j) {
            const size_t n = (size_t)ln_bwd_param_part_floats((int)lnsz[j]);
            g.ln_ppart[j] = carve(n);
            g.ln_ppart_floats = g.ln_ppart[j] + n - g.ln_ppart[0];
        }
        {   // slots 7..11 = enc6, enc5, enc4 (transposed, anchors = their INPUT maps), enc2, enc1 (stride-2 convs)
            const int mode[5] = {1, 1, 1, 0, 0}, ci[5] = {64, 96, 128, 64, 32};
            const int hin[5] = {p->H2, p->H4, p->H8, p->H4, p->H2}, win[5] = {p->W2, p->W4, p->W8, p->W4, p->W2};
            g.wg_part_floats = 0;
            for (int k = 0; k < 5; ++k) {
                const size_t n = (size_t)conv_backward_part_floats(mode[k], ci[k], ci[k], B, hin[k], win[k]);
                g.wg_part[k] = carve(n);
                g.wg_part_floats = g.wg_part[k] + n - g.wg_part[0];      // carve() aligns: measure the region, not the sum
            }
        }
    }
    p->ws_floats = (long long)off;
}

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
    const int H = cfg->height, W = cfg->width;
    p->H2 = H / 2; p->W2 = W / 2; p->H4 = H / 4; p->W4 = W / 4; p->H8 = H / 8; p->W8 = W / 8;
    p->NP = cfg->num_masks + 1;
    p->NE = cfg->model_type == PIVP_MODEL_DNA ? 25 : 3;
    p->K5 = 128 * p->H8 * p->W8;
    p->ws = nullptr; p->ws_floats = 0; p->last_steps = 0; p->last_sched = false;
    { const char* e = getenv("PIVP_SIDE_STREAM"); p->use_side = !(e && e[0] == '0'); }
    { const char* e = getenv("PIVP_FINISH_RIDER"); p->rider = !(e && e[0] == '0'); }
    { const char* e = getenv("PIVP_FUSE_ENC3"); p->fuse_enc3 = !(e && e[0] == '0'); }
    { const char* e = getenv("PIVP_LN_FOLD_TRAIN"); p->ln_fold_train = !(e && e[0] == '0'); }

    auto add = [&](const std::string& name, long long n) { p->params.push_back({name, n, nullptr, nullptr, grad_group_of(name)}); return (int)p->params.size() - 1; };
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

    plan_layout(p);
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
    plan->packs_valid = 0;
    return PIVP_OK;
}
extern "C" int pivp_param_group(const pivp_plan_t* plan, int idx) {
    if (!plan || idx < 0 || idx >= (int)plan->params.size()) return PIVP_ERR_BADARG;
    return plan->params[idx].group;
}
extern "C" int pivp_param_group_by_name(const char* name) { return name ? grad_group_of(name) : PIVP_ERR_BADARG; }
extern "C" int pivp_plan_set_grad_callback(pivp_plan_t* plan, pivp_grad_group_cb cb, void* user) {
    if (!plan) return PIVP_ERR_BADARG;
    plan->grad_cb = cb; plan->grad_cb_user = user;
    return PIVP_OK;
}
extern "C" int pivp_plan_set_grad(pivp_plan_t* plan, int idx, float* dptr) {
    if (!plan || idx < 0 || idx >= (int)plan->params.size() || !dptr) return PIVP_ERR_BADARG;
    plan->params[idx].grad = dptr;
    return PIVP_OK;
}
// Precision of the ConvLSTM gate convolutions (95 % of the FLOPs): PIVP_PRECISION_F32 (default), PIVP_PRECISION_BF16X3 = every fp32 operand of the
// FORWARD gate convolutions as two bf16 pieces and three MFMAs per product (fp32-grade results, everything else and the backward in fp32), or
// PIVP_PRECISION_BF16 = operands
// rounded to bf16, fp32 accumulation / gates / state (csrc/convlstm_bf16.hip), and in the backward sweep their data and weight gradients
// (csrc/convlstm_bf16.hip <NCH, false>, csrc/wgrad_bf16.hip).  Everything else stays fp32, as do the parameters, the gradients and Adam.  Refused when a layer's map does not fit the bf16 kernel's tiles (8-wide maps need an even batch).
extern "C" int pivp_plan_set_precision(pivp_plan_t* plan, int precision) {
    if (!plan || precision < PIVP_PRECISION_F32 || precision > PIVP_PRECISION_FP16X3) return PIVP_ERR_BADARG;
    if (precision == PIVP_PRECISION_BF16 || precision == PIVP_PRECISION_BF16X3) {   // (BF16X6: a layer its tile does not serve runs the fp32 kernel)
        const int hs[7] = {plan->H2, plan->H2, plan->H4, plan->H4, plan->H8, plan->H4, plan->H2};
        const int wsz[7] = {plan->W2, plan->W2, plan->W4, plan->W4, plan->W8, plan->W4, plan->W2};
        for (int i = 0; i < 7; ++i) {
            IgemmDesc d;
            memset(&d, 0, sizeof(d));
            d.ksize = 5; d.pad = 2; d.in_step = 1; d.C = kLstm[i].C; d.c0 = kLstm[i].cx; d.c1 = kLstm[i].C;
            d.ld0 = 4; d.ld1 = 4; d.B = plan->cfg.batch; d.Hin = hs[i]; d.Win = wsz[i];
            if (!convlstm_bf16_ok(d)) return PIVP_ERR_BADARG;
        }
    }
    pivp_plan& m = *plan;
    struct Modes { int lstm_bf16, lstm_planes, bwd_planes, bf16_all, precision; bool x3_wgrad, x6_wgrad; };
    const Modes old{m.lstm_bf16, m.lstm_planes, m.bwd_planes, m.bf16_all, m.precision, m.x3_wgrad, m.x6_wgrad};
    m.lstm_bf16 = precision != PIVP_PRECISION_F32;
    m.lstm_planes = precision == PIVP_PRECISION_BF16X3 ? 2 : precision == PIVP_PRECISION_BF16X6 ? 3 : precision == PIVP_PRECISION_FP16X3 ? -2 : 1;
    m.bwd_planes = m.lstm_planes;          // fp16 pieces in the sweep: the data gradients take dG times a power of two from its largest |value|
    m.x3_wgrad = m.lstm_planes == -2;      // ... and the weight gradients two fp16 pieces, two timesteps per launch
    m.x6_wgrad = m.lstm_planes == 3;       // three-piece mode: the weight gradients with three bf16 pieces, same schedule
    m.bf16_all = precision == PIVP_PRECISION_BF16;
    m.precision = precision;
    // The dG rings' depth follows the precision, so the order is set_precision -> workspace_bytes -> set_workspace (include/pivp_hip.h).  With a
    // workspace already bound the layout it was sized for stays; a mode that needs deeper rings than it has is refused, not run short.
    if (m.ws && m.has_grads && wg_cap_of(plan) > m.wg_cap) {
        m.lstm_bf16 = old.lstm_bf16; m.lstm_planes = old.lstm_planes; m.bwd_planes = old.bwd_planes; m.bf16_all = old.bf16_all;
        m.precision = old.precision; m.x3_wgrad = old.x3_wgrad; m.x6_wgrad = old.x6_wgrad;
        return PIVP_ERR_STATE;
    }
    m.packs_valid = 0;
    if (!plan->ws) plan_layout(plan);
    return PIVP_OK;
}
// Inference with constant weights: keep the bf16 / fp16 weight packs across rollouts (on = 1) instead of rebuilding them at the start of each.  The
// caller then owes pivp_plan_params_changed after EVERY modification of a parameter tensor (optimizer step, checkpoint load, host write); set_param,
// set_precision and set_workspace invalidate by themselves.  Default off.
extern "C" int pivp_plan_set_pack_cache(pivp_plan_t* plan, int on) {
    if (!plan) return PIVP_ERR_BADARG;
    plan->pack_cache = on ? 1 : 0; plan->packs_valid = 0;
    return PIVP_OK;
}
extern "C" int pivp_plan_params_changed(pivp_plan_t* plan) {
    if (!plan) return PIVP_ERR_BADARG;
    plan->packs_valid = 0;
    return PIVP_OK;
}
static int lstm_w_of(const pivp_plan_t* p, int i) { const int ws[7] = {p->W2, p->W2, p->W4, p->W4, p->W8, p->W4, p->W2}; return ws[i]; }
extern "C" int pivp_plan_get_precision(const pivp_plan_t* plan) { return plan ? plan->precision : PIVP_ERR_BADARG; }
extern "C" long long pivp_plan_workspace_bytes(const pivp_plan_t* plan) { return plan ? plan->ws_floats * 4 : PIVP_ERR_BADARG; }
extern "C" int pivp_plan_set_workspace(pivp_plan_t* plan, void* dptr, long long bytes) {
    if (!plan || !dptr || bytes < plan->ws_floats * 4 || ((uintptr_t)dptr & 255)) return PIVP_ERR_BADARG;
    plan->ws = (float*)dptr;
    plan->packs_valid = 0;
    return PIVP_OK;
}
extern "C" int pivp_reset_state(pivp_plan_t* plan, void* stream) {
    if (!plan || !plan->ws) return PIVP_ERR_STATE;
    const size_t n = (size_t)plan->cfg.batch * plan->H2 * plan->W2 * 32;
    if (hipMemsetAsync(plan->ws + plan->o_zero, 0, n * 4, (hipStream_t)stream) != hipSuccess) return PIVP_ERR_LAUNCH;
    return PIVP_OK;
}

#define RC(call) do { int rc_ = (call); if (rc_ != PIVP_OK) return rc_; } while (0)

// the plan's second stream (lowest priority: the critical path wins the CUs it asks for) and its fork / join events
static int ensure_side(pivp_plan* plan) {
    if (!plan->use_side || plan->side) return PIVP_OK;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    if (hipStreamCreateWithPriority(&plan->side, hipStreamNonBlocking, least) != hipSuccess) {
        plan->side = nullptr; plan->use_side = false; (void)hipGetLastError();      // single-stream sweeps from here on
        return PIVP_OK;
    }
    bool ok = true;
    for (int i = 0; i < pivp_plan::NSLOT && ok; ++i)
        for (int r = 0; r < 2 && ok; ++r)
            ok = hipEventCreateWithFlags(&plan->ev_ready[i][r], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&plan->ev_done[i][r], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 7 && ok; ++i)
        for (int r = 0; r < 2 && ok; ++r)
            ok = hipEventCreateWithFlags(&plan->ev_ring_done[i][r], hipEventDisableTiming) == hipSuccess;
    if (!ok) {      // never leave `side` set beside null events: back to "no side stream", and this plan runs its sweeps on one stream
        plan->destroy_side();
        plan->use_side = false;
        (void)hipGetLastError();
    }
    return PIVP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// forward, one timestep (TM:659-731)
// ------------------------------------------------------------------------------------------------------------
static int run_step(pivp_plan* p, int t, const float* prev, const float* action, const float* state_prev,
                    float* gen_out, float* state_out, hipStream_t s) {
    const pivp_config_t& c = p->cfg;
    const int B = c.batch, H = c.height, W = c.width;
    float* ws = p->ws;
    const Slab& S = p->slabs[t % p->nslabs];
    const Slab* Sp = t > 0 ? &p->slabs[(t - 1) % p->nslabs] : nullptr;
    float* lnp = ws + p->o_lnpart;
    const float eps = c.ln_eps;
    const bool train = c.keep_activations != 0;
    auto hp = [&](int i) -> const float* { return Sp ? ws + Sp->h[i] : nullptr; };   // t = 0: h == 0, skipped
    auto cp = [&](int i) { return Sp ? ws + Sp->c[i] : ws + p->o_zero; };
    const int ln_cap = ln_partial_cap(H * W);   // partial slots per sample in lnp (see the workspace carve)
    int np = 0;
    auto lstm = [&](int i, const float* x, int ldx, int hh, int wwid, const LnIn* ln_in = nullptr, float* part_out = nullptr) {
        const bool prof = p->prof_on && p->prof_used + 2 <= p->prof_ev.size();
        if (prof) (void)hipEventRecord(p->prof_ev[p->prof_used], s);
        // the LayerNorm behind every ConvLSTM gets its statistics from the ConvLSTM epilogue (np partials per sample)
        int rc = run_convlstm(x, kLstm[i].cx, ldx, hp(i), kLstm[i].C, P(p, p->i_lstm_w[i]), P(p, p->i_lstm_b[i]),
                              cp(i), ws + S.c[i], ws + S.h[i], B, hh, wwid, s, 0, train ? ws + S.gates[i] : nullptr,
                              part_out ? part_out : lnp, ln_cap, &np,
                              p->lstm_bf16 ? reinterpret_cast<const unsigned short*>(ws + p->o_wbf16[i]) : nullptr, p->lstm_planes, ln_in);
        if (prof) {
            (void)hipEventRecord(p->prof_ev[p->prof_used + 1], s);
            p->prof_layer[p->prof_used / 2] = i + (Sp ? 0 : 8);   // +8: first-step launch without the h half of K
            p->prof_used += 2;
        }
        return rc;
    };
    auto ln = [&](int j, const float* x, float* out, int n, int C, int ldo, int relu, int fused_nparts) {
        return run_layernorm(x, P(p, p->i_ln_g[j]), P(p, p->i_ln_b[j]), out, lnp, B, n, C, ldo, eps, relu, s,
                             ws + S.lnstat + (size_t)j * B * 2, fused_nparts);
    };
    const int n2 = 32 * p->H2 * p->W2, n4 = 64 * p->H4 * p->W4, n8 = 128 * p->H8 * p->W8;

    // group 0 (TM:595): enc0 -> norm_enc0 -> relu   => cat7[:, 32:64]
    RC(conv_enc0(prev, P(p, p->i_enc_w[0]), P(p, p->i_enc_b[0]), ws + S.e0raw, B, H, W, s, lnp, ln_cap, &np));
    RC(ln(0, ws + S.e0raw, ws + S.cat7 + 32, n2, 32, 64, 1, np));
    // group 1 (TM:596): lstm1 -> hidden1 -> lstm2 -> hidden2 -> enc1 -> relu  => cat6[:, 64:96]
    RC(lstm(0, ws + S.cat7 + 32, 64, p->H2, p->W2));
    // Inference rollouts in the split precision modes: hidden1 / hidden3 feed only lstm2 / lstm4, whose eight-wave kernels apply the norm while
    // they stage their patch (the partials of their own output go to the second buffer: their blocks finish while others still read the input's).
    float* const lnpA = ws + p->o_lnpart, * const lnpB = ws + p->o_lnpart2;
    auto other = [&](float* q) { return q == lnpA ? lnpB : lnpA; };
    const bool fold_l2 = !train && p->lstm_bf16 && np > 0 && convlstm_ln_in_ok(p->lstm_planes, 32, 32, 32, B, p->H2, p->W2) &&
                         (long)B * (p->H2 / 8) * (p->W2 / 16) >= 128;
    if (fold_l2) {
        const LnIn li{P(p, p->i_ln_g[1]), P(p, p->i_ln_b[1]), lnp, np, eps};
        float* q = other(lnp);
        RC(lstm(1, ws + S.h[0], 32, p->H2, p->W2, &li, q));
        lnp = q;
    } else {
        RC(ln(1, ws + S.h[0], ws + S.n1, n2, 32, 32, 0, np));
        RC(lstm(1, ws + S.n1, 32, p->H2, p->W2));
    }
    // Inference rollouts: hidden2 / hidden4 feed only enc1 / enc2, so their norms are applied while those convs stage their input
    // (run_conv3x3s2_ln) instead of by a launch of their own; training keeps the materialised tensors (the backward sweep reads them).
    // Training plans take the same launch (round 6) and have it WRITE the normalised hidden2 / hidden4 (every input pixel of a stride-2 3x3 conv is the
    // tap (1..2, 1..2) of exactly one anchor) and the samples' (mean, rstd) for the backward sweep: two ln_apply launches per timestep less.
    if (np > 0 && p->ln_fold_train && conv3x3s2_ln_ok(32, 32, B, p->H2, p->W2)) {
        RC(run_conv3x3s2_ln(ws + S.h[1], 32, P(p, p->i_enc_w[1]), P(p, p->i_enc_b[1]), ws + S.cat6 + 64, 32, 96, 1, B, p->H2, p->W2, s,
                            P(p, p->i_ln_g[2]), P(p, p->i_ln_b[2]), lnp, np, eps, nullptr,
                            train ? ws + S.n2 : nullptr, 32, train ? ws + S.lnstat + (size_t)2 * B * 2 : nullptr));
    } else if (!train && np > 0 && conv3x3s2_ln_ok(32, 32, B, p->H2, p->W2)) {
        RC(run_conv3x3s2_ln(ws + S.h[1], 32, P(p, p->i_enc_w[1]), P(p, p->i_enc_b[1]), ws + S.cat6 + 64, 32, 96, 1, B, p->H2, p->W2, s,
                            P(p, p->i_ln_g[2]), P(p, p->i_ln_b[2]), lnp, np, eps));
    } else {
        RC(ln(2, ws + S.h[1], ws + S.n2, n2, 32, 32, 0, np));
        RC(run_conv3x3s2(ws + S.n2, 32, 32, P(p, p->i_enc_w[1]), P(p, p->i_enc_b[1]), ws + S.cat6 + 64, 32, 96, 1, B, p->H2, p->W2, s));
    }
    // group 2 (TM:597)
    RC(lstm(2, ws + S.cat6 + 64, 96, p->H4, p->W4));
    const bool fold_l4 = !train && p->lstm_bf16 && np > 0 && convlstm_ln_in_ok(p->lstm_planes, 64, 64, 64, B, p->H4, p->W4) &&
                         (long)B * (p->H4 / 8) * (p->W4 / 16) >= 64;
    if (fold_l4) {
        const LnIn li{P(p, p->i_ln_g[3]), P(p, p->i_ln_b[3]), lnp, np, eps};
        float* q = other(lnp);
        RC(lstm(3, ws + S.h[2], 64, p->H4, p->W4, &li, q));
        lnp = q;
    } else {
        RC(ln(3, ws + S.h[2], ws + S.n3, n4, 64, 64, 0, np));
        RC(lstm(3, ws + S.n3, 64, p->H4, p->W4));
    }
    bool enc3_done = false;
    if (train && np > 0 && p->ln_fold_train && conv3x3s2_ln_ok(64, 64, B, p->H4, p->W4)) {
        RC(run_conv3x3s2_ln(ws + S.h[3], 64, P(p, p->i_enc_w[2]), P(p, p->i_enc_b[2]), ws + S.e2, 64, 64, 1, B, p->H4, p->W4, s,
                            P(p, p->i_ln_g[4]), P(p, p->i_ln_b[4]), lnp, np, eps, nullptr, ws + S.n4, 64, ws + S.lnstat + (size_t)4 * B * 2));
    } else
    if (!train && np > 0 && conv3x3s2_ln_ok(64, 64, B, p->H4, p->W4)) {
        // Inference: group 3 (TM:598) and the state predictor (TM:730) in enc2's epilogue (round 6): enc3_state_kernel's launch
        // disappears; bit-identical to it (the same fmaf chain on the matrix cores).  PIVP_FUSE_ENC3=0: two launches, as before.
        Enc3Fuse f3;
        memset(&f3, 0, sizeof(f3));
        if (p->fuse_enc3 && (p->H8 * p->W8) % 32 == 0) {
            f3.w3 = P(p, p->i_enc_w[3]); f3.b3 = P(p, p->i_enc_b[3]); f3.action = action; f3.state = state_prev; f3.wcs = P(p, p->i_cs_w); f3.bcs = P(p, p->i_cs_b);
            f3.e3 = ws + S.e3; f3.state_out = state_out; f3.use_state = c.use_state;
            enc3_done = true;
        }
        RC(run_conv3x3s2_ln(ws + S.h[3], 64, P(p, p->i_enc_w[2]), P(p, p->i_enc_b[2]), ws + S.e2, 64, 64, 1, B, p->H4, p->W4, s,
                            P(p, p->i_ln_g[4]), P(p, p->i_ln_b[4]), lnp, np, eps, &f3));
    } else {
        RC(ln(4, ws + S.h[3], ws + S.n4, n4, 64, 64, 0, np));
        RC(run_conv3x3s2(ws + S.n4, 64, 64, P(p, p->i_enc_w[2]), P(p, p->i_enc_b[2]), ws + S.e2, 64, 64, 1, B, p->H4, p->W4, s));
    }
    // group 3 (TM:598) + state predictor (TM:730)
    if (!enc3_done)
    RC(enc3_state(ws + S.e2, action, state_prev, P(p, p->i_enc_w[3]), P(p, p->i_enc_b[3]), P(p, p->i_cs_w), P(p, p->i_cs_b),
                  ws + S.e3, state_out, B, p->H8 * p->W8, c.use_state, s));
    // group 4 (TM:599)
    RC(lstm(4, ws + S.e3, 64, p->H8, p->W8));
    // (hidden5's norm was folded the same way into enc4's four-phase form and the motion head's Linear, which then has to run here, in front
    // of lstm6: the two consumers lost more than the launch saves, rollout 8.52 -> 8.60 ms.  It keeps its own launch.)
    RC(ln(5, ws + S.h[4], ws + S.n5, n8, 128, 128, 0, np));
    // enc4 and the motion head's Linear both read hidden5 and nothing else: when the output side runs as frame_head (which finishes the
    // Linear's partial sums itself), the two share ONE grid here -- 1,024 + 256 blocks that each filled a fraction of the chip
    const bool fh = frame_head_pays(c.model_type, B, H, W, c.num_masks);
    const bool fh_fin = fh && c.model_type != PIVP_MODEL_DNA && frame_head_finishes(p->K5);
    bool partials_done = false;
    if (fh_fin && !p->bf16_all && p->lstm_planes != 2) {
        RC(run_deconv3x3s2_and_partials(ws + S.n5, 128, P(p, p->i_enc_w[4]), P(p, p->i_enc_b[4]), ws + S.e4, 128, 128, 1, B, p->H8, p->W8, s,
                                        P(p, p->i_head_w), ws + p->o_linpart, c.model_type == PIVP_MODEL_STP ? 1 : 0));
        partials_done = true;
    } else
    RC(run_deconv3x3s2(ws + S.n5, 128, 128, P(p, p->i_enc_w[4]), P(p, p->i_enc_b[4]), ws + S.e4, 128, 128, 1, B, p->H8, p->W8, s, 0,
                       nullptr, 0, nullptr, p->bf16_all ? 1 : p->lstm_planes == 2 ? 2 : 0));
    // group 5 (TM:600): lstm6 -> hidden6 -> concat(., enc1) -> enc5 -> relu
    RC(lstm(5, ws + S.e4, 128, p->H4, p->W4));
    const int dprec = p->bf16_all ? 1 : p->lstm_planes == 2 ? 2 : p->lstm_planes == -2 ? 3 : 0;      // (-2: two fp16 pieces, fp32-grade)
    // The norms of hidden6 / hidden7 feed only enc5 / enc6, whose tile kernel applies them while it stages its patch ([hidden6 | enc1],
    // [hidden7 | enc0] as two sources).
    // Training plans take the same launch and have it WRITE the normalised hidden6 / hidden7 (each pixel by the block that owns it) and the
    // samples' (mean, rstd) for the backward sweep: no ln_apply launch there either.
    // Only while the norm is a launch-bound 5-us kernel (<= 6 MB of hidden tensor: 64 x 64 frames at B = 32): the consumer's column blocks
    // each stage the patch and its gamma / beta again, which at config 5's 128 x 128 costs more than one bandwidth-bound ln_apply pass
    // (B = 32, T = 20: rollout 65.5 -> 65.9 ms with the fold, so it is not taken there).
    auto fold_small = [&](long long floats) { return floats * 4 <= 6LL << 20; };
    // The motion head's finisher (sum of the Linear's K-slice partials, bias, activation, normalisation: per-SAMPLE work) rides behind enc5's tiles as
    // B extra blocks (192 tiles on 256 CUs at B = 32) instead of being repeated by each of frame_head's 16 bands per sample (round 6).
    bool finished = false;
    if (np > 0 && fold_small((long long)B * n4) && deconv3x3s2_ln_ok(64, 32, 96, B, p->H4, p->W4)) {
        MotionRider rd;
        memset(&rd, 0, sizeof(rd));
        if (fh_fin && p->rider) {
            if (!partials_done) { RC(motion_partials(ws + S.n5, P(p, p->i_head_w), ws + p->o_linpart, B, p->K5, c.model_type == PIVP_MODEL_STP ? 1 : 0, s)); partials_done = true; }
            rd.mode = c.model_type == PIVP_MODEL_STP ? 2 : 1; rd.KS = cdna_kernel_partials_slices(p->K5); rd.nout = 25 * c.num_masks;
            rd.partials = ws + p->o_linpart; rd.bias = P(p, p->i_head_b); rd.vpre = ws + S.vpre;
            if (rd.mode == 2) { rd.w2 = P(p, p->i_head2_w); rd.b2 = P(p, p->i_head2_b); rd.out = ws + S.theta; }
            else rd.out = ws + S.kerns;
            finished = true;
        }
        RC(run_deconv3x3s2_ln(ws + S.h[5], 64, ws + S.cat6 + 64, 32, 96, P(p, p->i_enc_w[5]), P(p, p->i_enc_b[5]), ws + S.e5, 96, 96, 1, B, p->H4, p->W4,
                              s, P(p, p->i_ln_g[6]), P(p, p->i_ln_b[6]), lnp, np, eps, nullptr, 0, nullptr, dprec,
                              train ? ws + S.cat6 : nullptr, 96, train ? ws + S.lnstat + (size_t)6 * B * 2 : nullptr, ws + p->o_wabs[0], &rd));
    } else {
        RC(ln(6, ws + S.h[5], ws + S.cat6, n4, 64, 96, 0, np));
        RC(run_deconv3x3s2(ws + S.cat6, 96, 96, P(p, p->i_enc_w[5]), P(p, p->i_enc_b[5]), ws + S.e5, 96, 96, 1, B, p->H4, p->W4, s, 0,
                           nullptr, 0, nullptr, dprec, ws + p->o_wabs[0]));
    }
    // group 6 (TM:601): lstm7 -> hidden7 -> concat(., enc0) -> enc6 -> norm_enc6 -> relu
    RC(lstm(6, ws + S.e5, 96, p->H2, p->W2));
    if (np > 0 && fold_small((long long)B * n2) && deconv3x3s2_ln_ok(32, 32, 64, B, p->H2, p->W2)) {
        // enc6's blocks write the partials of norm_enc6 while others still read hidden7's: the second partial buffer
        float* lnp2 = other(lnp);
        const int np_in = np;
        RC(run_deconv3x3s2_ln(ws + S.h[6], 32, ws + S.cat7 + 32, 32, 64, P(p, p->i_enc_w[6]), P(p, p->i_enc_b[6]), ws + S.e6raw, 64, 64, 0, B, p->H2, p->W2,
                              s, P(p, p->i_ln_g[7]), P(p, p->i_ln_b[7]), lnp, np_in, eps, lnp2, ln_cap, &np, dprec,
                              train ? ws + S.cat7 : nullptr, 64, train ? ws + S.lnstat + (size_t)7 * B * 2 : nullptr, ws + p->o_wabs[1]));
        lnp = lnp2;
    } else {
        RC(ln(7, ws + S.h[6], ws + S.cat7, n2, 32, 64, 0, np));
        RC(run_deconv3x3s2(ws + S.cat7, 64, 64, P(p, p->i_enc_w[6]), P(p, p->i_enc_b[6]), ws + S.e6raw, 64, 64, 0, B, p->H2, p->W2, s, 0,
                           lnp, ln_cap, &np, dprec, ws + p->o_wabs[1]));
    }
    // heads (TM:711-728).  One launch (csrc/frame_head.hip) for norm_enc6 + relu + the 1x1 heads + the motion head's finisher + flat softmax +
    // transform + compositing, behind the Linear's partial sums: bit-identical to the four launches below it, which remain for geometries
    // it does not take or where it is slower (frames wider than 64: frame_head_pays).  The softmaxed masks are kept for the rollout's last step only (pivp_get_tap).
    if (fh && np > 0) {
        const bool fin = !finished && (c.model_type == PIVP_MODEL_DNA || frame_head_finishes(p->K5));
        FrameHeadArgs a;
        memset(&a, 0, sizeof(a));
        if (finished) {
            a.aux = c.model_type == PIVP_MODEL_STP ? ws + S.theta : ws + S.kerns;
        } else if (c.model_type == PIVP_MODEL_CDNA) {
            if (fin) { if (!partials_done) RC(motion_partials(ws + S.n5, P(p, p->i_head_w), ws + p->o_linpart, B, p->K5, 0, s)); }
            else RC(cdna_kernels(ws + S.n5, P(p, p->i_head_w), P(p, p->i_head_b), ws + p->o_linpart, ws + S.kerns, B, p->K5, c.num_masks, s, ws + S.vpre));
            a.aux = ws + S.kerns; a.kerns_out = ws + S.kerns;
        } else if (c.model_type == PIVP_MODEL_STP) {
            if (fin) { if (!partials_done) RC(motion_partials(ws + S.n5, P(p, p->i_head_w), ws + p->o_linpart, B, p->K5, 1, s)); }
            else RC(stp_params(ws + S.n5, P(p, p->i_head_w), P(p, p->i_head_b), P(p, p->i_head2_w), P(p, p->i_head2_b), ws + p->o_linpart,
                               ws + S.theta, B, p->K5, s, ws + S.vpre));
            a.aux = ws + S.theta; a.kerns_out = ws + S.theta; a.w2 = P(p, p->i_head2_w); a.b2 = P(p, p->i_head2_b);
        }
        if (fin && c.model_type != PIVP_MODEL_DNA) {
            a.partials = ws + p->o_linpart; a.KS = cdna_kernel_partials_slices(p->K5); a.hbias = P(p, p->i_head_b); a.aux = nullptr;
            a.vpre_out = ws + S.vpre;
        } else {
            a.kerns_out = nullptr;      // already written by the separate finisher
        }
        a.e6raw = ws + S.e6raw; a.ln_part = lnp; a.ln_nparts = np; a.gamma = P(p, p->i_ln_g[8]); a.beta = P(p, p->i_ln_b[8]); a.eps = eps;
        a.wm = P(p, p->i_masks_w); a.bm = P(p, p->i_masks_b); a.we = P(p, p->i_enc7_w); a.be = P(p, p->i_enc7_b);
        a.prev = prev; a.out = gen_out; a.enc7 = ws + S.enc7;
        a.masks_out = t == c.seq_len - 2 ? ws + p->o_masks : nullptr;
        a.stat_out = ws + S.lnstat + (size_t)8 * B * 2;
        if (train) { a.logits_out = ws + S.logits; a.layer0_out = ws + S.layer0; a.y_out = ws + S.e6; }
        a.B = B; a.H = H; a.W = W; a.NM = c.num_masks; a.stp_zero = c.stp_zero_border;
        return frame_head(a, c.model_type, s);
    }
    if (np > 0 && (H * W) % 64 == 0) {
        RC(heads_1x1(ws + S.e6raw, P(p, p->i_masks_w), P(p, p->i_masks_b), P(p, p->i_enc7_w), P(p, p->i_enc7_b),
                     ws + S.logits, ws + S.enc7, ws + S.layer0, B, H * W, p->NP, p->NE, c.model_type, s,
                     lnp, np, P(p, p->i_ln_g[8]), P(p, p->i_ln_b[8]), eps, train ? ws + S.e6 : nullptr,
                     ws + S.lnstat + (size_t)8 * B * 2));
    } else {
        RC(ln(8, ws + S.e6raw, ws + S.e6, 64 * H * W, 64, 64, 1, np));
        RC(heads_1x1(ws + S.e6, P(p, p->i_masks_w), P(p, p->i_masks_b), P(p, p->i_enc7_w), P(p, p->i_enc7_b),
                     ws + S.logits, ws + S.enc7, ws + S.layer0, B, H * W, p->NP, p->NE, c.model_type, s));
    }
    // (The motion head's generator needs hidden5 only and could run on the side stream under lstm6 .. heads: measured, the rollout got
    // SLOWER, 8.68 -> 8.79 ms: the co-running kernels cost the ConvLSTMs more than the 16 us they hide; again with the ConvLSTM waves at
    // priority 3 and the generator at 0: 8.57 -> 8.67 / 8.73 ms; and forked only behind norm(hidden7), i.e. beside enc6 and the heads kernel,
    // no ConvLSTM: 8.60 -> 8.68 / 8.69 ms -- a cross-stream event pair per timestep costs more than the 16 us.  It stays in line.)
    const float* aux = nullptr;
    if (c.model_type == PIVP_MODEL_CDNA) {
        RC(cdna_kernels(ws + S.n5, P(p, p->i_head_w), P(p, p->i_head_b), ws + p->o_linpart, ws + S.kerns, B, p->K5, c.num_masks, s,
                        ws + S.vpre));
        aux = ws + S.kerns;
    } else if (c.model_type == PIVP_MODEL_STP) {
        RC(stp_params(ws + S.n5, P(p, p->i_head_w), P(p, p->i_head_b), P(p, p->i_head2_w), P(p, p->i_head2_b),
                      ws + p->o_linpart, ws + S.theta, B, p->K5, s, ws + S.vpre));
        aux = ws + S.theta;
    } else {
        aux = ws + S.enc7;
    }
    RC(composite(prev, ws + S.logits, ws + S.layer0, aux, gen_out, ws + p->o_masks, B, H, W, c.num_masks,
                 c.model_type, c.stp_zero_border, s));
    return PIVP_OK;
}

// the frame fed to step t (TM:663-673)
static const float* step_input(pivp_plan* plan, int t, const float* images, const unsigned char* gt_select, const float* gen_images, size_t fr) {
    const int ctx = plan->cfg.context_frames;
    if (t < ctx) return images + t * fr;
    if (!gt_select) return gen_images + (t - 1) * fr;
    return plan->ws + plan->slabs[t % plan->nslabs].prevsel;
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
    // The precision modes' weight packs: rebuilt at the start of every rollout (the parameters may have changed since the last call: optimizer step,
    // checkpoint load) unless the caller keeps them -- pivp_plan_set_pack_cache(plan, 1) -- and reports every change with pivp_plan_params_changed.
    const bool repack = !(plan->pack_cache && plan->packs_valid);
    if (repack && plan->lstm_planes == -2) {  // two fp16 pieces: the enc5 / enc6 weights' scales (their tile kernel splits the fp32 weights while it stages them)
        RC(absmax_partials(P(plan, plan->i_enc_w[5]), 9L * 96 * 96, plan->ws + plan->o_wabs[0], s));
        RC(absmax_partials(P(plan, plan->i_enc_w[6]), 9L * 64 * 64, plan->ws + plan->o_wabs[1], s));
    }
    if (repack && plan->lstm_bf16 && plan->lstm_planes == 1) {      // one bf16 plane: the seven packs in one launch
        WeightPrepJob jobs[7];
        for (int i = 0; i < 7; ++i)
            jobs[i] = WeightPrepJob{1, P(plan, plan->i_lstm_w[i]), plan->ws + plan->o_wbf16[i], kLstm[i].cx + kLstm[i].C, 4 * kLstm[i].C, 4 * kLstm[i].C, 0};
        RC(weight_prep_batch(jobs, 7, s));
    } else if (repack && plan->lstm_bf16)
        for (int i = 0; i < 7; ++i)
            RC(pack_lstm_bf16(P(plan, plan->i_lstm_w[i]), reinterpret_cast<unsigned short*>(plan->ws + plan->o_wbf16[i]),
                              kLstm[i].cx + kLstm[i].C, 4 * kLstm[i].C, s, 0, plan->lstm_planes,
                              0));
    plan->packs_valid = 1;
    for (int t = 0; t < T - 1; ++t) {
        if (t >= ctx && gt_select)                                     // TM:667-670
            RC(run_select_frames(images + t * fr, gen_images + (t - 1) * fr, gt_select + (size_t)t * B,
                                 plan->ws + plan->slabs[t % plan->nslabs].prevsel, B, (int)(fr / B), s));
        const float* prev = step_input(plan, t, images, gt_select, gen_images, fr);
        const float* st_prev = t == 0 ? states : gen_states + (size_t)(t - 1) * B * 5;   // TM:646, TM:730
        RC(run_step(plan, t, prev, actions + (size_t)t * B * 5, st_prev, gen_images + t * fr, gen_states + (size_t)t * B * 5, s));
    }
    plan->last_steps = T - 1;
    plan->last_sched = gt_select != nullptr;
    // loss (TM:737-759): frames ctx..T-1 vs gen[ctx-1..T-2]
    const int nf = T - ctx;
    float* lp = plan->ws + plan->o_losspart;
    // target frames and predictions are contiguous over time: when a frame is a whole number of partial chunks, ONE launch over
    // all nf frames leaves the partials in the same per-frame layout as nf launches would
    const bool whole_chunks = loss_partials_count((int)fr + 1) == loss_partials_count((int)fr) + 1;   // fr divisible by the chunk
    if (nf > 0 && (whole_chunks || nf == 1) && fr * nf < (1u << 30)) {
        RC(frame_sqerr_partials(images + (size_t)ctx * fr, gen_images + (size_t)(ctx - 1) * fr, lp, (int)(fr * nf), s));
    } else {
        for (int i = 0; i < nf; ++i)
            RC(frame_sqerr_partials(images + (size_t)(ctx + i) * fr, gen_images + (size_t)(ctx - 1 + i) * fr,
                                    lp + (size_t)i * plan->loss_nparts, (int)fr, s));
    }
    RC(loss_finalize(lp, plan->loss_nparts, nf, (int)fr, states + (size_t)ctx * B * 5, gen_states + (size_t)(ctx - 1) * B * 5,
                     B * 5, (float)(T - ctx), results, s));
    return PIVP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// backward through time (what loss.backward() does inside Chainer's optimizer.update, TM:950), all three heads.
// Gradients are ACCUMULATED into the buffers registered with pivp_plan_set_grad (same layouts as the parameters).
// ------------------------------------------------------------------------------------------------------------
// side-stream slots whose weight gradients belong to gradient group g (pivp_plan::NSLOT numbering)
static const int kGroupSlots[6][4] = {{7, 13, -1, -1}, {6, -1, -1, -1}, {8, 5, -1, -1}, {9, 4, -1, -1}, {10, 3, 2, -1}, {11, 1, 0, 12}};
// make `stream` wait for the side stream's latest work of one slot (no side stream: nothing to do).  NB: a null hipStream_t is the
// legacy default stream, a perfectly good stream to wait on -- never a "no stream" marker.
static int wait_slot(pivp_plan* p, int sl, hipStream_t stream) {
    if (!p->side) return PIVP_OK;
    for (int r = 0; r < 2; ++r)
        if (hipStreamWaitEvent(stream, sl >= 7 ? p->ev_done[sl][r] : p->ev_ring_done[sl][r], 0) != hipSuccess) return PIVP_ERR_LAUNCH;
    return PIVP_OK;
}
// Wave priority of the main stream's kernels during this plan's calls (csrc/pivp_common.h): the device word is rewritten only when the wanted value
// differs from what this process last wrote on the device.
// The cache of what the device words hold is process-wide (the words are per device, not per plan): guarded by a mutex, marked unknown while the
// seven copies are being enqueued and valid only once all of them were accepted.  Plans with DIFFERENT wishes driven concurrently on one device (other
// host threads / streams) may each run with either value: the copies are stream-ordered on the enqueuing plan's stream only.  Results never depend on it.
static int apply_main_prio(pivp_plan* p, hipStream_t s) {
    static std::mutex mu;
    static int current[PIVP_MAX_DEV] = {};      // device words start at 0; -1 = unknown (a setter failed midway)
    const int want = p->main_prio < 0 ? (p->grad_cb ? 0 : 1) : (p->main_prio ? 1 : 0);
    const int dev = pivp_current_device();
    std::lock_guard<std::mutex> lock(mu);
    if (current[dev] == want) return PIVP_OK;
    current[dev] = -1;
    RC(main_prio_set_backward(want, s)); RC(main_prio_set_backward_heads(want, s)); RC(main_prio_set_convlstm_bf16(want, s)); RC(main_prio_set_conv5x5_bf16(want, s));
    RC(main_prio_set_deconv_tile(want, s)); RC(main_prio_set_igemm_f32(want, s)); RC(main_prio_set_igemm_small(want, s));
    current[dev] = want;
    return PIVP_OK;
}
extern "C" int pivp_plan_set_main_priority(pivp_plan_t* plan, int mode) {
    if (!plan || mode < -1 || mode > 1) return PIVP_ERR_BADARG;
    plan->main_prio = mode;
    return PIVP_OK;
}
extern "C" int pivp_plan_set_group_join(pivp_plan_t* plan, int join) {
    if (!plan) return PIVP_ERR_BADARG;
    plan->group_join = join != 0;
    return PIVP_OK;
}
extern "C" int pivp_plan_group_wait(pivp_plan_t* plan, int group, void* stream) {
    if (!plan || group < 0 || group >= 6) return PIVP_ERR_BADARG;
    for (int k = 0; k < 4; ++k) if (kGroupSlots[group][k] >= 0) RC(wait_slot(plan, kGroupSlots[group][k], (hipStream_t)stream));
    return PIVP_OK;
}

// wg_ring / wg_slot: where this step's gate gradients go in the ConvLSTMs' dG rings; wg_flush: this step closes its batch.
// eg_ring / eg_slot / eg_flush: the same for the dY rings of the five stride-2 3x3 layers (Grads::cat6); eq_next: the ring slot step t + 1 used.
static int backward_step(pivp_plan* p, int t, const float* prev, bool prev_has_grad, const float* action, const float* state_prev,
                         bool has_go, float* go, float* go_prev, bool last_step, int wg_ring, int wg_slot, bool wg_flush,
                         int eg_ring, int eg_slot, bool eg_flush, int eq_next, hipStream_t s) {
    const pivp_config_t& c = p->cfg;
    const int B = c.batch, H = c.height, W = c.width, HW = H * W;
    float* ws = p->ws;
    const Slab& S = p->slabs[t];
    const Slab* Sp = t > 0 ? &p->slabs[t - 1] : nullptr;
    const Grads& g = p->g;
    const int par = t & 1, npar = par ^ 1;
    const int eq = eg_ring * p->eg_cap + eg_slot;                              // this step's slot of the enc dY rings
    float* const d_cat6 = ws + g.cat6 + (size_t)eq * g.cat6_sz;
    float* const d_e2 = ws + g.e2 + (size_t)eq * g.e2_sz;
    float* const d_e6raw = ws + g.e6raw + (size_t)eq * g.e6raw_sz;
    // a cell's input gradient: this step's buffer (cur) or the one step t + 1 wrote (its h columns are this step's d h)
    auto DIN = [&](int i, bool cur) -> float* { return ws + g.din[i] + (size_t)(i >= 5 ? (cur ? eq : eq_next) : (cur ? par : npar)) * g.din_sz[i]; };
    const int n2 = 32 * p->H2 * p->W2, n4 = 64 * p->H4 * p->W4, n8 = 128 * p->H8 * p->W8;
    float* lnpart = ws + g.lnpart;
    const long long ln_n[9] = {n2, n2, n2, n4, n4, n8, n4, n2, 64LL * HW};   // elements per sample of norm_enc0, hidden1..7, norm_enc6
    auto ln_finish = [&](int j) -> int {
        if (!p->ln_touched[j]) return PIVP_OK;
        p->ln_touched[j] = false;
        return ln_bwd_params_reduce(ws + g.ln_ppart[j], G(p, p->i_ln_g[j]), G(p, p->i_ln_b[j]), B, (int)ln_n[j], s);
    };
    auto lnb = [&](int j, const float* dy, int lddy, const float* y, int ldy, const float* x, float* dx, int n, int C, int relu) -> int {
        RC(ln_backward(dy, lddy, y, ldy, x, ws + S.lnstat + (size_t)j * B * 2, P(p, p->i_ln_g[j]), lnpart, dx,
                       G(p, p->i_ln_g[j]), G(p, p->i_ln_b[j]), B, n, C, relu, s, ws + g.ln_ppart[j]));
        p->ln_touched[j] = true;
        if (t == 0) RC(ln_finish(j));       // the sweep's last timestep: the partial planes become the gradient
        return PIVP_OK;
    };
    // weight-gradient slots (pivp_plan::NSLOT): join = the main stream waits for the slot's last weight-gradient kernels
    auto fork_of = [&](int slot, SideFork& f) -> const SideFork* {
        if (!p->side) return nullptr;
        const int r = slot >= 7 && slot < 12 ? eg_ring : par;
        f.side = p->side_of(slot); f.ready = p->ev_ready[slot][r]; f.done = p->ev_done[slot][r];
        return &f;
    };
    auto join = [&](int slot) -> int {
        if (!p->side) return PIVP_OK;
        if (slot >= 7 && slot < 12) {      // an enc layer's dY ring: only a batch's first step meets a slot the ring's previous launch (two batches ago) read
            if (eg_slot) return PIVP_OK;
            return hipStreamWaitEvent(s, p->ev_done[slot][eg_ring], 0) == hipSuccess ? PIVP_OK : PIVP_ERR_LAUNCH;
        }
        return hipStreamWaitEvent(s, p->ev_done[slot][par], 0) == hipSuccess ? PIVP_OK : PIVP_ERR_LAUNCH;   // this parity's last use: two timesteps ago (never recorded: returns at once)
    };
    const long long slab_bytes = p->nslabs > 1 ? ((long long)p->slabs[1].cat7 - (long long)p->slabs[0].cat7) * 4 : 0;
    // LayerNorm behind ConvLSTM i (hidden<i+1>): one launch leaves the two sums per sample and the norm's partial parameter planes
    // (ln_bwd_sums_params_kernel); the norm's dx is formed inside the cell's gate backward (lstm_gates_bwd_kernel with LnFuse) and never written.
    // (Round 4 also built the sums from the data gradients' epilogues and the parameter gradients inside the gate kernel -- 63 launches fewer per
    // step, and the step got SLOWER: fp32 28.21 -> 28.48 ms, bf16 11.85 -> 11.89, profiles/r04/NOTES.md 2.  Removed in round 5; in the history.)
    LnFuse lf[7];
    auto lnb_cell = [&](int i, const float* dy, int lddy, int n, int C) -> int {
        const int j = i + 1;
        memset(&lf[i], 0, sizeof(lf[i]));
        lf[i].dy = dy; lf[i].lddy = lddy; lf[i].gamma = P(p, p->i_ln_g[j]); lf[i].stat = ws + S.lnstat + (size_t)j * B * 2;
        lf[i].partials = lnpart; lf[i].S = ln_bwd_slices(n); lf[i].h = ws + S.h[i];
        p->ln_touched[j] = true;
        return ln_backward(dy, lddy, nullptr, 0, ws + S.h[i], lf[i].stat, lf[i].gamma, lnpart, nullptr,
                           G(p, p->i_ln_g[j]), G(p, p->i_ln_b[j]), B, n, C, 0, s, ws + g.ln_ppart[j]);
    };
    auto lstmb = [&](int i, const float* x, int ldx, int hh, int wwid, const EpSpec* ep = nullptr) -> int {
        const LstmSpec& L = kLstm[i];
        const int cin = L.cx + L.C, N = 4 * L.C;
        const size_t dG1 = (size_t)B * hh * wwid * N;                      // floats of one timestep's dG
        float* ring = ws + g.dG[i] + (size_t)wg_ring * p->wg_cap * dG1;
        const float* h_prev = Sp ? ws + Sp->h[i] : nullptr;
        SideFork f;
        if (wg_slot == 0) {     // a new batch: the ring's previous weight-gradient launch (two batches ago) must have read it
            if (p->side && hipStreamWaitEvent(s, p->ev_ring_done[i][wg_ring], 0) != hipSuccess) return PIVP_ERR_LAUNCH;
            p->wg_x[i] = x; p->wg_h[i] = h_prev;
        }
        RC(run_convlstm_backward(x, L.cx, ldx, h_prev, L.C, P(p, p->i_lstm_w[i]), ws + S.gates[i],
                                 Sp ? ws + Sp->c[i] : ws + p->o_zero, ws + S.c[i], nullptr, L.C,
                                 last_step ? nullptr : DIN(i, false) + L.cx, cin, ws + g.dc[i], last_step ? 0 : 1,
                                 ring + (size_t)wg_slot * dG1, ws + g.wt_lstm[i], DIN(i, true), nullptr, nullptr, B, hh, wwid,
                                 s, 1, (p->lstm_bf16 && ((p->bwd_planes != 3 && p->bwd_planes != -2) || wwid % 16 == 0 || B % 2 == 0)) ? reinterpret_cast<unsigned short*>(ws + g.wtb_lstm[i]) : nullptr, p->bwd_planes,
                                 wg_flush ? fork_of(i, f) : nullptr, &lf[i],    // dW = null: only the fork's `ready` (behind the gate math) is used
                                 t == 0 ? 1 : 0,
                                 (p->bwd_planes == -2 || p->x3_wgrad) ? ws + g.dg_absmax + ((size_t)(i * 2 + wg_ring) * p->wg_cap + wg_slot) * 72 : nullptr,    // t = 0: nobody reads d h_{-1}
                                 ep));
        if (t == 0) RC(ln_finish(i + 1));       // the sweep's last timestep: the norm's partial parameter planes (written by the gate kernel) become its gradient
        if (!wg_flush) return PIVP_OK;
        // weight + bias gradient of the whole batch: timestep j of it reads slab (first - j) and ring slot j; on the side stream it
        // starts as soon as this step's dG exists, next to this step's own data gradient
        hipStream_t sw = p->side ? p->side_of(i) : s;
        const int cnt = wg_slot + 1;
        int bias_done = 0;
        RC(run_wgrad(0, p->wg_x[i], L.cx, ldx, p->wg_h[i], L.C, L.C, cin, ring, N, N, G(p, p->i_lstm_w[i]), B, hh, wwid, hh, wwid, 5, 2, 1, sw,
                     G(p, p->i_lstm_b[i]), &bias_done, p->bf16_all ? 1 : (p->x6_wgrad && (wwid % 16 == 0 || B % 2 == 0)) ? 3 : 0, cnt, -slab_bytes, -slab_bytes,
                     (long long)dG1 * 4, nullptr, nullptr,
                     // (fp16 pieces; an 8-wide map with an odd batch does not fit that kernel's two-image tiles: the fp32 kernel takes the batch)
                     (p->x3_wgrad && (wwid % 16 == 0 || B % 2 == 0)) ? ws + g.dg_absmax + (size_t)(i * 2 + wg_ring) * p->wg_cap * 72 : nullptr, 72,
                     (long)B * c.height * c.width > 32L * 64 * 64 ? 2 : 0));      // (frames above 64 x 64 x 32: the eight-wave weight gradient on every layer, WgradDesc::form)
        if (!bias_done)
            for (int j = 0; j < cnt; ++j) RC(bias_grad(ring + (size_t)j * dG1, N, N, B * hh * wwid, G(p, p->i_lstm_b[i]), sw));
        if (p->side && hipEventRecord(p->ev_ring_done[i][wg_ring], p->side_of(i)) != hipSuccess) return PIVP_ERR_LAUNCH;
        return PIVP_OK;
    };
    const long px2 = (long)B * p->H2 * p->W2, px8 = (long)B * p->H8 * p->W8;

    // The side stream's weight gradients read dY buffers that a later step rewrites (enc0 / the motion head: the buffer of two timesteps ago; the stride-2
    // 3x3 layers: a ring slot of two batches ago): each is joined right in front of the first kernel that rewrites its buffer, not here -- at the top of a
    // timestep the side stream still has the previous timestep's last launches (lstm1's and enc0's weight gradients) in front of it, and the main stream
    // sat idle for ~40 us per timestep.
    SideFork fe;
    const int encp = p->bf16_all ? 1 : 0;      // bf16 mode: enc1's data gradient (the transposed conv's tile kernel) on bf16 operands too
    // Weight gradients of the stride-2 3x3 layers (k = 0..4: enc6, enc5, enc4, enc2, enc1): a step adds itself to the layer's open batch -- its dY sits in slot
    // eg_slot of the ring, its forward input one slab below the previous step's -- and the step that closes the batch launches ONE weight gradient over all of
    // them on the side stream (wgrad3x3s2.hip: per launch a block pays 37-150 KB of partial planes, per timestep nothing).  run_conv_backward (dW = null) has
    // recorded the fork's `ready` behind the final dY and in front of the data gradient.
    struct EncSpec { int mode, layer, c, ldx, ldy, Hin, Win; };
    const EncSpec kEnc[5] = {{1, 6, 64, 64, 64, p->H2, p->W2}, {1, 5, 96, 96, 128, p->H4, p->W4}, {1, 4, 128, 128, 192, p->H8, p->W8},
                             {0, 2, 64, 64, 64, p->H4, p->W4}, {0, 1, 32, 32, 96, p->H2, p->W2}};
    auto enc_add = [&](int k, const float* x) { if (!p->enc_cnt[k]++) p->enc_x0[k] = x; };
    auto enc_flush = [&](int k, bool forked) -> int {
        const int cnt = p->enc_cnt[k];
        if (!cnt) return PIVP_OK;
        const EncSpec& E = kEnc[k];
        const size_t ring0 = (size_t)eg_ring * p->eg_cap;
        const float* dy0 = k == 0 ? ws + g.e6raw + ring0 * g.e6raw_sz : k == 1 ? ws + g.din[6] + ring0 * g.din_sz[6] : k == 2 ? ws + g.din[5] + ring0 * g.din_sz[5]
                         : k == 3 ? ws + g.e2 + ring0 * g.e2_sz : ws + g.cat6 + ring0 * g.cat6_sz + 64;
        const long long dy_step = 4LL * (long long)(k == 0 ? g.e6raw_sz : k == 1 ? g.din_sz[6] : k == 2 ? g.din_sz[5] : k == 3 ? g.e2_sz : g.cat6_sz);
        hipStream_t sw = s;
        if (p->side) {
            if (!forked && (hipEventRecord(p->ev_ready[7 + k][eg_ring], s) != hipSuccess ||
                            hipStreamWaitEvent(p->side, p->ev_ready[7 + k][eg_ring], 0) != hipSuccess)) return PIVP_ERR_LAUNCH;
            sw = p->side_of(7 + k);
        }
        const int Hout = E.mode ? 2 * E.Hin : E.Hin / 2, Wout = E.mode ? 2 * E.Win : E.Win / 2;
        int bias_done = 0;
        RC(run_wgrad(E.mode, p->enc_x0[k], E.c, E.ldx, nullptr, 0, 0, E.c, dy0, E.ldy, E.c, G(p, p->i_enc_w[E.layer]), B, E.Hin, E.Win, Hout, Wout, 3, 1, 2, sw,
                     G(p, p->i_enc_b[E.layer]), &bias_done, 0, cnt, -slab_bytes, 0, dy_step, ws + g.wg_part[k], &p->enc_desc[k], nullptr, 0, 0,
                     p->enc_started[k] ? 0 : 1));
        if (!bias_done)
            for (int j = 0; j < cnt; ++j) RC(bias_grad(dy0 + (size_t)j * (dy_step / 4), E.ldy, E.c, B * Hout * Wout, G(p, p->i_enc_b[E.layer]), sw));
        if (p->side && hipEventRecord(p->ev_done[7 + k][eg_ring], sw) != hipSuccess) return PIVP_ERR_LAUNCH;
        p->enc_started[k] = true; p->enc_desc_valid[k] = true; p->enc_cnt[k] = 0;
        return PIVP_OK;
    };
    // ---- heads (TM:711-728) ----
    if (has_go) {
        if (c.model_type == PIVP_MODEL_CDNA)
            RC(composite_bwd_cdna(prev, ws + S.logits, ws + S.layer0, ws + S.kerns, go, ws + g.dmk, ws + g.dz, ws + g.dkpart,
                                  prev_has_grad ? go_prev : nullptr, 1, B, H, W, c.num_masks, s));
        else if (c.model_type == PIVP_MODEL_STP)
            RC(composite_bwd_stp(prev, ws + S.logits, ws + S.layer0, ws + S.theta, go, ws + g.dmk, ws + g.dz, ws + g.dkpart,
                                 prev_has_grad ? go_prev : nullptr, B, H, W, c.num_masks, c.stp_zero_border, s));
        else
            RC(composite_bwd_dna(prev, ws + S.logits, ws + S.enc7, go, ws + g.dmk, ws + g.dz, prev_has_grad ? go_prev : nullptr, 1,
                                 B, H, W, s));
        RC(mask_softmax_bwd(ws + S.logits, ws + g.dmk, B, HW, p->NP, s));
        RC(heads_bwd(ws + S.e6, P(p, p->i_masks_w), P(p, p->i_enc7_w), ws + g.dmk, ws + g.dz, ws + g.e6, G(p, p->i_masks_w),
                     G(p, p->i_masks_b), G(p, p->i_enc7_w), G(p, p->i_enc7_b), B, HW, p->NP, p->NE, s));
        RC(join(13));      // d v (the kernel generator's weight gradient reads it)
        if (c.model_type == PIVP_MODEL_CDNA)
            RC(cdna_kernels_bwd(ws + S.n5, P(p, p->i_head_w), ws + S.vpre, ws + g.dkpart, composite_bwd_tiles(H, W), ws + g.dv[par], ws + g.n5, 0,
                                G(p, p->i_head_w), G(p, p->i_head_b), B, p->K5, c.num_masks, s, fork_of(13, fe)));
        else if (c.model_type == PIVP_MODEL_STP)
            RC(stp_params_bwd(ws + S.n5, P(p, p->i_head_w), ws + S.vpre, P(p, p->i_head2_w), ws + g.dkpart, composite_bwd_tiles(H, W),
                              ws + g.dv[par], ws + g.n5, G(p, p->i_head_w), G(p, p->i_head_b), G(p, p->i_head2_w), G(p, p->i_head2_b),
                              B, p->K5, s));
        else if (hipMemsetAsync(ws + g.n5, 0, (size_t)px8 * 128 * 4, s) != hipSuccess) return PIVP_ERR_LAUNCH;   // DNA: no hidden5 head
        // group 6 (TM:601), reversed: norm_enc6 (+relu) <- enc6 deconv <- [hidden7 | enc0]
        RC(join(7));       // d e6raw
        RC(lnb(8, ws + g.e6, 64, ws + S.e6, 64, ws + S.e6raw, d_e6raw, 64 * HW, 64, 1));
        enc_add(0, ws + S.cat7);
        RC(run_conv_backward(1, ws + S.cat7, 64, 64, P(p, p->i_enc_w[6]), d_e6raw, 64, 64, nullptr, 0, ws + g.wt_enc[6], ws + g.cat7, 64, 0,
                             nullptr, nullptr, B, p->H2, p->W2, s, 1, eg_flush ? fork_of(7, fe) : nullptr, nullptr, nullptr, nullptr, 0, encp));
        if (eg_flush) RC(enc_flush(0, true));
    } else {
        if (eg_flush) RC(enc_flush(0, false));      // a batch whose last steps no frame gradient reaches
        // no gradient reaches this step's frame: only the recurrent paths are live
        if (hipMemsetAsync(ws + g.cat7, 0, (size_t)px2 * 64 * 4, s) != hipSuccess) return PIVP_ERR_LAUNCH;
        if (hipMemsetAsync(ws + g.n5, 0, (size_t)px8 * 128 * 4, s) != hipSuccess) return PIVP_ERR_LAUNCH;
    }
    if (t == 0) RC(ln_finish(8));   // norm_enc6's backward does not run in a step no gradient reaches (the usual t = 0): finish it here
    // enc conv k's partial weight-gradient sums -> its gradient (once per sweep, behind its last weight-gradient launch)
    auto reduce_enc = [&](int k) -> int {
        if (!p->enc_desc_valid[k]) return PIVP_OK;
        p->enc_desc_valid[k] = false;
        hipStream_t sw = p->side ? p->side_of(7 + k) : s;
        RC(igemm_wgrad_reduce(p->enc_desc[k], sw));
        if (p->side && hipEventRecord(p->ev_done[7 + k][eg_ring], sw) != hipSuccess) return PIVP_ERR_LAUNCH;      // (t = 0: behind this step's own launch, same ring)
        return PIVP_OK;
    };
    // t = 0 is the sweep's final timestep: a gradient group is final once the side stream's weight gradients of its layers are in too
    auto done = [&](int group) -> int {
        if (t == 0 && !p->grad_cb) {       // no listener: still turn the group's partial sums into gradients
            static const int encs[6][1] = {{0}, {-1}, {1}, {2}, {3}, {4}};
            if (encs[group][0] >= 0) RC(reduce_enc(encs[group][0]));
        }
        if (t != 0 || !p->grad_cb) return PIVP_OK;
        const int (&slots)[6][4] = kGroupSlots;
        for (int k = 0; k < 4; ++k) {
            const int sl = slots[group][k];
            if (sl >= 7 && sl < 12) RC(reduce_enc(sl - 7));          // behind the conv's last weight-gradient launch, on the same stream
            if (sl >= 0 && p->group_join) RC(wait_slot(p, sl, s));
        }
        p->grad_cb(p->grad_cb_user, group);
        return PIVP_OK;
    };
    RC(done(0));
    RC(lnb_cell(6, ws + g.cat7, 64, n2, 32));
    RC(join(8));           // enc5's dY = the x part of lstm7's d_in: a ring slot (a batch's first step waits for the ring's previous weight-gradient launch)
    // The x columns of a cell's input gradient are the dY of the enc conv in front of the cell: its ReLU mask (enc5, enc4) or the second gradient path into
    // the same tensor (enc0: + enc6's concat part) is met in the data gradient's epilogue where that kernel has the hook and its grid is unsplit (the bf16 /
    // split-precision forms: 27 relu_mask / add_strided launches fewer per train step at B = 32); otherwise the separate pass runs as before.
    int ep_ok6 = 0, ep_ok5 = 0, ep_ok0 = 0;
    const EpSpec ep6{ws + S.e5, 96, 96, 1, &ep_ok6}, ep5{ws + S.e4, 128, 128, 1, &ep_ok5}, ep0{ws + g.cat7 + 32, 64, 32, 2, &ep_ok0};
    RC(lstmb(6, ws + S.e5, 96, p->H2, p->W2, &ep6));
    RC(done(1));
    RC(join(11));          // enc1's dY lives in d cat6, which enc5's data gradient rewrites
    // group 5 (TM:600): d e5 = x-part of lstm7's d_in (ReLU fused in enc5) <- enc5 deconv <- [hidden6 | enc1]
    enc_add(1, ws + S.cat6);
    RC(run_conv_backward(1, ws + S.cat6, 96, 96, P(p, p->i_enc_w[5]), DIN(6, true), 96, 128, ep_ok6 ? nullptr : ws + S.e5, 96, ws + g.wt_enc[5], d_cat6, 96, 0,
                         nullptr, nullptr, B, p->H4, p->W4, s, 1, eg_flush ? fork_of(8, fe) : nullptr, nullptr, nullptr, nullptr, 0, encp));
    if (eg_flush) RC(enc_flush(1, true));
    RC(lnb_cell(5, d_cat6, 96, n4, 64));
    RC(join(9));           // enc4's dY = the x part of lstm6's d_in, likewise
    RC(lstmb(5, ws + S.e4, 128, p->H4, p->W4, &ep5));
    RC(done(2));
    // group 4 (TM:599): d e4 = x-part of lstm6's d_in <- enc4 deconv <- hidden5 (also read by the CDNA kernel generator)
    enc_add(2, ws + S.n5);
    RC(run_conv_backward(1, ws + S.n5, 128, 128, P(p, p->i_enc_w[4]), DIN(5, true), 128, 192, ep_ok5 ? nullptr : ws + S.e4, 128, ws + g.wt_enc[4], ws + g.n5, 128, 1,
                         nullptr, nullptr, B, p->H8, p->W8, s, 1, eg_flush ? fork_of(9, fe) : nullptr, nullptr, nullptr, nullptr, 0, encp));
    if (eg_flush) RC(enc_flush(2, true));
    RC(lnb_cell(4, ws + g.n5, 128, n8, 128));
    RC(lstmb(4, ws + S.e3, 64, p->H8, p->W8));
    RC(done(3));
    // group 3 (TM:598) + state predictor (TM:730): d e3 = x-part of lstm5's d_in (ld 192)
    RC(join(10));          // d e2
    RC(enc3_state_bwd(ws + S.e2, ws + S.e3, DIN(4, true), 192, action, state_prev, P(p, p->i_enc_w[3]), P(p, p->i_cs_w),
                      ws + g.dstate + (size_t)t * B * 5, d_e2, G(p, p->i_enc_w[3]), G(p, p->i_enc_b[3]), G(p, p->i_cs_w),
                      G(p, p->i_cs_b), t > 0 ? ws + g.dstate + (size_t)(t - 1) * B * 5 : ws + g.dstate + (size_t)(c.seq_len - 1) * B * 5,
                      B, p->H8 * p->W8, c.use_state, s, 1));      // d e2 comes out masked by enc2's ReLU
    // group 2 (TM:597): enc2 conv (ReLU) <- hidden4 <- lstm4 <- hidden3 <- lstm3 <- enc1
    enc_add(3, ws + S.n4);
    RC(run_conv_backward(0, ws + S.n4, 64, 64, P(p, p->i_enc_w[2]), d_e2, 64, 64, nullptr, 0, ws + g.wt_enc[2], ws + g.n4, 64, 0,
                         nullptr, nullptr, B, p->H4, p->W4, s, 1, eg_flush ? fork_of(10, fe) : nullptr, nullptr, nullptr, nullptr, 0, encp));
    if (eg_flush) RC(enc_flush(3, true));
    RC(lnb_cell(3, ws + g.n4, 64, n4, 64));
    RC(lstmb(3, ws + S.n3, 64, p->H4, p->W4));      // its data gradient's x columns = the dy of hidden3
    RC(lnb_cell(2, DIN(3, true), 128, n4, 64));
    RC(lstmb(2, ws + S.cat6 + 64, 96, p->H4, p->W4));
    RC(done(4));
    // group 1 (TM:596): enc1 conv (ReLU) <- hidden2 <- lstm2 <- hidden1 <- lstm1 <- enc0
    enc_add(4, ws + S.n2);
    RC(run_conv_backward(0, ws + S.n2, 32, 32, P(p, p->i_enc_w[1]), d_cat6 + 64, 32, 96, ws + S.cat6 + 64, 96, ws + g.wt_enc[1], ws + g.n2, 32, 0,
                         nullptr, nullptr, B, p->H2, p->W2, s, 1, eg_flush ? fork_of(11, fe) : nullptr, nullptr, nullptr,
                         DIN(2, true), 96, encp));      // d enc1 = enc5's concat part (in d cat6) + lstm3's x gradient, summed in the ReLU-mask pass
    if (eg_flush) RC(enc_flush(4, true));
    RC(lnb_cell(1, ws + g.n2, 32, n2, 32));
    RC(lstmb(1, ws + S.n1, 32, p->H2, p->W2));      // ... of hidden1
    RC(lnb_cell(0, DIN(1, true), 64, n2, 32));
    RC(lstmb(0, ws + S.cat7 + 32, 64, p->H2, p->W2, &ep0));
    if (!ep_ok0) RC(add_strided(ws + g.cat7 + 32, 64, DIN(0, true), 64, 32, px2, s));          // d enc0: from enc6's concat + from lstm1
    // group 0 (TM:595): norm_enc0 (+relu) <- enc0 conv <- frame
    RC(join(12));          // d e0raw
    RC(lnb(0, ep_ok0 ? DIN(0, true) : ws + g.cat7 + 32, 64, ws + S.cat7 + 32, 64, ws + S.e0raw, ws + g.e0raw[par], n2, 32, 1));
    RC(enc0_bwd(prev, P(p, p->i_enc_w[0]), ws + g.e0raw[par], G(p, p->i_enc_w[0]), G(p, p->i_enc_b[0]), prev_has_grad ? go_prev : nullptr, 1,
                B, H, W, s, fork_of(12, fe)));
    RC(done(5));
    return PIVP_OK;
}

static int rollout_backward_sweep(pivp_plan_t* plan, const float* images, const float* actions, const float* states,
                                  const unsigned char* gt_select, const float* gen_images, const float* gen_states, void* stream);
extern "C" int pivp_rollout_backward(pivp_plan_t* plan, const float* images, const float* actions, const float* states,
                                     const unsigned char* gt_select, const float* gen_images, const float* gen_states, void* stream) {
    if (!plan) return PIVP_ERR_BADARG;
    const int rc = rollout_backward_sweep(plan, images, actions, states, gt_select, gen_images, gen_states, stream);
    // The side stream's weight gradients are joined whatever the sweep returned: on success the caller's stream is made to wait for
    // them (what it enqueues next -- the all-reduce, Adam -- sees every gradient); after a failure half-way the host waits for both
    // streams, so nothing is still reading the workspace or writing gradients when the caller clears, reuses or frees them.
    if (plan->side) {
        hipStream_t s = (hipStream_t)stream;
        bool joined = rc == PIVP_OK;
        if (joined) {
            for (int i = 7; i < pivp_plan::NSLOT && joined; ++i)
                for (int r = 0; r < 2 && joined; ++r) joined = hipStreamWaitEvent(s, plan->ev_done[i][r], 0) == hipSuccess;
            for (int i = 0; i < 7 && joined; ++i)
                for (int r = 0; r < 2 && joined; ++r) joined = hipStreamWaitEvent(s, plan->ev_ring_done[i][r], 0) == hipSuccess;
        }
        if (!joined) {
            (void)hipStreamSynchronize(plan->side);
            (void)hipStreamSynchronize(s);
            return rc != PIVP_OK ? rc : PIVP_ERR_LAUNCH;
        }
    }
    return rc;
}
static int rollout_backward_sweep(pivp_plan_t* plan, const float* images, const float* actions, const float* states,
                                  const unsigned char* gt_select, const float* gen_images, const float* gen_states, void* stream) {
    if (!plan || !images || !actions || !states || !gen_images || !gen_states) return PIVP_ERR_BADARG;
    if (!plan->ws || !plan->has_grads || plan->last_steps != plan->cfg.seq_len - 1) return PIVP_ERR_STATE;
    if ((gt_select != nullptr) != plan->last_sched) return PIVP_ERR_STATE;
    for (const ParamInfo& pi : plan->params) if (!pi.ptr || !pi.grad) return PIVP_ERR_STATE;
    hipStream_t s = (hipStream_t)stream;
    const pivp_config_t& c = plan->cfg;
    const int B = c.batch, T = c.seq_len, ctx = c.context_frames;
    const size_t fr = (size_t)B * 3 * c.height * c.width;
    float* ws = plan->ws;
    const Grads& g = plan->g;
    const float fscale = 2.0f / ((float)fr * (float)(T - ctx));              // d/d gen of mean-squared error / (T - ctx)
    const float sscale = 2.0f * 1e-4f / ((float)(B * 5) * (float)(T - ctx));
    RC(ensure_side(plan));
    RC(apply_main_prio(plan, s));
    {   // Timesteps per ConvLSTM weight-gradient launch (PIVP_WGRAD_BATCH overrides, 1..wg_cap).  fp32: 1 (batches arrive in bursts and
        // overlap the sweep worse: 29.9 / 30.3 ms for 1 / 2, profiles/r02).  bf16 mode: as many as the rings hold -- its 25-tap kernel
        // fetches every operand tile once per 32 x 64 output slice, and what a block pays per launch (205 KB of atomics, the first tile's
        // latency) is amortised over the batch (csrc/wgrad_bf16.hip).
        const int e = wgrad_batch_env();
        int gb = e ? e : ((plan->bf16_all || plan->x3_wgrad || plan->x6_wgrad) ? plan->wg_cap : 1);
        if (gb < 1) gb = 1;
        if (gb > plan->wg_cap) gb = plan->wg_cap;
        plan->wg_batch = gb;
    }
    // the norms' partial parameter gradients start from zero every sweep; the enc convs' partial planes are STORED by each layer's first launch of the
    // sweep (WgradDesc::part_overwrite; 240 MB of planes at B = 32: a memset of them would cost what the weight gradients do)
    if (hipMemsetAsync(ws + g.ln_ppart[0], 0, g.ln_ppart_floats * 4, s) != hipSuccess) return PIVP_ERR_LAUNCH;
    for (int j = 0; j < 9; ++j) plan->ln_touched[j] = false;
    for (int k = 0; k < 5; ++k) { plan->enc_desc_valid[k] = false; plan->enc_cnt[k] = 0; plan->enc_started[k] = false; }
    // d loss / d gen_states[t] for every t (zero before ctx-1), later accumulated with the state recurrence
    if (hipMemsetAsync(ws + g.dstate, 0, (size_t)T * B * 5 * 4, s) != hipSuccess) return PIVP_ERR_LAUNCH;
    // (gen_states[t] against states[t + 1], t = ctx-1 .. T-2: contiguous in t, one launch; likewise the frames' loss terms below -- 16 launches per sweep before)
    RC(scaled_diff(gen_states + (size_t)(ctx - 1) * B * 5, states + (size_t)ctx * B * 5, ws + g.dstate + (size_t)(ctx - 1) * B * 5, (long)(T - ctx) * B * 5, sscale, 0, s));
    // d loss / d gen[t], t = ctx-1 .. T-2 (gen[t] is compared with images[t + 1]); the sweep adds the feed-back terms of step t + 1 into frame t in place
    RC(scaled_diff(gen_images + (size_t)(ctx - 1) * fr, images + (size_t)ctx * fr, ws + g.go + (size_t)(ctx - 1) * fr, (long)(T - ctx) * (long)fr, fscale, 0, s));
    // weights are constant during the sweep: the transposed packs for the data gradients, all twelve in one launch (7 + 5 launches before round 5)
    {
        const int ecin[7] = {0, 32, 64, 0, 128, 96, 64};
        WeightPrepJob jobs[12];
        int n = 0;
        for (int i = 0; i < 7; ++i)
            jobs[n++] = WeightPrepJob{0, P(plan, plan->i_lstm_w[i]), ws + g.wt_lstm[i], 25, kLstm[i].cx + kLstm[i].C, 4 * kLstm[i].C, 1};
        for (int i : {1, 2, 4, 5, 6}) jobs[n++] = WeightPrepJob{0, P(plan, plan->i_enc_w[i]), ws + g.wt_enc[i], 9, ecin[i], ecin[i], 0};
        RC(weight_prep_batch(jobs, n, s));
    }
    if (plan->lstm_bf16 && plan->bwd_planes == 1) {      // bf16 mode: the ConvLSTM data gradients run on one-plane bf16 packs of those: one launch
        WeightPrepJob jobs[7];
        for (int i = 0; i < 7; ++i) {
            const int cin = kLstm[i].cx + kLstm[i].C;
            jobs[i] = WeightPrepJob{1, ws + g.wt_lstm[i], ws + g.wtb_lstm[i], 4 * kLstm[i].C, cin, conv5x5_bf16_rows(cin), 0};
        }
        RC(weight_prep_batch(jobs, 7, s));
    } else if (plan->lstm_bf16)        // split modes: the hi / lo pair, the three pieces or the fp16 pieces (their own pack kernels, per layer)
        for (int i = 0; i < 7; ++i) {
            const int cin = kLstm[i].cx + kLstm[i].C;
            RC(pack_lstm_bf16(ws + g.wt_lstm[i], reinterpret_cast<unsigned short*>(ws + g.wtb_lstm[i]), 4 * kLstm[i].C, cin, s,
                              conv5x5_bf16_rows(cin), plan->bwd_planes,
                              1));
        }
    bool has_go = false;
    int eq_next = 0;      // the enc dY ring slot of step t + 1 (unused at t = T - 2)
    for (int t = T - 2; t >= 0; --t) {
        float* go = ws + g.go + (size_t)t * fr;
        float* go_prev = t >= 1 ? ws + g.go + (size_t)(t - 1) * fr : nullptr;
        const bool last = t == T - 2;
        if (last) has_go = true;                                            // d gen[T-2] comes from the loss only
        const bool prev_has_grad = (t >= ctx) && !gt_select;                // feed-self: prev = gen[t-1] is differentiable (TM:664-666)
        // d gen[t-1] holds its loss term (if it has one: t - 1 >= ctx - 1, which prev_has_grad implies); this step's composite / enc0 backward add the feed-back term
        const bool next_loss = (t - 1) >= ctx - 1 && t >= 1;
        const float* prev = step_input(plan, t, images, gt_select, gen_images, fr);
        const float* st_prev = t == 0 ? states : gen_states + (size_t)(t - 1) * B * 5;
        // the ConvLSTM weight gradients: batches of wg_batch timesteps counted from the top of the sweep; t = 0 on its own
        // With deep batches (G > 2: the bf16 mode's 8) the LAST TWO batched timesteps (t = 2, 1) form a batch of their own: one batch of everything
        // is launched at t = 1, when two timesteps of main-stream work are left for 1.4 ms of weight gradients to hide behind -- the sweep then ended with
        // the main stream waiting ~0.2 ms per step for the side stream (profiles/r05/NOTES.md).
        auto sched = [&](int G, int& b, int& slot, bool& flush) {
            const int k = T - 2 - t, n = T - 2;
            const int tail = (G > 2 && n > 2) ? 2 : 0, body = n - tail, nb_body = (body + G - 1) / G;
            if (t == 0) { b = nb_body + (tail ? 1 : 0); slot = 0; flush = true; }
            else if (k < body) { b = k / G; slot = k % G; flush = slot == G - 1 || k == body - 1; }
            else { b = nb_body; slot = k - body; flush = k == n - 1; }
        };
        int wg_b, wg_slot, eg_b, eg_slot; bool wg_flush, eg_flush;
        sched(plan->wg_batch, wg_b, wg_slot, wg_flush);
        // (1 / 2 / 4 timesteps per launch instead of the rings' 8, r05_c23: config 3 11.63 / 11.24 / 11.22 ms against 11.10, fp32 train 28.08 / 27.90 / 27.67 against 27.41)
        sched(plan->eg_cap, eg_b, eg_slot, eg_flush);      // the stride-2 3x3 layers' weight gradients: as many timesteps per launch as their rings hold, every mode
        RC(backward_step(plan, t, prev, prev_has_grad && has_go, actions + (size_t)t * B * 5, st_prev, has_go, go, go_prev, last,
                         wg_b & 1, wg_slot, wg_flush, eg_b & 1, eg_slot, eg_flush, eq_next, s));
        eq_next = (eg_b & 1) * plan->eg_cap + eg_slot;
        has_go = next_loss || (prev_has_grad && has_go);
    }
    return PIVP_OK;      // the side stream is joined by the caller, pivp_rollout_backward, on every path
}

// ------------------------------------------------------------------------------------------------------------
// measurement hooks and taps
// ------------------------------------------------------------------------------------------------------------
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
    if (strcmp(name, "enc6") == 0 && !c.keep_activations) {
        // inference does not materialise relu(norm_enc6(.)) (fused into the heads kernel): rebuild it from the raw map
        // and the saved (mean, rstd)
        int rc = ln_apply(ws + S.e6raw, ws + S.lnstat + (size_t)8 * B * 2, P(plan, plan->i_ln_g[8]), P(plan, plan->i_ln_b[8]),
                          ws + S.e6, B, 64 * HW, 64, 64, c.ln_eps, 1, s, nullptr, -1);
        if (rc != PIVP_OK) return rc;
    }
    if (!c.keep_activations) {
        // inference applies the norms of hidden2 / hidden4 / hidden6 / hidden7 inside their consumers (run_step): rebuild the tensor on request from
        // the raw ConvLSTM output (statistics recomputed by ln_stats: equal to the fused ones up to fp32 summation order)
        struct F { const char* n; int layer, norm; size_t dst; int C, hw, ld; int split_only = 0; };
        const F folded[] = {{"hidden2", 1, 2, S.n2, 32, HW2, 32}, {"hidden4", 3, 4, S.n4, 64, HW4, 64}, {"hidden6", 5, 6, S.cat6, 64, HW4, 96},
                            {"hidden7", 6, 7, S.cat7, 32, HW2, 64},
                            {"hidden1", 0, 1, S.n1, 32, HW2, 32, 1}, {"hidden3", 2, 3, S.n3, 64, HW4, 64, 1}};      // (the split modes: inside lstm2 / lstm4)
        const bool split = plan->lstm_planes == 3 || plan->lstm_planes == -2;
        for (const F& f : folded)
            if (strcmp(f.n, name) == 0 && (!f.split_only || split)) {
                int rc = run_layernorm(ws + S.h[f.layer], P(plan, plan->i_ln_g[f.norm]), P(plan, plan->i_ln_b[f.norm]), ws + f.dst,
                                       ws + plan->o_lnpart, B, f.C * f.hw, f.C, f.ld, c.ln_eps, 0, s, nullptr, 0);
                if (rc != PIVP_OK) return rc;
            }
    }
    for (const T& t : taps) {
        if (strcmp(t.n, name) == 0) {
            int rc = nhwc_to_nchw(ws + t.off, out, B, t.C, t.hw, t.ld, s);
            return rc == PIVP_OK ? (long long)B * t.C * t.hw : rc;
        }
    }
    size_t off = 0, n = 0;
    if (strcmp(name, "enc7") == 0) { off = S.enc7; n = (size_t)B * plan->NE * HW; }
    else if (strcmp(name, "cdna_kerns") == 0 && c.model_type == PIVP_MODEL_CDNA) { off = S.kerns; n = (size_t)B * 25 * c.num_masks; }
    else if (strcmp(name, "stp_theta") == 0 && c.model_type == PIVP_MODEL_STP) { off = S.theta; n = (size_t)B * 6; }
    else if (strcmp(name, "masks") == 0) {
        if (step != plan->last_steps - 1) return PIVP_ERR_STATE;      // softmaxed masks exist only for the most recent step
        off = plan->o_masks; n = (size_t)B * plan->NP * HW;
    } else return PIVP_ERR_BADARG;
    if (hipMemcpyAsync(out, ws + off, n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return PIVP_ERR_LAUNCH;
    return (long long)n;
}
