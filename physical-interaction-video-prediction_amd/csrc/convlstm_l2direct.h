#pragma once
#include "convlstm_bf16_common.h"

namespace pivp {

// =================================================================================================================================
// Three-piece ConvLSTM with the weights read STRAIGHT FROM L2 into the MFMA's operand registers: no weight ring, no loader waves, no block
// barrier in the tap loop.  The ring form above (16-channel blocks) spends a fifth of its k-step on the LDS-DMAs and the per-k-step barrier,
// and its 16-channel blocks need two rounds on 32 x 32 maps.  Here a block is 128 anchors x 32 channels x 4 gates and all eight waves
// multiply (2 x 4: wave tile = 64 anchors x the four gates of 8 channels, two waves per SIMD); the LDS holds only the three patch planes.
// A wave's B fragment of a (tap, k-step, plane) is 1 KB of the fragment-major pack: one coalesced global_load_dwordx4 per wave, requested
// FOUR k-steps (one tap) ahead into a register ring of 4 x 3 fragments: with two waves per SIMD a wave's k-step lasts ~0.45 us, so a tap of
// lookahead covers an L2 round trip of 1-2 us; the two waves that share a fragment (wm = 0 / 1) ask for the same lines at about the same time.  Same arithmetic, term for term, as the ring form.
// =================================================================================================================================
// NWM x NWN = eight waves over the 128 anchors x (8 NWN channels x 4 gates) of a block: 2 x 4 (32 channels) or, for layers whose 32-channel blocks
// would leave CUs idle, 4 x 2 (16 channels, still two waves per SIMD: a wave's tile is 32 anchors).  (A four-wave 2 x 2 form with a fragment ring two
// taps deep, 16 x 16-anchor tiles, fragments shared through LDS, two k-steps per wait, loads issued mid-k-step: all built and measured in round 4,
// none faster -- profiles/r04/NOTES.md 9-11; in the history.)
// LSTM = false: the plain 5x5 convolution with the same loop (the data gradient): a wave's 32 columns are consecutive output columns, the channel
// groups may be split over gridDim.y (partial sums then meet in `out` by atomic adds), the epilogue stores / adds the accumulators.
// PCS = 2: TWO FP16 pieces per operand instead of three bf16 ones (22 bits of operand mantissa; the weights arrive times 2^8 and the sum is scaled
// back) and three MFMAs per product: hi*hi on the main accumulator, lo*hi + hi*lo on the second.  scripts/split_fp16_study.py: the truncation is a
// quarter of the fp32 path's own error.  Forward gate convolutions only (gradients are too small for fp16's exponent range).
// IN_LN: the x operand (d.x0, c0 <= 64 channels) is a RAW ConvLSTM output whose LayerNorm (per-element gamma / beta d.in_g / d.in_b [H W][c0], statistics
// merged from the producer's partials d.in_part) is applied while the patch is staged -- the expression of ln_apply_kernel; out-of-image pixels load 0
// for v, gamma and beta alike and stay 0.  Inference rollouts: hidden1 -> lstm2 and hidden3 -> lstm4 lose their ln_apply launch.
// W8 (round 5): maps 8 pixels wide (lstm5 on 64 x 64 frames) -- the block's 128 anchors are 8 x 8 pixels of TWO images (an even batch), the patch two
// 12 x 12-pixel images with the ring kernel's row pitch RP8 (three planes: 138 KB, which is why the LDS-ring form never had room for it); everything
// else -- fragments, k-steps, epilogue -- is the 16-wide form's.  Not with IN_LN (two images = two sets of statistics).
template <int NWM, int NWN, bool LSTM = true, int PCS = 3, bool IN_LN = false, bool W8 = false>
__global__ __launch_bounds__(64 * NWM * NWN, 1) void convlstm_x6g_kernel(const IgemmDesc d, const unsigned short* __restrict__ wb, int wbytes, int ncols) {
    constexpr int THX = TH, PHX = PH;                  // anchor rows per tile, patch rows
    constexpr int TW = W8 ? 8 : 16, TIN = W8 ? 2 : 1;  // tile width in pixels, images per tile
    constexpr int RP = W8 ? RP8 : RP16;                // patch row pitch (bytes)
    constexpr int PB = TIN * PHX * RP;                 // one patch plane: 36,864 B (W8: 46,080 B)
    static_assert(!(W8 && IN_LN), "LayerNorm-on-load: one image per tile");
    static_assert(PCS == 3 || PCS == 2, "pieces");
    static_assert(NWM * NWN == 8, "eight waves, two per SIMD");
    constexpr int PW = TW + 4;
    constexpr int NW = NWM * NWN;                      // waves
    constexpr int MT = THX / 2 / NWM;                  // 32-anchor M tiles per wave
    constexpr int NT = 64 * NW;                        // threads
    constexpr int PPP = NT / 8;                        // patch pixels per staging pass
    constexpr int NPJX = IN_LN ? 2 : 4;                // staging passes per round: 4 x 64 pixels cover the patch's 240 with 512 threads (IN_LN: gamma and beta
                                                       // travel with the pixels: two rounds of 2, or the prologue spills);
    constexpr int NPIX = TIN * PHX * PW;               // patch pixels: 240 (W8: 288)
    constexpr int NRND = (NPIX + NPJX * PPP - 1) / (NPJX * PPP);     // 256 threads take two rounds of 4 x 32 (eight passes in one round put the staged pixels in scratch)
    constexpr int RD = 4;                              // k-steps (one tap) of B fragments in registers
    PIVP_SET_MAIN_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const patch = lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave8 % NWM, wn = wave8 / NWM;
    const int half = lane >> 5, l31 = lane & 31;
    const int C = d.C;
    const int n_nblk = LSTM ? C / (8 * NWN) : d.N / (32 * NWN);      // (plain: d.N = rows of the padded pack, a multiple of 64)
    const int H = d.Hin, W = d.Win;
    const int tpr = W / TW, tpi = (H / THX) * tpr;
    const int n_tiles = (d.B / TIN) * tpi;
    int lid = blockIdx.x;
    if ((gridDim.x & 7) == 0) lid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-aware, column-block major
    const int nblk = pivp_fdiv(lid, d.fd_mb_mul, d.fd_mb_sh), tile = lid - nblk * n_tiles;      // (the launcher's multipliers: pivp_fastdiv)
    const int timg = pivp_fdiv(tile, d.fd_hw_mul, d.fd_hw_sh);
    const int b0 = timg * TIN, trem = tile - timg * tpi;
    const int trow = pivp_fdiv(trem, d.fd_w_mul, d.fd_w_sh);
    const int y0 = trow * THX, x0 = (trem - trow * tpr) * TW;
    // anchor i of the block's 128 -> (image of the tile, row, column): 8 x 16 of one image, or 8 x 8 of two
    auto anchor = [&](int i, int& ti, int& ay, int& ax) {
        if constexpr (W8) { ti = i >> 6; ay = (i >> 3) & 7; ax = i & 7; }
        else { ti = 0; ay = i >> 4; ax = i & 15; }
    };
    BF_STAMP(0);
    const int c0 = d.c0, ld0 = d.ld0, ld1 = d.ld1;
    const int cin = c0 + d.c1;
    const int ncg_all = (cin + 63) >> 6;
    const int cgbase = (int)blockIdx.y * ncg_all / (int)gridDim.y;                 // this block's channel groups: [cgbase, cgbase + ncg)
    const int ncg = ((int)blockIdx.y + 1) * ncg_all / (int)gridDim.y - cgbase;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x0), 0, d.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.c1 ? d.x1 : d.x0), 0, d.c1 ? d.bytes1 : d.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wb), 0, wbytes, 0x00020000);
    constexpr unsigned OOB = 0xC0000000u;

    // ---- patch staging (all 8 waves), as in convlstm_bf16_kernel: thread = (pixel (tid >> 3) + 64 j, 8-channel piece tid & 7), three planes ----
    const int cpiece = tid & 7;
    // (pixel -> image / patch offsets are recomputed where they are used: ten index registers held across the tap loop cost more than the divisions)
    auto pix_of = [&](int j, int& a_pix, int& a_lds) {      // j: pass index over all rounds
        const int p = (tid >> 3) + PPP * j;
        const int ti = W8 ? p / (PHX * PW) : 0, pr = p - ti * (PHX * PW);
        const int py = pr / PW, px = pr - py * PW;
        const int iy = y0 - 2 + py, ix = x0 - 2 + px;
        const bool ok = p < NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        a_pix = ok ? ((b0 + ti) * H + iy) * W + ix : -1;
        a_lds = p < NPIX ? (ti * PHX + py) * RP + px * PP : PW * PP;
    };
    float a_scale = 1.0f;                              // (plain form with fp16 pieces: see inv_wscale below)
    if constexpr (PCS == 2 && !LSTM) a_scale = pivp_x3_scale_wave(d.wscale_part);
    f32x4 plo[NPJX], phi[NPJX];
    f32x4 glo[IN_LN ? NPJX : 1], ghi[IN_LN ? NPJX : 1], blo[IN_LN ? NPJX : 1], bhi[IN_LN ? NPJX : 1];      // IN_LN: gamma / beta of the staged pieces
    float ln_mean = 0.f, ln_rstd = 1.f;
    if constexpr (IN_LN) ln_merge_partials(d.in_part, b0, d.in_np, d.in_eps, ln_mean, ln_rstd);      // every wave for itself: a few partials per sample
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(IN_LN ? d.in_g : d.x0), 0, IN_LN ? H * W * c0 * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(IN_LN ? d.in_b : d.x0), 0, IN_LN ? H * W * c0 * 4 : 0, 0x00020000);
    auto patch_load = [&](int cg, int rnd) {
        const int gch = cg * 64;                          // block-uniform: the source choice below is a uniform branch (see convlstm_ring.h: as a
        const int ch = gch + cpiece * 8;                  // per-lane choice of descriptor every load was a 14-instruction waterfall loop)
        const bool s0 = ch < c0, s1 = !s0 && ch < cin;
        const bool use0 = gch < c0, use1 = gch + 64 > c0 && gch < cin;
        const int co = s0 ? ch : ch - c0;
#pragma unroll
        for (int j = 0; j < NPJX; ++j) {
            int a_pix, a_lds;
            pix_of(rnd * NPJX + j, a_pix, a_lds);
            const unsigned off0 = (a_pix >= 0 && s0) ? (unsigned)((a_pix * ld0 + ch) * 4) : OOB;
            const unsigned off1 = (a_pix >= 0 && s1) ? (unsigned)((a_pix * ld1 + ch - c0) * 4) : OOB;
            if constexpr (IN_LN) {          // (pieces of h channels and pixels outside the image: the zeros of an out-of-range load)
                const unsigned go = (a_pix >= 0 && s0) ? (unsigned)(((a_pix - b0 * H * W) * c0 + co) * 4) : OOB;
                glo[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsg, go, 0, 0));
                ghi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsg, go, 16, 0));
                blo[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, go, 0, 0));
                bhi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, go, 16, 0));
            }
            if (use0 && use1) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 alo = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, off0, 0, 0)), ahi = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, off0, 16, 0));
                const u32x4 blo2 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, off1, 0, 0)), bhi2 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, off1, 16, 0));
                plo[j] = __builtin_bit_cast(f32x4, alo | blo2);      // (one of the two is the hardware's zeros)
                phi[j] = __builtin_bit_cast(f32x4, ahi | bhi2);
            } else if (use0) {
                plo[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, off0, 0, 0));
                phi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, off0, 16, 0));
            } else {
                plo[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, off1, 0, 0));
                phi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, off1, 16, 0));
            }
        }
    };
    auto patch_store = [&](int rnd, int cg = 0) {      // cg: which 64-channel group was loaded (IN_LN: its x pieces are normalised)
#pragma unroll
        for (int j = 0; j < NPJX; ++j) {
            int a_pix, a_lds;
            pix_of(rnd * NPJX + j, a_pix, a_lds);
            float r[8] = {plo[j][0], plo[j][1], plo[j][2], plo[j][3], phi[j][0], phi[j][1], phi[j][2], phi[j][3]};
            if constexpr (IN_LN) {
                if (cg * 64 + cpiece * 8 < c0) {
                    const float gm[8] = {glo[j][0], glo[j][1], glo[j][2], glo[j][3], ghi[j][0], ghi[j][1], ghi[j][2], ghi[j][3]};
                    const float bt[8] = {blo[j][0], blo[j][1], blo[j][2], blo[j][3], bhi[j][0], bhi[j][1], bhi[j][2], bhi[j][3]};
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = (r[e] - ln_mean) * ln_rstd * gm[e] + bt[e];
                }
            }
            if constexpr (PCS == 2) {
                if constexpr (!LSTM) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] *= a_scale;
                }
                uint4 hh, ll;
                hh.x = pivp_pack2h_rest(r[0], r[1]); hh.y = pivp_pack2h_rest(r[2], r[3]); hh.z = pivp_pack2h_rest(r[4], r[5]); hh.w = pivp_pack2h_rest(r[6], r[7]);
                ll.x = pivp_pack2h_rest(r[0], r[1]); ll.y = pivp_pack2h_rest(r[2], r[3]); ll.z = pivp_pack2h_rest(r[4], r[5]); ll.w = pivp_pack2h_rest(r[6], r[7]);
                *reinterpret_cast<uint4*>(patch + a_lds + cpiece * 16) = hh;
                *reinterpret_cast<uint4*>(patch + PB + a_lds + cpiece * 16) = ll;
                continue;
            }
            uint4 v;
            v.x = pack2(r[0], r[1]); v.y = pack2(r[2], r[3]); v.z = pack2(r[4], r[5]); v.w = pack2(r[6], r[7]);
            *reinterpret_cast<uint4*>(patch + a_lds + cpiece * 16) = v;
            auto rest = [&](unsigned p2, int i) {          // r[i], r[i + 1] become the remainders (exact in fp32); returns them as bf16
                r[i] -= __builtin_bit_cast(float, p2 << 16); r[i + 1] -= __builtin_bit_cast(float, p2 & 0xffff0000u);
                return pack2(r[i], r[i + 1]);
            };
            uint4 l, q;
            l.x = rest(v.x, 0); l.y = rest(v.y, 2); l.z = rest(v.z, 4); l.w = rest(v.w, 6);
            *reinterpret_cast<uint4*>(patch + PB + a_lds + cpiece * 16) = l;
            q.x = rest(l.x, 0); q.y = rest(l.y, 2); q.z = rest(l.z, 4); q.w = rest(l.w, 6);
            *reinterpret_cast<uint4*>(patch + 2 * PB + a_lds + cpiece * 16) = q;
        }
    };
    const int tap0 = (lid * 7) % 25;
    float inv_wscale = 1.0f;
    if constexpr (PCS == 2) inv_wscale = 1.0f / *reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(wb) + wbytes);   // the pack's tail
    // plain form with fp16 pieces (the data gradient): the activations are gradients, far below fp16's normal range -- they are staged times the power of
    // two that puts the tensor's largest |value| into [2^14, 2^15) (d.wscale_part = absmax_partials of d.x0, one partial per lane), and the sums scaled back
    if constexpr (PCS == 2 && !LSTM) inv_wscale *= 1.0f / a_scale;

    f32x16 acc[MT], accl[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[mt][r] = 0.f; accl[mt][r] = 0.f; }
    int a_off[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int ti, ay, ax;
        anchor(32 * MT * wm + 32 * mt + l31, ti, ay, ax);
        a_off[mt] = (ti * PHX + ay) * RP + ax * PP + half * 16;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;

    // ---- the weights: fragment (group, tap, k-step, plane, c8 = nblk * 4 + wn) of the pack, 1 KB in lane order ------------------------
    const unsigned pls = (unsigned)(LSTM ? C / 8 : d.N / 32) * 1024u;    // bytes between the planes of a k-step (one KB per 32-column fragment)
    const unsigned kss = (unsigned)PCS * pls, tps = 4u * kss;     // ... between k-steps, between taps
    const unsigned voff = (unsigned)((nblk * NWN + wn) * 1024 + lane * 16);
    bf16x8 Bf[RD][PCS];                                  // [k-step (of the even / odd tap when RD = 8)][plane]: behind each k-step its registers take the fragments RD k-steps on
    auto bload = [&](bf16x8 (&dst)[PCS], unsigned soff) {
#pragma unroll
        for (int pl = 0; pl < PCS; ++pl)
            dst[pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, voff, (int)(soff + pl * pls), 0));
    };
    auto adv = [&](int& tp, int& cg) { tp = tp == 24 ? 0 : tp + 1; cg += tp == tap0 ? 1 : 0; };

    // ---- prologue ---------------------------------------------------------------------------------------------------------------------
    patch_load(cgbase, 0);
    int tap = tap0, cg = cgbase, tap1 = tap0, cg1 = cgbase;
    adv(tap1, cg1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bload(Bf[ks], (unsigned)(cg * 25 + tap) * tps + ks * kss);
    const int chl = wn * 8 + (l31 & 7);
    const int ch = nblk * 8 * NWN + chl;
    const int grp = l31 >> 3;
    float bj = 0.f, bi = 0.f, bf = 0.f, bo = 0.f;
    float cpre[MT][4];
    patch_store(0, cgbase);
#pragma unroll
    for (int rnd = 1; rnd < NRND; ++rnd) { patch_load(cgbase, rnd); patch_store(rnd, cgbase); }
    BF_STAMP(1);
    __syncthreads();
    BF_STAMP(2);

    bf16x8 fa[2][MT], fal[2][MT], fa3[2][MT];          // [register set][M tile]: the A fragments of a k-step, hi / mid / lo planes
    auto wait_a = [&](auto SET) {
        constexpr int st = decltype(SET)::value;
        if constexpr (PCS == 2 && MT == 2) wait_lgkm(fa[st][0], fa[st][1], fal[st][0], fal[st][1]);
        else if constexpr (PCS == 2) wait_lgkm(fa[st][0], fal[st][0]);
        else if constexpr (MT == 2) wait_lgkm(fa[st][0], fa[st][1], fal[st][0], fal[st][1], fa3[st][0], fa3[st][1]);
        else wait_lgkm(fa[st][0], fal[st][0], fa3[st][0]);
    };
    auto read_a = [&](auto SET, auto KS, auto I, unsigned ab) {       // read I of the 3 MT: plane I / MT, M tile I % MT
        constexpr int st = decltype(SET)::value, ks = decltype(KS)::value, i = decltype(I)::value, pl = i / MT, mt = i % MT;
        bf16x8 v = lds_read_b128<ks * 32>(ab + pl * PB + a_off[mt]);
        if constexpr (pl == 0) fa[st][mt] = v; else if constexpr (pl == 1) fal[st][mt] = v; else fa3[st][mt] = v;
    };
    // twelve MFMAs of register set CUR against the fragments b[3] (hi, mid, lo); corrections into accl, the leading term into acc
    auto mfma = [&](auto CUR, auto I, const bf16x8 (&b)[PCS]) {
        constexpr int st = decltype(CUR)::value, i = decltype(I)::value, term = i / MT, mt = i % MT;
        if constexpr (PCS == 2) {      // fp16 pieces: lo * hi, hi * lo into the corrections, hi * hi into the main accumulator
            auto h = [](const bf16x8& v) { return __builtin_bit_cast(f16x8, v); };
            if constexpr (term == 0) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fal[st][mt]), h(b[0]), accl[mt], 0, 0, 0);
            else if constexpr (term == 1) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fa[st][mt]), h(b[1]), accl[mt], 0, 0, 0);
            else acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fa[st][mt]), h(b[0]), acc[mt], 0, 0, 0);
            return;
        } else {
        if constexpr (term == 0) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa3[st][mt], b[0], accl[mt], 0, 0, 0);        // lo * hi
        else if constexpr (term == 1) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][mt], b[2], accl[mt], 0, 0, 0);    // hi * lo
        else if constexpr (term == 2) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[st][mt], b[1], accl[mt], 0, 0, 0);   // mid * mid
        else if constexpr (term == 3) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[st][mt], b[0], accl[mt], 0, 0, 0);   // mid * hi
        else if constexpr (term == 4) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][mt], b[1], accl[mt], 0, 0, 0);    // hi * mid
        else acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][mt], b[0], acc[mt], 0, 0, 0);                               // hi * hi
        }
    };
    // one k-step: wait for its A fragments, then the MFMAs with the six A reads of the NEXT k-step (set NXT, k-step KSN at patch offset abn)
    // behind the first three
    auto kstep = [&](auto CUR, auto NXT, auto KSN, unsigned abn, const bf16x8 (&b)[PCS], auto RD) {
        constexpr bool rd = decltype(RD)::value;
        wait_a(CUR);
#define PIVP_X6_M(I) mfma(CUR, std::integral_constant<int, I>{}, b);
#define PIVP_X6_R(I) if constexpr (rd) read_a(NXT, KSN, std::integral_constant<int, I>{}, abn);
#define PIVP_X6_S __builtin_amdgcn_sched_barrier(0);
        PIVP_X6_S
        if constexpr (PCS == 2 && MT == 2) {
            PIVP_X6_M(0) PIVP_X6_R(0) PIVP_X6_R(1) PIVP_X6_S
            PIVP_X6_M(1) PIVP_X6_R(2) PIVP_X6_R(3) PIVP_X6_S
            PIVP_X6_M(2) PIVP_X6_S
            PIVP_X6_M(3) PIVP_X6_M(4) PIVP_X6_M(5)
        } else if constexpr (PCS == 2) {
            PIVP_X6_M(0) PIVP_X6_R(0) PIVP_X6_R(1) PIVP_X6_S
            PIVP_X6_M(1) PIVP_X6_S
            PIVP_X6_M(2)
        } else if constexpr (MT == 2) {
            PIVP_X6_M(0) PIVP_X6_R(0) PIVP_X6_R(1) PIVP_X6_S
            PIVP_X6_M(1) PIVP_X6_R(2) PIVP_X6_R(3) PIVP_X6_S
            PIVP_X6_M(2) PIVP_X6_R(4) PIVP_X6_R(5) PIVP_X6_S
            PIVP_X6_M(3) PIVP_X6_S
            PIVP_X6_M(4) PIVP_X6_M(5) PIVP_X6_M(6) PIVP_X6_M(7) PIVP_X6_M(8) PIVP_X6_M(9) PIVP_X6_M(10) PIVP_X6_M(11)
        } else {
            PIVP_X6_M(0) PIVP_X6_R(0) PIVP_X6_R(1) PIVP_X6_S
            PIVP_X6_M(1) PIVP_X6_R(2) PIVP_X6_S
            PIVP_X6_M(2) PIVP_X6_S
            PIVP_X6_M(3) PIVP_X6_M(4) PIVP_X6_M(5)
        }
        PIVP_X6_S
#undef PIVP_X6_M
#undef PIVP_X6_R
#undef PIVP_X6_S
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    auto read_a_all = [&](unsigned ab) {               // the first k-step of a tap into set 0 (prologue, and behind a restaged patch)
        read_a(S0{}, K0{}, I0{}, ab); read_a(S0{}, K0{}, I1{}, ab);
        if constexpr (PCS * MT > 2) read_a(S0{}, K0{}, I2{}, ab);
        if constexpr (PCS * MT > 3) read_a(S0{}, K0{}, I3{}, ab);
        if constexpr (PCS * MT > 4) { read_a(S0{}, K0{}, I4{}, ab); read_a(S0{}, K0{}, I5{}, ab); }
    };
    auto a_base = [&](int tp) { const int ty = tp / 5; return lds0 + ty * RP + (tp - ty * 5) * PP; };

    for (int g = 0; g < ncg; ++g) {                    // 64 input channels of concat(x, h) at a time
        if (g > 0) {                                   // every wave is done with the old patch
            __syncthreads();
#pragma unroll
            for (int rnd = 0; rnd < NRND; ++rnd) { patch_load(cgbase + g, rnd); patch_store(rnd, cgbase + g); }
            __syncthreads();
        }
        if (LSTM && g == ncg - 1) {
            // the epilogue's operands, requested in front of the last 25 taps (read in the epilogue they cost an exposed HBM round trip per row;
            // requested in the prologue they hold 12 registers through every tap loop)
            bj = d.bias[ch]; bi = d.bias[C + ch]; bf = d.bias[2 * C + ch] + 1.0f; bo = d.bias[3 * C + ch];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = k * 4 + grp;
                    int ti, ay, ax;
                    anchor(32 * MT * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half, ti, ay, ax);
                    const int m = ((b0 + ti) * H + y0 + ay) * W + x0 + ax;
                    cpre[mt][k] = d.cstate_in[(size_t)m * C + ch];
                }
        }
        read_a_all(a_base(tap));
        for (int t = 0; t < 25; ++t) {                 // one tap = four k-steps; behind each k-step its registers take the next tap's fragments
            const unsigned ab = a_base(tap), ab1 = a_base(tap1);
            const bool has1 = t < 24 || g + 1 < ncg;
            const unsigned so1 = (unsigned)(cg1 * 25 + tap1) * tps;
            kstep(S0{}, S1{}, K1{}, ab, Bf[0], std::true_type{});
            if (has1) bload(Bf[0], so1);
            kstep(S1{}, S0{}, K2{}, ab, Bf[1], std::true_type{});
            if (has1) bload(Bf[1], so1 + kss);
            kstep(S0{}, S1{}, K3{}, ab, Bf[2], std::true_type{});
            if (has1) bload(Bf[2], so1 + 2 * kss);
            if (t < 24) kstep(S1{}, S0{}, K0{}, ab1, Bf[3], std::true_type{});
            else kstep(S1{}, S0{}, K0{}, ab1, Bf[3], std::false_type{});      // (the next tap's A fragments come from the next patch)
            if (has1) bload(Bf[3], so1 + 3 * kss);
            tap = tap1; cg = cg1;
            adv(tap1, cg1);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[mt][r] += accl[mt][r];
            if constexpr (PCS == 2) acc[mt][r] *= inv_wscale;            // (the weights were packed times a power of two)
        }
    BF_STAMP(3);

    if constexpr (!LSTM) {
        // ---- plain epilogue: accumulator row = anchor, column = output channel; 32 lanes write 128 contiguous bytes -------------------
        // (the epilogue hook's second tensor is requested for all of the lane's outputs first: see convlstm_bf16_kernel)
        float ev[MT][16];
        if (d.ep_mode) {                                // block-uniform
            const int col = (nblk * NWN + wn) * 32 + l31;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int ti, ay, ax;
                    anchor(32 * MT * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half, ti, ay, ax);
                    const size_t m = (size_t)(((b0 + ti) * H + y0 + ay) * W + x0 + ax);
                    ev[mt][r] = (col < ncols && col < d.ep_cols) ? d.ep_src[m * d.ep_ld + col] : (d.ep_mode == 1 ? 1.f : 0.f);
                }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int ti, ay, ax;
                anchor(32 * MT * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half, ti, ay, ax);
                const size_t m = (size_t)(((b0 + ti) * H + y0 + ay) * W + x0 + ax);
                const int col = (nblk * NWN + wn) * 32 + l31;
                if (col < ncols) {                      // the pack's rows past the real column count are zero padding
                    float* o = d.out + m * d.ldo + col;
                    float v = acc[mt][r];
                    if (d.ep_mode == 1) v = ev[mt][r] > 0.f ? v : 0.f;         // (unsplit grids only: the launcher clears ep_mode otherwise)
                    else if (d.ep_mode == 2) v += ev[mt][r];
                    if (gridDim.y > 1) atomicAdd(o, v);
                    else if (d.accum) *o += v;
                    else *o = v;
                }
            }
        return;
    }
    // ---- epilogue: the gate math of convlstm_bf16_kernel's 16-channel blocks (a wave's 32 columns = 4 gates x 8 channels) -------------
    auto pick = [&](const float (&v)[4], int idx) -> float {
        const float lo = (idx & 1) ? v[1] : v[0], hi = (idx & 1) ? v[3] : v[2];
        return (idx & 2) ? hi : lo;
    };
    float sv[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float rows[4], val[4], g4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) rows[g] = acc[mt][k * 4 + g];
            val[0] = pick(rows, grp);
#pragma unroll
            for (int x = 1; x < 4; ++x) val[x] = __shfl_xor(pick(rows, grp ^ x), x * 8, 64);
#pragma unroll
            for (int g = 0; g < 4; ++g) g4[g] = pick(val, g ^ grp);
            const int r = k * 4 + grp;
            int ti, ay, ax;
            anchor(32 * MT * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half, ti, ay, ax);
            const int m = ((b0 + ti) * H + y0 + ay) * W + x0 + ax;
            const size_t o = (size_t)m * C + ch;
            const float aj = b_tanh(g4[0] + bj), ai = b_sigmoid(g4[1] + bi);
            const float af = b_sigmoid(g4[2] + bf), ao = b_sigmoid(g4[3] + bo);
            const float cn = cpre[mt][k] * af + ai * aj;
            d.cstate_out[o] = cn;
            const float hn = b_tanh(cn) * ao;
            d.hout[o] = hn;
            sv[mt][k] = hn;
            if (d.gates_out) {
                float* gp = d.gates_out + (size_t)m * 4 * C + ch;
                gp[0] = aj; gp[C] = ai; gp[2 * C] = af; gp[3 * C] = ao;
            }
        }
    BF_STAMP(4);
#ifdef PIVP_BF16_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BF_STAMP(5);
#endif
    if (d.ln_part) {                                   // (count, mean, M2) of the block's h tile, two passes, fixed order over the eight waves
        float* red = reinterpret_cast<float*>(lds);
        float s1 = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int k = 0; k < 4; ++k) s1 += sv[mt][k];
        s1 = wave_sum(s1);
        __syncthreads();
        if (lane == 0) red[wave8] = s1;
        __syncthreads();
        if constexpr (!W8) {
            const float cnt = (float)NW * 64.f * 4.f * MT;
            const float ssum = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
            const float mean = ssum / cnt;
            float q = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float dd = sv[mt][k] - mean; q = fmaf(dd, dd, q); }
            q = wave_sum(q);
            if (lane == 0) red[8 + wave8] = q;
            __syncthreads();
            if (tid == 0) {
                float* p = d.ln_part + ((size_t)b0 * d.ln_nparts + (size_t)trem * n_nblk + nblk) * 4;
                const float qs = ((red[8] + red[9]) + (red[10] + red[11])) + ((red[12] + red[13]) + (red[14] + red[15]));
                p[0] = cnt; p[1] = mean; p[2] = qs; p[3] = 0.f;
            }
        } else {
            // two images per tile: the waves wm < NWM / 2 (of every wave column) own image 0, the others image 1 -- a wave's 32 MT anchors lie inside one image
            const int img = (wm * MT) >> 1;
            const float cnt = (float)(NW / 2) * 64.f * 4.f * MT;
            float ssum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) ssum += ((((w % NWM) * MT) >> 1) == img) ? red[w] : 0.f;      // fixed order
            const float mean = ssum / cnt;
            float q = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float dd = sv[mt][k] - mean; q = fmaf(dd, dd, q); }
            q = wave_sum(q);
            if (lane == 0) red[8 + wave8] = q;
            __syncthreads();
            if (lane == 0 && wn == 0 && wm == img * (NWM / 2)) {      // one writer per image
                float* p = d.ln_part + ((size_t)(b0 + img) * d.ln_nparts + (size_t)trem * n_nblk + nblk) * 4;
                float qs = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) qs += ((((w % NWM) * MT) >> 1) == img) ? red[8 + w] : 0.f;
                p[0] = cnt; p[1] = mean; p[2] = qs; p[3] = 0.f;
            }
        }
    }
}

// the divisors of a block's tile decode (tiles, tiles per image, tiles per row) as multiplier / shift pairs
static inline void x6g_decode_divisors(IgemmDesc& dd, int n_tiles, int tpi, int tpr) {
    pivp_fastdiv((unsigned)n_tiles, &dd.fd_mb_mul, &dd.fd_mb_sh);
    pivp_fastdiv((unsigned)tpi, &dd.fd_hw_mul, &dd.fd_hw_sh);
    pivp_fastdiv((unsigned)tpr, &dd.fd_w_mul, &dd.fd_w_sh);
}
template <int NWM, int NWN, int PCS, bool IN_LN, bool W8 = false>
static int launch_x6g_impl(const IgemmDesc& d, const unsigned short* wb, hipStream_t stream, int* ln_nparts) {
    constexpr int THX = TH, lds_bytes = PCS * (W8 ? PATCH_BYTES : PH * RP16);
    static_assert(lds_bytes <= 160 * 1024, "LDS");
    static PerDeviceOnce once;
    if (pivp_ensure_dyn_lds(once, reinterpret_cast<const void*>(&convlstm_x6g_kernel<NWM, NWN, true, PCS, IN_LN, W8>), lds_bytes) != PIVP_OK) return PIVP_ERR_LAUNCH;
    PIVP_CHECK_ARG(d.Hin % THX == 0 && d.Win % (W8 ? 8 : 16) == 0 && (!W8 || d.B % 2 == 0));
    IgemmDesc dd = d;
    const int tpi = (d.Hin / THX) * (d.Win / (W8 ? 8 : 16)), nb = d.C / (8 * NWN);
    const int np = tpi * nb;
    dd.ln_nparts = (d.ln_part && np <= d.ln_cap) ? np : 0;
    if (!dd.ln_nparts) dd.ln_part = nullptr;
    if (ln_nparts) *ln_nparts = dd.ln_nparts;
    const long long wbytes = (long long)lstm_bf16_weight_elems(d.c0 + (d.c1 ? d.c1 : d.C), 4 * d.C) * PCS * 2;
    if (wbytes >= (1LL << 31)) return PIVP_ERR_BADARG;
    x6g_decode_divisors(dd, (d.B / (W8 ? 2 : 1)) * tpi, tpi, d.Win / (W8 ? 8 : 16));
    hipLaunchKernelGGL((convlstm_x6g_kernel<NWM, NWN, true, PCS, IN_LN, W8>), dim3((d.B / (W8 ? 2 : 1)) * tpi * nb), dim3(64 * NWM * NWN), lds_bytes, stream, dd, wb, (int)wbytes, 0);
    return PIVP_LAUNCH_STATUS();
}
// ... on maps 8 pixels wide (an even batch): tiles of two images
template <int NWM, int NWN, int PCS = 3>
static int launch_x6g_w8(const IgemmDesc& d, const unsigned short* wb, hipStream_t stream, int* ln_nparts) {
    PIVP_CHECK_ARG(!d.in_g);
    return launch_x6g_impl<NWM, NWN, PCS, false, true>(d, wb, stream, ln_nparts);
}
template <int NWM, int NWN, int PCS = 3>
static int launch_x6g(const IgemmDesc& d, const unsigned short* wb, hipStream_t stream, int* ln_nparts) {
    if (d.in_g) {       // the x operand's LayerNorm applied while staging (x in one 64-channel group)
        PIVP_CHECK_ARG(d.in_b && d.in_part && d.in_np > 0 && d.c0 <= 64 && d.ld0 == d.c0 && d.in_part != d.ln_part);
        return launch_x6g_impl<NWM, NWN, PCS, true>(d, wb, stream, ln_nparts);
    }
    return launch_x6g_impl<NWM, NWN, PCS, false>(d, wb, stream, ln_nparts);
}

// the plain 5x5 convolution on the same kernel: dd.N = rows of the padded pack, nb = its 64-column blocks, ks = split of the channel groups
template <int PCS, bool W8 = false>
static int launch_x6g_plain(const IgemmDesc& dsc, const unsigned short* wb, hipStream_t stream, int nb, int ks, int ncols) {
    IgemmDesc dd = dsc;
    constexpr int lds_bytes = PCS * (W8 ? PATCH_BYTES : PH * RP16);
    static PerDeviceOnce once;
    if (pivp_ensure_dyn_lds(once, reinterpret_cast<const void*>(&convlstm_x6g_kernel<4, 2, false, PCS, false, W8>), lds_bytes) != PIVP_OK) return PIVP_ERR_LAUNCH;
    PIVP_CHECK_ARG(dd.Win % (W8 ? 8 : 16) == 0 && (!W8 || dd.B % 2 == 0));
    const int tpi = (dd.Hin / TH) * (dd.Win / (W8 ? 8 : 16));
    const long long wbytes = (long long)lstm_bf16_weight_elems(dd.c0 + dd.c1, dd.N) * PCS * 2;
    if (wbytes >= (1LL << 31)) return PIVP_ERR_BADARG;
    x6g_decode_divisors(dd, (dd.B / (W8 ? 2 : 1)) * tpi, tpi, dd.Win / (W8 ? 8 : 16));
    hipLaunchKernelGGL((convlstm_x6g_kernel<4, 2, false, PCS, false, W8>), dim3((dd.B / (W8 ? 2 : 1)) * tpi * nb, ks), dim3(512), lds_bytes, stream, dd, wb, (int)wbytes, ncols);
    return PIVP_LAUNCH_STATUS();
}

}  // namespace pivp
