// ConvLSTM gate convolution with bf16 operands and fp32 accumulation (BASELINE.json config 3, "bf16"): the opt-in
// reduced-precision form of igemm_lstm (igemm_f32.hip).  Reference op: BasicConvLSTMCell.__call__, TM:234-276.
//
// With v_mfma_f32_32x32x16_bf16 a 32x32 tile of a K = 32 chunk costs 64 MFMA cycles instead of the 1024 of the fp32
// instruction, so the gather scheme of igemm_f32.hip (a fresh A tile per tap) would be bound by its staging, not by the
// matrix pipe.  This kernel therefore keeps the block's INPUT PATCH resident: a block owns 128 anchors of one image
// (8 x 16 pixels; on 8-wide maps 8 x 8 pixels of two images) and 4 gates x NCH channels.  For every group of 64 input
// channels of concat(x, h) it stages the patch with its 2-pixel halo ONCE (fp32 NHWC in HBM -> bf16 in LDS, pixel pitch
// 144 B so that the 16-lane phases of a ds_read_b128 hit distinct banks; out-of-image pixels and channels past Cin are
// the hardware zeros of an out-of-range buffer load) and runs the 25 taps against it: a tap only shifts the LDS address
// of the A fragment.  The weights ([group][tap][4C][64] bf16, packed once per rollout by pack_lstm_bf16) stream through a
// 4-slot LDS ring filled by global_load_lds_dwordx4 three taps ahead of their use: no VGPRs, no ds_writes, and the
// loads stay in flight across the per-tap barrier (raw s_barrier + counted vmcnt).  The DMAs are issued by four LOADER
// waves (waves 4..7 of the 8-wave block) that do nothing else in the tap loop; waves 0..3 multiply (2 x 2 over the
// 128 x 4 NCH block tile) and issue no VMEM instruction there.  A ring slot is lane-linear (the DMA
// writes base + lane * 16), so bank conflicts of the B fragment reads are removed by an XOR swizzle of the 16-B pieces
// of a 128-B weight row, applied to the per-lane SOURCE address of the DMA and to the read address alike.
// Accumulators, gate math, cell state, h and the LayerNorm partial stay fp32; the epilogue is that of igemm_f32.hip.
// The same kernel with a plain epilogue (LSTM = false) is a general 5x5 stride-1 convolution: the ConvLSTM data gradient.
#pragma once
#include "convlstm_bf16_common.h"

namespace pivp {

// LSTM = true: the ConvLSTM cell (block columns = 4 gates x NCH channels, gate epilogue).  LSTM = false: a plain 5x5 stride-1 "same"
// convolution out[m][n] (+)= sum x[m + tap][k] w[tap][k][n] with block columns = 4 NCH consecutive n (the ConvLSTM DATA gradient: x = dG,
// w = the flipped transposed weights); gridDim.y splits the channel groups, partial sums then meet in `out` by atomic adds.
// PL = 2: split mode.  Every fp32 operand travels as TWO bf16 numbers, hi = bf16(v) and lo = bf16(v - hi) (two patch planes, two weight
// planes per ring slot), and a product a * b is formed as a_lo * b_hi + a_hi * b_lo + a_hi * b_hi on three MFMAs (each exact in fp32):
// 16 bits of product mantissa instead of 8, 3e-5 instead of 2e-2 per-pixel on the config 1 rollout (scripts/split_bf16_study.py).
// (Round 4's fp16-piece form of this kernel -- the fp16x3 mode's layers on 8-wide maps -- went in round 5: the L2-direct kernel's two-image tiles
// serve them at the same speed, convlstm_l2direct.h.)
template <int NCH, bool LSTM, int PL = 1>
__global__ __launch_bounds__(512, 1) void convlstm_bf16_kernel(const IgemmDesc d, const unsigned short* __restrict__ wb, int tw, int ncols) {
    constexpr int BN = 4 * NCH;                 // block columns: [gate][channel]
    constexpr int PLANE = BN * 128;             // one weight plane of a ring slot: BN rows x 64 bf16
    constexpr int SLOT = PL * PLANE;            // bytes of one ring slot
    // Split mode with 32-channel blocks: two patch planes (92 KB) leave room for TWO 32 KB ring slots only, so the schedule changes: the
    // loaders bring tap it + 1 in during tap it (one tap of lookahead), and the block barrier sits at the END of a tap (LATE).
    constexpr bool LATE = PL == 2 && NCH == 32;
    constexpr int PB = patch_plane_bytes<PL>();     // bytes of one patch plane
    static_assert(PL == 1 || PL == 2, "one plane, or the hi / lo pair (three pieces: convlstm_x6g_kernel)");
    constexpr int DEP = ring_depth<NCH, PL>();      // taps of weight prefetch
    constexpr int NSL = DEP + 1;                    // ring slots
    static_assert(PL * PB + NSL * PL * BN * 128 <= 160 * 1024 && DEP >= 1, "the ring must fit beside the patch");
    constexpr int G = BN / 32;                  // global_load_lds per loader thread and tap
    constexpr int TPW = NCH / 16;               // MFMA column tiles per wave
    constexpr int CPW = NCH / 2;                // channels per wave (all 4 gates of a channel stay in one wave)
    constexpr int GPT = 32 / CPW;               // gates per MFMA tile
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // patch | ring
    unsigned char* const patch = lds;                  // patch plane(s) | ring
    const int tid = threadIdx.x, lane = tid & 63;
    // waves 0..3 multiply (2 x 2 over the 128 x BN block tile), waves 4..7 only feed the weight ring: a global_load_lds costs
    // 60-180 cycles of its wave's issue time, which in a multiplying wave is time the matrix pipe idles (4 per tap: a quarter
    // of the tap); issued by a second wave of the same SIMD they overlap the MFMAs.
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave8 >= 4;
    const int wave = wave8 & 3;
    const int wm = wave & 1, wn = wave >> 1;
    const int half = lane >> 5, l31 = lane & 31;
    const int C = d.C, N = d.N;                       // (plain conv: N = rows of the padded weight pack, a multiple of BN)
    const int n_nblk = LSTM ? C / NCH : N / BN;
    const int H = d.Hin, W = d.Win;
    // tile geometry: tw = 16: one image, 8 x 16 anchors; tw = 8: two images, 8 x 8 anchors each
    const int ti_n = tw == 16 ? 1 : 2;
    const int PW = tw + 4;
    const int RP = tw == 16 ? RP16 : RP8;             // patch row pitch (bytes)
    const int npix = ti_n * PH * PW;
    const int tpr = W / tw, tpi = (H / TH) * tpr;          // tiles per row / per image
    const int n_tiles = (d.B / ti_n) * tpi;
    int lid = blockIdx.x;
    if ((gridDim.x & 7) == 0) lid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-aware, column-block major
    // (divisions by run-time values through the launcher's multipliers, pivp_fastdiv: a block's time in front of its first load is its instruction count --
    // 600 instructions here before round 6, most of them the division sequences of this decode and of the patch pixels below)
    const int nblk = pivp_fdiv(lid, d.fd_mb_mul, d.fd_mb_sh), tile = lid - nblk * n_tiles;
    const int timg = pivp_fdiv(tile, d.fd_hw_mul, d.fd_hw_sh);
    const int b0 = timg * ti_n, trem = tile - timg * tpi;
    const int trow = pivp_fdiv(trem, d.fd_w_mul, d.fd_w_sh);
    const int y0 = trow * TH, x0 = (trem - trow * tpr) * tw;
    BF_STAMP(0);
    const int c0 = d.c0, ld0 = d.ld0, ld1 = d.ld1;
    const int cin = c0 + d.c1;
    const int ncg_all = (cin + 63) >> 6;
    const int cgbase = (int)blockIdx.y * ncg_all / (int)gridDim.y;                 // this block's channel groups: [cgbase, cgbase + ncg)
    const int ncg = ((int)blockIdx.y + 1) * ncg_all / (int)gridDim.y - cgbase;
    const int nchunks = 25 * ncg;

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x0), 0, d.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.c1 ? d.x1 : d.x0), 0, d.c1 ? d.bytes1 : d.bytes0, 0x00020000);
    constexpr unsigned OOB = 0xC0000000u;

    // ---- patch staging (all 8 waves): thread = (pixel (tid >> 3) + 64 j, 8-channel piece tid & 7), j < 5 -------------------
    const int cpiece = tid & 7;
    int a_pix[NPJ];                                    // global pixel index, or -1 outside the image / past the patch
    int a_lds[NPJ];                                    // the pixel's byte offset in the patch (pixels past the patch: row 0's padding)
#pragma unroll
    for (int j = 0; j < NPJ; ++j) {
        const int p = (tid >> 3) + 64 * j;
        const int ti = tw == 16 ? p / (PH * 20) : p / (PH * 12), pr = p - ti * (PH * PW);      // PW is 20 or 12: constant divisors
        const int py = tw == 16 ? pr / 20 : pr / 12, px = pr - py * PW;
        const int iy = y0 - 2 + py, ix = x0 - 2 + px;
        const bool ok = p < npix && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        a_pix[j] = ok ? ((b0 + ti) * H + iy) * W + ix : -1;
        a_lds[j] = p < npix ? (ti * PH + py) * RP + px * PP : PW * PP;
    }
    f32x4 plo[NPJ], phi[NPJ];                          // a patch in flight (live only between the two halves of a staging)
    // The source of a thread's 8-channel piece (x or h) differs between the lanes of a wave only in the ONE 64-channel group that straddles c0.
    // Written as a per-lane choice of descriptor (rounds 2-5) every load became a waterfall loop -- readfirstlane x 4, two 64-bit compares, exec
    // juggling: 14 instructions per load, 42-107 such loops per kernel.  Now the choice is block-uniform: a group inside one source loads from
    // it; the straddling group loads from both with the other source's lanes out of range (the hardware's zeros) and ORs the two.
    auto patch_load = [&](int cg) {
        const int gch = (cgbase + cg) * 64;               // the group's first channel: block-uniform
        const int ch = gch + cpiece * 8;                  // first of this thread's 8 channels of concat(x, h)
        const bool s0 = ch < c0, s1 = !s0 && ch < cin;
        const bool use0 = gch < c0, use1 = gch + 64 > c0 && gch < cin;
        auto ld2 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off, f32x4& lo, f32x4& hi) {
            lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 16, 0));
        };
#pragma unroll
        for (int j = 0; j < NPJ; ++j) {
            const unsigned off0 = (a_pix[j] >= 0 && s0) ? (unsigned)((a_pix[j] * ld0 + ch) * 4) : OOB;
            const unsigned off1 = (a_pix[j] >= 0 && s1) ? (unsigned)((a_pix[j] * ld1 + ch - c0) * 4) : OOB;
            if (use0 && use1) {
                f32x4 blo, bhi;
                ld2(rs0, off0, plo[j], phi[j]);
                ld2(rs1, off1, blo, bhi);
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                plo[j] = __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, plo[j]) | __builtin_bit_cast(u32x4, blo));
                phi[j] = __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, phi[j]) | __builtin_bit_cast(u32x4, bhi));
            } else if (use0) {
                ld2(rs0, off0, plo[j], phi[j]);
            } else {
                ld2(rs1, off1, plo[j], phi[j]);           // (a group past cin: every lane out of range)
            }
        }
    };
    auto patch_store = [&]() {
#pragma unroll
        for (int j = 0; j < NPJ; ++j) {
            // unconditional (pixels past npix write their zeros into the padding behind row 0, which nobody reads): a predicated write
            // leaves the loads "pending" on the skipped path for hipcc's wait-count pass, which then drains vmcnt inside the tap loop
            uint4 v;
            v.x = pack2(plo[j][0], plo[j][1]); v.y = pack2(plo[j][2], plo[j][3]);
            v.z = pack2(phi[j][0], phi[j][1]); v.w = pack2(phi[j][2], phi[j][3]);
            *reinterpret_cast<uint4*>(patch + a_lds[j] + cpiece * 16) = v;
            if constexpr (PL >= 2) {                   // second plane: bf16(v - hi); hi as a float is its 16 bits shifted up
                float r[8] = {plo[j][0], plo[j][1], plo[j][2], plo[j][3], phi[j][0], phi[j][1], phi[j][2], phi[j][3]};
                auto rest = [&](unsigned p2, int i) {      // r[i], r[i + 1] become the remainders (exact in fp32); returns them as bf16
                    r[i] -= __builtin_bit_cast(float, p2 << 16); r[i + 1] -= __builtin_bit_cast(float, p2 & 0xffff0000u);
                    return pack2(r[i], r[i + 1]);
                };
                uint4 l;
                l.x = rest(v.x, 0); l.y = rest(v.y, 2); l.z = rest(v.z, 4); l.w = rest(v.w, 6);
                *reinterpret_cast<uint4*>(patch + PB + a_lds[j] + cpiece * 16) = l;
            }
        }
    };

    // Every block walks the 25 taps of a channel group in its own rotation (tap0, tap0 + 1, ... mod 25): blocks that run in
    // step would otherwise all pull the same 16 KB of weights out of the same few L2 channels at the same time.
    const int tap0 = (lid * 7) % 25;

    // =========================================================================================================================
    // loader waves: DMA j of a tap writes ring bytes [(j * 256 + lt) * 16, +16): row (j * 32 + lt / 8), piece lt % 8, which holds
    // SOURCE piece (lt % 8) ^ (row % 8) of that row (lt = thread index within the four loader waves)
    // =========================================================================================================================
    if (loader) {
        const int lt = tid - 256;
        unsigned char* const ring = lds + PL * PB;
        const unsigned char* wsrc[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int row = j * 32 + (lt >> 3), g = row / NCH, cl = row - g * NCH;
            const int piece = (lt & 7) ^ ring_swizzle<NCH, LSTM>(row);
            const int grow = LSTM ? g * C + nblk * NCH + cl : nblk * BN + row;
            wsrc[j] = reinterpret_cast<const unsigned char*>(wb) + (size_t)grow * 128 + piece * 16;
        }
        const size_t wstep = (size_t)PL * N * 128;     // bytes between consecutive (group, tap) weight tiles ([PL][N][64] bf16 each)
        int i_slot = 0, i_tap = tap0, i_cg = 0;        // the ring slot of the next tap to issue, and which tap that is
        auto issue_weights = [&]() {
            const int slot = i_slot;
            i_slot = i_slot + 1 == NSL ? 0 : i_slot + 1;
            const size_t goff = (size_t)((cgbase + i_cg) * 25 + i_tap) * wstep;
            i_tap = i_tap == 24 ? 0 : i_tap + 1;
            i_cg += i_tap == tap0 ? 1 : 0;
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    unsigned char* dst = ring + slot * SLOT + pl * PLANE + (j * 256 + wave * 64) * 16;   // wave-uniform; the DMA adds lane * 16
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[j] + goff + (size_t)pl * N * 128),
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                }
        };
        if constexpr (LATE) {
            issue_weights();                                   // tap 0
            patch_load(0);
            patch_store();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // tap 0 and the patch are published
            int tap = tap0, cg = 0;
            for (int it = 0; it < nchunks; ++it) {
                if (it + 1 < nchunks) issue_weights();         // tap it + 1 into the slot tap it - 1 left at the last barrier
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                  // end of tap it: tap it + 1 is published
                tap = tap == 24 ? 0 : tap + 1;
                if (tap == tap0 && ++cg < ncg) {               // next 64 input channels: all 8 waves restage the patch
                    patch_load(cg);
                    patch_store();
                    __syncthreads();
                }
            }
            return;
        }
        constexpr int DEP0 = DEP < 3 ? DEP : 3;      // taps requested in front of the patch (the patch's loads return behind them)
#pragma unroll
        for (int i = 0; i < DEP0; ++i)
            if (i < nchunks) issue_weights();
        patch_load(0);
        patch_store();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // taps 0..2 and this thread's part of the patch are in LDS
#pragma unroll
        for (int i = DEP0; i < DEP; ++i)             // the rest of the ring: in flight across the barrier
            if (i < nchunks) issue_weights();
        __builtin_amdgcn_s_barrier();
        int tap = tap0, cg = 0;
        for (int it = 0; it < nchunks; ++it) {
            // the multiplying waves' mid-tap barrier publishes tap it + 1: this wave's share of it must have landed.  Issued so far:
            // taps up to it + DEP - 1; the `newer` ones behind tap it + 1 may stay in flight (G * PL DMAs per thread and tap).
            {
                int newer = nchunks - it - 2;
                newer = newer < 0 ? 0 : newer > DEP - 2 ? DEP - 2 : newer;
                constexpr int Q = G * PL;
                if (newer <= 0) wait_vmcnt<0>();
                else if (newer == 1) wait_vmcnt<Q>();
                else if (newer == 2) wait_vmcnt<2 * Q>();
                else if (newer == 3) wait_vmcnt<3 * Q>();
                else if (newer == 4) wait_vmcnt<4 * Q>();
                else wait_vmcnt<5 * Q>();
                static_assert(DEP - 2 <= 5 && 5 * Q <= 63, "vmcnt immediates");
            }
            __builtin_amdgcn_s_barrier();
            // every multiplying wave is past tap it - 1: its ring slot takes tap it + DEP
            if (it + DEP < nchunks) issue_weights();
            tap = tap == 24 ? 0 : tap + 1;
            if (tap == tap0 && ++cg < ncg) {           // next 64 input channels: all 8 waves restage the patch
                __syncthreads();
                patch_load(cg);
                patch_store();
                __syncthreads();
            }
        }
        return;                                        // (the last iteration drained this wave's DMAs)
    }

    // =========================================================================================================================
    // multiplying waves
    // =========================================================================================================================
    f32x16 acc[2][TPW];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][t][r] = 0.f;

    // fragment addresses (bytes).  A: row l31 of M tile mt = anchor 64 wm + 32 mt + l31; k piece `half` of the k-step
    int a_off[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int i = 64 * wm + 32 * mt + l31;
        const int ti = tw == 16 ? 0 : i >> 6, ay = tw == 16 ? i >> 4 : (i >> 3) & 7, ax = tw == 16 ? i & 15 : i & 7;
        a_off[mt] = (ti * PH + ay) * RP + ax * PP + half * 16;
    }
    // B: MFMA column l31 of tile t = gate t * GPT + l31 / CPW, channel wn * CPW + l31 % CPW; ring row = gate * NCH + channel
    int b_row[TPW], b_sw[4];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
        b_row[t] = (LSTM ? (t * GPT + l31 / CPW) * NCH + wn * CPW + (l31 % CPW) : (wn * TPW + t) * 32 + l31) * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)      // (the tiles of a wave are 32 or 64 ring rows apart: one swizzle value serves them all)
        b_sw[ks] = ((2 * ks + half) ^ ring_swizzle<NCH, LSTM>(b_row[0] >> 7)) * 16;

    // LDS reads go through inline asm: hipcc knows that an LDS-DMA writes LDS and puts s_waitcnt vmcnt(0) in front of every
    // ds_read it can see.  The waits below are explicit instead.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    bf16x8 fa[2][2], fb[2][TPW];                       // [register set][tile]
    bf16x8 fal[2][2], fbl[2][TPW];                     // ... and their second planes (split modes)
    auto wait_frags = [&](auto SET) {
        constexpr int st = decltype(SET)::value;
        if constexpr (PL == 2 && TPW == 2) wait_lgkm(fa[st][0], fa[st][1], fb[st][0], fb[st][1], fal[st][0], fal[st][1], fbl[st][0], fbl[st][1]);
        else if constexpr (PL == 2) wait_lgkm(fa[st][0], fa[st][1], fb[st][0], fal[st][0], fal[st][1], fbl[st][0]);
        else if constexpr (TPW == 2) wait_lgkm(fa[st][0], fa[st][1], fb[st][0], fb[st][1]);
        else wait_lgkm(fa[st][0], fa[st][1], fb[st][0]);
    };
    auto read_frags = [&](auto SET, auto KS, int tp, int slot) {   // fragments of k-step KS of tap tp (weights in ring slot `slot`)
        constexpr int st = decltype(SET)::value, ks = decltype(KS)::value;
        const int ty = tp / 5, tx = tp - ty * 5;
        const unsigned ab = lds0 + ty * RP + tx * PP;
        const unsigned bb = lds0 + PL * PB + slot * SLOT + b_sw[ks];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) fa[st][mt] = lds_read_b128<ks * 32>(ab + a_off[mt]);
#pragma unroll
        for (int t = 0; t < TPW; ++t) fb[st][t] = lds_read_b128<0>(bb + b_row[t]);
        if constexpr (PL >= 2) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) fal[st][mt] = lds_read_b128<ks * 32>(ab + PB + a_off[mt]);
#pragma unroll
            for (int t = 0; t < TPW; ++t) fbl[st][t] = lds_read_b128<0>(bb + PLANE + b_row[t]);
        }
    };
    auto mfmas = [&](auto SET) {
        constexpr int st = decltype(SET)::value;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                if constexpr (PL == 2) {               // the two cross terms first, the leading term last
                    acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[st][mt], fb[st][t], acc[mt][t], 0, 0, 0);
                    acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][mt], fbl[st][t], acc[mt][t], 0, 0, 0);
                }
                acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][mt], fb[st][t], acc[mt][t], 0, 0, 0);
            }
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;

    // ---- prologue: this thread's part of the patch, the epilogue's operands ---------------------------------------------------
    // bias and c_{t-1} are requested here: read in the epilogue they cost one exposed HBM round trip per accumulator row
    // (16-32 in a row), more than the whole tap loop.  Lane (grp, channel) owns accumulator rows r with r % GPT == grp.
    patch_load(0);
    const int chl = wn * CPW + (l31 % CPW);
    const int ch = nblk * NCH + chl;
    const int grp = l31 / CPW;
    constexpr int OWN = 16 / GPT;
    float bj = 0.f, bi = 0.f, bf = 0.f, bo = 0.f;
    float cpre[2][OWN];
    if constexpr (LSTM) {
        bj = d.bias[ch]; bi = d.bias[C + ch]; bf = d.bias[2 * C + ch] + 1.0f; bo = d.bias[3 * C + ch];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int k = 0; k < OWN; ++k) {
                const int r = k * GPT + grp;
                const int i = 64 * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int ti = tw == 16 ? 0 : i >> 6, ay = tw == 16 ? i >> 4 : (i >> 3) & 7, ax = tw == 16 ? i & 15 : i & 7;
                const int m = ((b0 + ti) * H + y0 + ay) * W + x0 + ax;
                cpre[mt][k] = d.cstate_in[(size_t)m * C + ch];
            }
    }
    patch_store();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    BF_STAMP(1);
    __builtin_amdgcn_s_barrier();                      // patch and taps 0..2 are in LDS
    BF_STAMP(2);
    if constexpr (LATE) {
        // end-of-tap barrier schedule: the first fragments of a tap are requested right behind the barrier that published it
        int tap = tap0, cg = 0;
        for (int it = 0; it < nchunks; ++it) {
            const int slot = it & 1;
            read_frags(S0{}, K0{}, tap, slot);
            wait_frags(S0{}); read_frags(S1{}, K1{}, tap, slot); mfmas(S0{});
            __builtin_amdgcn_sched_barrier(0);
            wait_frags(S1{}); read_frags(S0{}, K2{}, tap, slot); mfmas(S1{});
            __builtin_amdgcn_sched_barrier(0);
            wait_frags(S0{}); read_frags(S1{}, K3{}, tap, slot); mfmas(S0{});
            __builtin_amdgcn_sched_barrier(0);
            wait_frags(S1{}); mfmas(S1{});
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            tap = tap == 24 ? 0 : tap + 1;
            if (tap == tap0 && ++cg < ncg) {
                patch_load(cg);
                patch_store();
                __syncthreads();
            }
        }
    } else {
    read_frags(S0{}, K0{}, tap0, 0);

    // One tap = 4 k-steps of 16 channels; fragments of the next k-step are requested before the MFMAs of the current one.  The
    // barrier that publishes the NEXT tap's weights sits in the middle of the tap (its skew hides behind queued MFMAs), so the
    // first fragments of the next tap can be requested right after the last k-step.  No VMEM instruction in this loop.
    int tap = tap0, cg = 0, slot = 0;
    for (int it = 0; it < nchunks; ++it) {
        const int nslot = slot + 1 == NSL ? 0 : slot + 1;
        wait_frags(S0{}); read_frags(S1{}, K1{}, tap, slot); mfmas(S0{});
        __builtin_amdgcn_sched_barrier(0);
        wait_frags(S1{}); read_frags(S0{}, K2{}, tap, slot); mfmas(S1{});
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        wait_frags(S0{}); read_frags(S1{}, K3{}, tap, slot); mfmas(S0{});
        __builtin_amdgcn_sched_barrier(0);
        tap = tap == 24 ? 0 : tap + 1;
        const bool regroup = tap == tap0;
        wait_frags(S1{});
        if (!regroup) read_frags(S0{}, K0{}, tap, nslot);
        mfmas(S1{});
        if (regroup && ++cg < ncg) {                   // next 64 input channels: all 8 waves restage the patch
            __syncthreads();                           // every wave is done with the old patch
            patch_load(cg);
            patch_store();
            __syncthreads();
            read_frags(S0{}, K0{}, tap, nslot);
        }
        slot = nslot;
        __builtin_amdgcn_sched_barrier(0);
    }
    }   // !LATE
    if constexpr (!LSTM) {
        // ---- plain epilogue: accumulator row = anchor, column = output channel; 32 lanes write 128 contiguous bytes ----------
        // The epilogue hook's second tensor (d.ep_src) is requested for ALL of the lane's outputs first and met below: read next to each store
        // it cost one exposed round trip per accumulator row (the data gradients ran 9-11 us longer per launch with it, more than the pass it
        // replaces).  Neutral elements where the hook does not apply: 1 for the mask, 0 for the sum.
        const long long out_bytes = (long long)d.B * H * W * d.ldo * 4, ep_bytes = d.ep_mode ? (long long)d.B * H * W * d.ep_ld * 4 : 0;
        if (out_bytes < (1LL << 31) && ep_bytes < (1LL << 31)) {
            // Round 6: as the fp32 data gradient's epilogue (igemm_f32.hip): the lane-dependent part of an element's address is ONE 32-bit offset of a
            // buffer descriptor (wave tile base, + 4 * half rows, + column), the accumulator register's row a scalar offset, columns past the pack's
            // real count an out-of-range offset the hardware drops -- instead of a 64-bit address, a tile decode and a predicate per element
            // (~25 instructions each, 32-64 elements per lane, run by four of the block's eight waves while the others idle).
            const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, (int)out_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rse = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.ep_mode ? d.ep_src : d.out), 0, d.ep_mode ? (int)ep_bytes : 0, 0x00020000);
            const int basem = ((b0 + (tw == 16 ? 0 : wm)) * H + y0 + (tw == 16 ? 4 * wm : 0)) * W + x0 + 4 * half;      // anchor of accumulator row 0 of M tile 0
            constexpr unsigned OOBX = 0xC0000000u;
            unsigned vo[TPW], ve[TPW];
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int col = nblk * BN + (wn * TPW + t) * 32 + l31;
                vo[t] = col < ncols ? (unsigned)((basem * d.ldo + col) * 4) : OOBX;
                ve[t] = (col < ncols && col < d.ep_cols) ? (unsigned)((basem * d.ep_ld + col) * 4) : OOBX;
            }
            auto srow = [&](int mt, int r) -> int {     // anchors between accumulator row r of M tile mt and row 0 of tile 0 (block-uniform)
                return tw == 16 ? (2 * mt + (r >> 3)) * W + (r & 3) + 8 * ((r >> 2) & 1) : (4 * mt + (r >> 2)) * W + (r & 3);
            };
            float ev[2][TPW][16];
            if (d.ep_mode) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
#pragma unroll
                        for (int t = 0; t < TPW; ++t)
                            ev[mt][t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rse, ve[t], srow(mt, r) * d.ep_ld * 4, 0));
            }
            const int mode = d.ep_mode, split = gridDim.y > 1, accum = d.accum;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int so = srow(mt, r) * d.ldo * 4;
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        float v = acc[mt][t][r];
                        if (mode == 1) v = (ve[t] != OOBX ? ev[mt][t][r] > 0.f : true) ? v : 0.f;      // (columns the hook does not cover: unmasked)
                        else if (mode == 2) v += ev[mt][t][r];                                       // (out-of-range loads returned 0)
                        if (split) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rso, vo[t], so, 0);
                        else {
                            if (accum) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rso, vo[t], so, 0));
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rso, vo[t], so, 0);
                        }
                    }
                }
            return;
        }
        float ev[2][TPW][16];
        if (d.ep_mode) {                                // block-uniform
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = 64 * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int ti = tw == 16 ? 0 : i >> 6, ay = tw == 16 ? i >> 4 : (i >> 3) & 7, ax = tw == 16 ? i & 15 : i & 7;
                    const size_t m = (size_t)(((b0 + ti) * H + y0 + ay) * W + x0 + ax);
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        const int col = nblk * BN + (wn * TPW + t) * 32 + l31;
                        ev[mt][t][r] = (col < ncols && col < d.ep_cols) ? d.ep_src[m * d.ep_ld + col] : (d.ep_mode == 1 ? 1.f : 0.f);
                    }
                }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = 64 * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int ti = tw == 16 ? 0 : i >> 6, ay = tw == 16 ? i >> 4 : (i >> 3) & 7, ax = tw == 16 ? i & 15 : i & 7;
                const size_t m = (size_t)(((b0 + ti) * H + y0 + ay) * W + x0 + ax);
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const int col = nblk * BN + (wn * TPW + t) * 32 + l31;
                    if (col < ncols) {                  // the pack's rows past the real column count are zero padding
                        float* o = d.out + m * d.ldo + col;
                        float v = acc[mt][t][r];
                        if (d.ep_mode == 1) v = ev[mt][t][r] > 0.f ? v : 0.f;      // (unsplit grids only: the launcher clears ep_mode otherwise)
                        else if (d.ep_mode == 2) v += ev[mt][t][r];
                        if (gridDim.y > 1) atomicAdd(o, v);
                        else if (d.accum) *o += v;
                        else *o = v;
                    }
                }
            }
        return;
    }
    BF_STAMP(3);
    // ---- epilogue: gates, state update, optional gate activations and LayerNorm partial ----------------------------------
    // Accumulator row r of a lane is one anchor; its column is (gate t * GPT + grp, channel): the 4 gates of an (anchor, channel)
    // sit in the GPT lanes lane ^ (x * CPW) and the TPW tiles.  Lane grp takes rows r = k * GPT + grp: it keeps its own gate of
    // that row and receives the others from its partners in GPT - 1 xor-shuffles per tile, each partner sending the row its
    // receiver owns.  Every lane then updates one (anchor, channel) per k: no idle lanes, 2 (NCH 32) or 3 (NCH 16) shuffles per
    // cell instead of 8 or 16.  Register arrays are only indexed statically; per-lane choices are select chains.
    auto pick = [&](const float (&v)[GPT], int idx) -> float {
        if constexpr (GPT == 2) {
            return idx ? v[1] : v[0];
        } else {
            const float lo = (idx & 1) ? v[1] : v[0], hi = (idx & 1) ? v[3] : v[2];
            return (idx & 2) ? hi : lo;
        }
    };
    float sv[2][OWN];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int k = 0; k < OWN; ++k) {
            float g4[4];
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                float rows[GPT], val[GPT];
#pragma unroll
                for (int g = 0; g < GPT; ++g) rows[g] = acc[mt][t][k * GPT + g];
                val[0] = pick(rows, grp);
#pragma unroll
                for (int x = 1; x < GPT; ++x) val[x] = __shfl_xor(pick(rows, grp ^ x), x * CPW, 64);
#pragma unroll
                for (int g = 0; g < GPT; ++g) g4[t * GPT + g] = pick(val, g ^ grp);
            }
            const int r = k * GPT + grp;
            const int i = 64 * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half;       // anchor within the block
            const int ti = tw == 16 ? 0 : i >> 6, ay = tw == 16 ? i >> 4 : (i >> 3) & 7, ax = tw == 16 ? i & 15 : i & 7;
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the block's stores have left
    BF_STAMP(5);
#endif
    if (d.ln_part) {
        // (count, mean, M2) of the h values of each image of the tile; with two images wave pair wm owns image wm.
        float* red = reinterpret_cast<float*>(lds);
        float s1 = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int k = 0; k < OWN; ++k) s1 += sv[mt][k];
        s1 = wave_sum(s1);
        const float c1 = 64.f * 2 * OWN;
        __syncthreads();
        if (lane == 0) { red[wave] = s1; red[4 + wave] = c1; }
        __syncthreads();
        float cnt, mean;
        if (ti_n == 1) {
            cnt = (red[4] + red[5]) + (red[6] + red[7]);
            mean = ((red[0] + red[1]) + (red[2] + red[3])) / cnt;
        } else {
            cnt = red[4 + wm] + red[6 + wm];
            mean = (red[wm] + red[2 + wm]) / cnt;
        }
        float q = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int k = 0; k < OWN; ++k) { const float dd = sv[mt][k] - mean; q = fmaf(dd, dd, q); }
        q = wave_sum(q);
        if (lane == 0) red[8 + wave] = q;
        __syncthreads();
        if (ti_n == 1) {
            if (tid == 0) {
                float* p = d.ln_part + ((size_t)b0 * d.ln_nparts + (size_t)trem * n_nblk + nblk) * 4;
                p[0] = cnt; p[1] = mean; p[2] = (red[8] + red[9]) + (red[10] + red[11]); p[3] = 0.f;
            }
        } else if (lane == 0 && wn == 0) {
            float* p = d.ln_part + ((size_t)(b0 + wm) * d.ln_nparts + (size_t)trem * n_nblk + nblk) * 4;
            p[0] = cnt; p[1] = mean; p[2] = red[8 + wm] + red[10 + wm]; p[3] = 0.f;
        }
    }
}

template <int NCH, bool LSTM, int PL = 1>
static int launch_bf16(const IgemmDesc& d, const unsigned short* wb, hipStream_t stream, int* ln_nparts, int nb, int ksplit, int ncols) {
    constexpr int lds_bytes = PL * patch_plane_bytes<PL>() + (ring_depth<NCH, PL>() + 1) * PL * 4 * NCH * 128;
    static PerDeviceOnce once;
    if (pivp_ensure_dyn_lds(once, reinterpret_cast<const void*>(&convlstm_bf16_kernel<NCH, LSTM, PL>), lds_bytes) != PIVP_OK) return PIVP_ERR_LAUNCH;
    IgemmDesc dd = d;
    const int tw = d.Win % 16 == 0 ? 16 : 8, ti_n = tw == 16 ? 1 : 2;
    const int tpi = (d.Hin / TH) * (d.Win / tw);
    const int np = tpi * nb;
    dd.ln_nparts = (LSTM && d.ln_part && np <= d.ln_cap) ? np : 0;
    if (!dd.ln_nparts) dd.ln_part = nullptr;
    if (ln_nparts) *ln_nparts = dd.ln_nparts;
    const int blocks = (d.B / ti_n) * tpi * nb;
    pivp_fastdiv((unsigned)((d.B / ti_n) * tpi), &dd.fd_mb_mul, &dd.fd_mb_sh);      // tiles, tiles per image, tiles per row: the block's decode
    pivp_fastdiv((unsigned)tpi, &dd.fd_hw_mul, &dd.fd_hw_sh);
    pivp_fastdiv((unsigned)(d.Win / tw), &dd.fd_w_mul, &dd.fd_w_sh);
    hipLaunchKernelGGL((convlstm_bf16_kernel<NCH, LSTM, PL>), dim3(blocks, ksplit), dim3(512), lds_bytes, stream, dd, wb, tw, ncols);
    return PIVP_LAUNCH_STATUS();
}

}  // namespace pivp
