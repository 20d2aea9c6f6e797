// Transposed 3x3 stride-2 convolution (pad 1, outsize = 2 x in: enc4 / enc5 / enc6, TM:505-507, and the data gradient of the 3x3 stride-2
// convs), all four output parities in one block.
//
// igemm_small.hip runs the four sub-pixel phases (1 / 2 / 2 / 4 taps) as four grids of 32-anchor tiles: 16,384 blocks of a few
// microseconds each for enc6, every one re-gathering its input rows, 40 % of the fp32 MFMA rate.  Here a block owns 8 x 16 INPUT
// pixels of one image and 32 output columns, for ALL parities: the input patch with its one-pixel right / bottom halo (9 x 17 pixels x
// 32 channels, zero outside the image = the hardware's out-of-range load result) is staged once per 32-channel chunk, the 9 taps run
// against it -- tap (ky, kx) reads the patch shifted by (ky == 0, kx == 0) and accumulates into parity (ky != 1, kx != 1) -- and the
// wave's four 32 x 32 accumulator tiles become a 2 x 2 output pixel block per input pixel.  Weights ([tap][Cin / 32][N][32]: nine 4 KB
// tiles per channel chunk) and the patch travel global -> register -> LDS one channel chunk ahead, so the 9 taps run without a barrier.  Epilogue: bias, ReLU, optional accumulate, optional LayerNorm
// partial of the block's 16,384 outputs (the norm behind enc6).  Results equal igemm_small's up to fp32 summation order.
#include <type_traits>

#include "pivp_kernels.h"
#include "skinny_linear.h"

#ifdef PIVP_DT_STAMPS   // per-block phase stamps (scripts/deconv_stamps.py): [block][entry, statistics merged, first chunk staged, loop done, stores done]
__device__ long long pivp_dt_stamps[2048 * 8];
#define DT_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 2048) pivp_dt_stamps[blockIdx.x * 8 + (i)] = (long long)wall_clock64(); } while (0)
#else
#define DT_STAMP(i)
#endif

namespace pivp {

namespace {
constexpr int DP = 36;                 // LDS row pitch (floats): conflict-free ds_read_b128
constexpr int PR = 9, PC = 17;         // patch rows / columns (8 x 16 anchors + halo)
constexpr int A_ROWS = 160;             // patch pixel rows allocated: the 5 x 32 staging slots (153 used)
constexpr int A_FL = A_ROWS * DP;      // 5,760 floats
constexpr int B_FL = 32 * DP;          // one weight tile [32 columns][36]
// bf16 form (precision mode bf16): the same images with 32 bf16 + 16 B of padding per row
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int HP = 80;                 // row pitch (bytes)
constexpr int A_HB = A_ROWS * HP, B_HB = 32 * HP;
__device__ __forceinline__ unsigned dpack2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
}

// PREC 1 (bf16): operands rounded to bf16 on their way into LDS (activations AND weights: both stay fp32 in HBM), v_mfma_f32_32x32x16_bf16, fp32
// accumulation and epilogue.  PREC 2 (split): each operand as hi = bf16(v) and lo = bf16(v - hi) in two image planes, a product as
// lo*hi + hi*lo + hi*hi on three MFMAs (the split mode of convlstm_bf16.hip).  PREC 0: fp32 MFMA.
// PREC 3: two FP16 pieces (22 bits of operand; the weights times the power of two of d.wscale_part, the sum scaled back), lo*hi + hi*lo + hi*hi on
// three fp16 MFMAs: fp32-grade (precision mode PIVP_PRECISION_FP16X3).
// The input may be a concat of two tensors (x0: c0 channels | x1: c1 channels, whole 32-channel chunks each: [hidden6 | enc1], [hidden7 | enc0]),
// and with IN_LN the x0 part is a RAW ConvLSTM output whose LayerNorm (TM:203-208: per-element gamma / beta [Hin*Win][c0], statistics merged
// from the producer's partials d.in_part) is applied while the patch is staged -- the expression of ln_apply_kernel, bit-identical to the
// two-launch form; pixels outside the image load 0 for v, gamma and beta alike and stay 0.  Inference rollouts use it for enc5 / enc6.
template <int PREC, bool IN_LN>
__global__ __launch_bounds__(256, 2) void deconv3x3s2_tile_kernel(const IgemmDesc d) {
    PIVP_SET_MAIN_PRIO();
    DT_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) float lds[];   // A patch | the 9 weight tiles
    const int gx = (int)gridDim.x - d.rd_blocks;                  // the launch's own tiles; behind them: the motion head's finisher, one block per sample
    if (d.rd_blocks && (int)blockIdx.x >= gx) {
        const int rb = (int)blockIdx.x - gx;
        if (d.rd_mode == 1) cdna_finish_block(d.rd_partials, d.rd_bias, d.rd_out, d.rd_blocks, d.rd_KS, d.rd_nout, d.rd_vpre, rb, lds);
        else stp_finish_block(d.rd_partials, d.rd_bias, d.rd_w2, d.rd_b2, d.rd_out, d.rd_blocks, d.rd_KS, d.rd_vpre, rb, lds);
        return;
    }
    float* const At = lds;
    float* const Bt = lds + A_FL;
    constexpr bool BF16 = PREC != 0;
    constexpr int HPL = A_HB + 9 * B_HB;                                     // bytes of one plane of bf16 images (patch | 9 weight tiles)
    unsigned char* const Ah = reinterpret_cast<unsigned char*>(lds);       // BF16: byte-addressed images; split mode: the lo plane at + HPL
    unsigned char* const Bh = Ah + A_HB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int H = d.Hin, W = d.Win, N = d.N;
    const int n_nblk = N >> 5, tpr = W >> 4, tpi = (H >> 3) * tpr;
    const int n_tiles = d.B * tpi;
    int lid = blockIdx.x;
    if ((gx & 7) == 0) lid = (blockIdx.x & 7) * (gx >> 3) + (blockIdx.x >> 3);   // XCD-aware, column-block major
    const int nblk = lid / n_tiles, tile = lid - nblk * n_tiles;
    const int b = tile / tpi, trem = tile - b * tpi;
    const int y0 = (trem / tpr) * 8, x0 = (trem - (trem / tpr) * tpr) * 16;
    const int ncc0 = d.c0 >> 5, ncc = (d.c0 + d.c1) >> 5;

    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x0), 0, d.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.c1 ? d.x1 : d.x0), 0, d.c1 ? d.bytes1 : d.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(IN_LN ? d.in_g : d.x0), 0, IN_LN ? H * W * d.c0 * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(IN_LN ? d.in_b : d.x0), 0, IN_LN ? H * W * d.c0 * 4 : 0, 0x00020000);
    float ln_mean = 0.f, ln_rstd = 1.f;
    f32x4 ln_first = {0.f, 0.f, 0.f, 0.f};
    if constexpr (IN_LN) ln_first = ln_partial_first(d.in_part, b, d.in_np);      // requested here, merged behind the first chunk's loads (below)
    DT_STAMP(1);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, d.bytesw, 0x00020000);
    float wscale = 1.0f;
    if constexpr (PREC == 3) wscale = d.wscale_part ? pivp_x3_scale_wave(d.wscale_part) : 1.0f;
    constexpr unsigned OOB = 0xC0000000u;

    // ---- staging roles ----------------------------------------------------------------------------------------------------
    // patch: 153 pixels x 8 float4: thread -> (pixel tid / 8 + 32 j, c4 = tid % 8), j < 5
    const int c4 = tid & 7;
    unsigned a_go[5], a_g1[5], g_go[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int p = (tid >> 3) + 32 * j;
        const int py = p / PC, px = p - py * PC;
        const int iy = y0 + py, ix = x0 + px;
        const bool ok = p < PR * PC && iy < H && ix < W;
        a_go[j] = ok ? (unsigned)((((b * H + iy) * W + ix) * d.ld0 + c4 * 4) * 4) : OOB;
        a_g1[j] = ok ? (unsigned)((((b * H + iy) * W + ix) * d.ld1 + c4 * 4) * 4) : OOB;
        g_go[j] = ok ? (unsigned)(((iy * W + ix) * d.c0 + c4 * 4) * 4) : OOB;
    }
    // Per 32-channel chunk the patch AND the weight tiles of all 9 taps ([9][32 columns][32 k], 41 KB) are staged together, so the 9 taps
    // (144 MFMAs per wave) run without a barrier; the next chunk's 14 float4 per thread are in flight in registers meanwhile.
    const int b_go = (((nblk * 32 + (tid >> 3)) * 32) + c4 * 4) * 4;
    const int b_lw = (tid >> 3) * DP + c4 * 4;
    f32x4 rp[5], rw[9];
    f32x4 rgm[5], rbt[5];              // IN_LN: gamma / beta of the staged float4s
    bool ln_chunk = false;             // the chunk in rp is part of the normalised tensor (block-uniform)
    int ln_cc = 0;                     // ... and which of its 32-channel chunks
    auto load_chunk = [&](int cc) {
        const bool first = cc < ncc0;
        ln_cc = cc;
#pragma unroll
        for (int j = 0; j < 5; ++j)
            rp[j] = __builtin_bit_cast(f32x4, first ? __builtin_amdgcn_raw_buffer_load_b128(rsx, a_go[j], cc * 128, 0)
                                                    : __builtin_amdgcn_raw_buffer_load_b128(rsx1, a_g1[j], (cc - ncc0) * 128, 0));
        if constexpr (IN_LN) {
            ln_chunk = first;
            if (first) {
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    rgm[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsg, g_go[j], cc * 128, 0));
                    rbt[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, g_go[j], cc * 128, 0));
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int sbase = __builtin_amdgcn_readfirstlane((t * (d.wcin >> 5) + cc) * N * 128);
            rw[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, b_go, sbase, 0));
        }
    };
    auto store_chunk = [&]() {
        if constexpr (IN_LN) {
            if (ln_chunk) {
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) rp[j][e] = (rp[j][e] - ln_mean) * ln_rstd * rgm[j][e] + rbt[j][e];
                if (d.in_out && nblk == 0) {      // the tile's own 8 x 16 pixels (not the halo), by the first column block
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const int p = (tid >> 3) + 32 * j;
                        const int py = p / PC, px = p - py * PC;
                        if (py < 8 && px < 16)
                            *reinterpret_cast<f32x4*>(d.in_out + ((size_t)(b * H + y0 + py) * W + x0 + px) * d.in_out_ld + ln_cc * 32 + c4 * 4) = rp[j];
                    }
                }
            }
        }
        if constexpr (PREC == 3) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                float r[4] = {rp[j][0], rp[j][1], rp[j][2], rp[j][3]};
                uint2 v, l;
                v.x = pivp_pack2h_rest(r[0], r[1]); v.y = pivp_pack2h_rest(r[2], r[3]);
                l.x = pivp_pack2h_rest(r[0], r[1]); l.y = pivp_pack2h_rest(r[2], r[3]);
                *reinterpret_cast<uint2*>(Ah + ((tid >> 3) + 32 * j) * HP + c4 * 8) = v;
                *reinterpret_cast<uint2*>(Ah + HPL + ((tid >> 3) + 32 * j) * HP + c4 * 8) = l;
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float r[4] = {rw[t][0] * wscale, rw[t][1] * wscale, rw[t][2] * wscale, rw[t][3] * wscale};
                uint2 v, l;
                v.x = pivp_pack2h_rest(r[0], r[1]); v.y = pivp_pack2h_rest(r[2], r[3]);
                l.x = pivp_pack2h_rest(r[0], r[1]); l.y = pivp_pack2h_rest(r[2], r[3]);
                *reinterpret_cast<uint2*>(Bh + t * B_HB + (tid >> 3) * HP + c4 * 8) = v;
                *reinterpret_cast<uint2*>(Bh + HPL + t * B_HB + (tid >> 3) * HP + c4 * 8) = l;
            }
        } else if constexpr (BF16) {
            auto lo2 = [](unsigned hi2, float a, float b) {     // bf16(v - hi): hi as a float is its 16 bits shifted up
                return dpack2(a - __builtin_bit_cast(float, hi2 << 16), b - __builtin_bit_cast(float, hi2 & 0xffff0000u));
            };
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                uint2 v; v.x = dpack2(rp[j][0], rp[j][1]); v.y = dpack2(rp[j][2], rp[j][3]);
                *reinterpret_cast<uint2*>(Ah + ((tid >> 3) + 32 * j) * HP + c4 * 8) = v;
                if constexpr (PREC == 2) {
                    uint2 l; l.x = lo2(v.x, rp[j][0], rp[j][1]); l.y = lo2(v.y, rp[j][2], rp[j][3]);
                    *reinterpret_cast<uint2*>(Ah + HPL + ((tid >> 3) + 32 * j) * HP + c4 * 8) = l;
                }
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                uint2 v; v.x = dpack2(rw[t][0], rw[t][1]); v.y = dpack2(rw[t][2], rw[t][3]);
                *reinterpret_cast<uint2*>(Bh + t * B_HB + (tid >> 3) * HP + c4 * 8) = v;
                if constexpr (PREC == 2) {
                    uint2 l; l.x = lo2(v.x, rw[t][0], rw[t][1]); l.y = lo2(v.y, rw[t][2], rw[t][3]);
                    *reinterpret_cast<uint2*>(Bh + HPL + t * B_HB + (tid >> 3) * HP + c4 * 8) = l;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j)      // unconditional: 160 pixel rows are allocated, rows past 153 receive zeros
                *reinterpret_cast<f32x4*>(At + ((tid >> 3) + 32 * j) * DP + c4 * 4) = rp[j];
#pragma unroll
            for (int t = 0; t < 9; ++t) *reinterpret_cast<f32x4*>(Bt + t * B_FL + b_lw) = rw[t];
        }
    };

    f32x16 acc[4];                     // output parity (py, px) -> acc[2 py + px]
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ph][r] = 0.f;

    // A fragment: row l31 of the wave's 32 anchors = patch pixel (2 wave + l31 / 16, l31 % 16), shifted by the tap; B: column l31
    const int a_lane = ((2 * wave + (l31 >> 4)) * PC + (l31 & 15)) * DP + 4 * half;
    const int b_lane = l31 * DP + 4 * half;

    // one tap of one 32-channel chunk: 16 MFMAs (k = 32) per wave
    auto tap_mfmas = [&](auto TAP) {
        constexpr int tap = decltype(TAP)::value, ky = tap / 3, kx = tap % 3;
        constexpr int ph = 2 * (ky != 1) + (kx != 1);
        if constexpr (BF16) {              // k = 32 in two MFMAs; lane: row l31, k = 16 ks + 8 half .. + 7
            constexpr int prow = (ky == 0) * PC + (kx == 0);
            const unsigned char* Ar = Ah + ((2 * wave + (l31 >> 4)) * PC + (l31 & 15) + prow) * HP + half * 16;
            const unsigned char* Br = Bh + tap * B_HB + l31 * HP + half * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(Ar + ks * 32), bh = *reinterpret_cast<const bf16x8*>(Br + ks * 32);
                if constexpr (PREC == 3) {
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(Ar + HPL + ks * 32), bl = *reinterpret_cast<const bf16x8*>(Br + HPL + ks * 32);
                    auto h = [](const bf16x8& v) { return __builtin_bit_cast(pivp_f16x8, v); };
                    // (one accumulator: K is 36-54 k-steps here, three roundings per k-step stay below the fp32 MFMA's one per product; a second
                    // set of four accumulators spilled 109 registers in the form that applies the LayerNorm while staging)
                    acc[ph] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(al), h(bh), acc[ph], 0, 0, 0);
                    acc[ph] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(ah), h(bl), acc[ph], 0, 0, 0);
                    acc[ph] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(ah), h(bh), acc[ph], 0, 0, 0);
                    continue;
                }
                if constexpr (PREC == 2) {
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(Ar + HPL + ks * 32), bl = *reinterpret_cast<const bf16x8*>(Br + HPL + ks * 32);
                    acc[ph] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[ph], 0, 0, 0);
                    acc[ph] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[ph], 0, 0, 0);
                }
                acc[ph] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[ph], 0, 0, 0);
            }
            return;
        }
        constexpr int shift = ((ky == 0) * PC + (kx == 0)) * DP;
        const float* As = At + a_lane + shift;
        const float* Bs = Bt + tap * B_FL + b_lane;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const f32x4 fa = *reinterpret_cast<const f32x4*>(As + 8 * qd);
            const f32x4 fb = *reinterpret_cast<const f32x4*>(Bs + 8 * qd);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[ph] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc[ph], 0, 0, 0);
        }
    };

    // ---- main loop over the 32-channel chunks ----------------------------------------------------------------------------------------
    load_chunk(0);
    if constexpr (IN_LN) {
        ln_merge_partials(ln_first, d.in_part, b, d.in_np, d.in_eps, ln_mean, ln_rstd);      // every wave for itself: a few partials per sample
        if (d.in_stat_out && nblk == 0 && trem == 0 && tid == 0) { d.in_stat_out[b * 2] = ln_mean; d.in_stat_out[b * 2 + 1] = ln_rstd; }
    }
    store_chunk();
    __syncthreads();
    DT_STAMP(2);
    for (int cc = 0; cc < ncc; ++cc) {
        if (cc + 1 < ncc) load_chunk(cc + 1);
        tap_mfmas(std::integral_constant<int, 0>{}); tap_mfmas(std::integral_constant<int, 1>{}); tap_mfmas(std::integral_constant<int, 2>{});
        tap_mfmas(std::integral_constant<int, 3>{}); tap_mfmas(std::integral_constant<int, 4>{}); tap_mfmas(std::integral_constant<int, 5>{});
        tap_mfmas(std::integral_constant<int, 6>{}); tap_mfmas(std::integral_constant<int, 7>{}); tap_mfmas(std::integral_constant<int, 8>{});
        if (cc + 1 < ncc) {
            __syncthreads();               // every wave is done with this chunk's images
            store_chunk();
            __syncthreads();
        }
    }

    DT_STAMP(3);
    // ---- epilogue: accumulator row i -> anchor (2 wave + i / 16, i % 16), parity (py, px) -> output pixel (2 y + py, 2 x + px) ----------
    const int col = nblk * 32 + l31;
    const float bias = d.bias ? d.bias[col] : 0.f;
    const float inv_wscale = 1.0f / wscale;
    float s1 = 0.f;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int oy = 2 * (y0 + 2 * wave + (i >> 4)) + (ph >> 1), ox = 2 * (x0 + (i & 15)) + (ph & 1);
            float* o = d.out + ((size_t)(b * d.Hout + oy) * d.Wout + ox) * d.ldo + col;
            float v = acc[ph][r];
            if constexpr (PREC == 3) v *= inv_wscale;
            v += bias;
            if (d.relu) v = fmaxf(v, 0.f);
            if (d.accum) v += *o;
            *o = v;      // (as a nontemporal store: enc5 / enc6 -1.3 us per launch, frame_head +1.0 reading what then comes from memory: not taken, profiles/r06/NOTES.md 5)
            acc[ph][r] = v;
            s1 += v;
        }
    if (d.ln_part) {   // (count, mean, M2) of the block's 128 x 4 x 32 outputs: two passes over registers, fixed order
        float* red = lds;
        s1 = wave_sum(s1);
        __syncthreads();
        if (lane == 0) red[wave] = s1;
        __syncthreads();
        const float cnt = 16384.f;
        const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / cnt;
        float q = 0.f;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float dd = acc[ph][r] - mean; q = fmaf(dd, dd, q); }
        q = wave_sum(q);
        if (lane == 0) red[4 + wave] = q;
        __syncthreads();
        if (tid == 0) {
            float* p = d.ln_part + ((size_t)b * d.ln_nparts + (size_t)trem * n_nblk + nblk) * 4;
            p[0] = cnt; p[1] = mean; p[2] = (red[4] + red[5]) + (red[6] + red[7]); p[3] = 0.f;
        }
    }
#ifdef PIVP_DT_STAMPS
    DT_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DT_STAMP(5);
#endif
}

bool deconv_tile_ok(const IgemmDesc& d) {
    return d.deconv && d.nphase == 4 && d.c0 > 0 && d.c0 % 32 == 0 && d.c1 % 32 == 0 && (d.c1 == 0 || (d.x1 && d.ld1 % 4 == 0)) && d.N % 32 == 0 &&
           d.Hin % 8 == 0 && d.Win % 16 == 0 && d.Hout == 2 * d.Hin && d.Wout == 2 * d.Win && d.ld0 % 4 == 0 && d.out != nullptr &&
           (!d.in_g || (d.in_b && d.in_part && d.in_np > 0 && d.ld0 == d.c0));
}

// d as igemm_conv takes it for the transposed conv (validated by the caller); ln_nparts as there.
int deconv_tile(const IgemmDesc& d, hipStream_t stream, int* ln_nparts, int prec) {
    PIVP_CHECK_ARG(deconv_tile_ok(d));
    constexpr int lds_f32 = (A_FL + 9 * B_FL) * 4;        // 64,512 (the bf16 images fit inside)
    static PerDeviceOnce once0, once2, once0n, once2n, once3, once3n;
    if (prec == 3 && (pivp_ensure_dyn_lds(once3, reinterpret_cast<const void*>(&deconv3x3s2_tile_kernel<3, false>), 2 * (A_HB + 9 * B_HB)) != PIVP_OK ||
                      pivp_ensure_dyn_lds(once3n, reinterpret_cast<const void*>(&deconv3x3s2_tile_kernel<3, true>), 2 * (A_HB + 9 * B_HB)) != PIVP_OK))
        return PIVP_ERR_LAUNCH;
    if (pivp_ensure_dyn_lds(once0, reinterpret_cast<const void*>(&deconv3x3s2_tile_kernel<0, false>), lds_f32) != PIVP_OK ||
        pivp_ensure_dyn_lds(once2, reinterpret_cast<const void*>(&deconv3x3s2_tile_kernel<2, false>), 2 * (A_HB + 9 * B_HB)) != PIVP_OK ||
        pivp_ensure_dyn_lds(once0n, reinterpret_cast<const void*>(&deconv3x3s2_tile_kernel<0, true>), lds_f32) != PIVP_OK ||
        pivp_ensure_dyn_lds(once2n, reinterpret_cast<const void*>(&deconv3x3s2_tile_kernel<2, true>), 2 * (A_HB + 9 * B_HB)) != PIVP_OK)
        return PIVP_ERR_LAUNCH;
    IgemmDesc dd = d;
    const int tpi = (d.Hin / 8) * (d.Win / 16), nb = d.N / 32;
    const int np = tpi * nb;
    dd.ln_nparts = (d.ln_part && np <= d.ln_cap) ? np : 0;
    if (!dd.ln_nparts) dd.ln_part = nullptr;
    if (ln_nparts) *ln_nparts = dd.ln_nparts;
    PIVP_CHECK_ARG(d.rd_mode == 0 || ((d.rd_mode == 1 || d.rd_mode == 2) && d.rd_blocks == d.B && d.rd_partials && d.rd_bias && d.rd_out && d.rd_KS >= 1 &&
                                      (d.rd_mode == 1 ? (d.rd_nout >= 25 && d.rd_nout <= 256) : (d.rd_w2 && d.rd_b2))));
    if (!d.rd_mode) dd.rd_blocks = 0;
    const dim3 grid(d.B * tpi * nb + dd.rd_blocks);
    if (prec == 3) {
        if (d.in_g) hipLaunchKernelGGL((deconv3x3s2_tile_kernel<3, true>), grid, dim3(256), 2 * (A_HB + 9 * B_HB), stream, dd);
        else hipLaunchKernelGGL((deconv3x3s2_tile_kernel<3, false>), grid, dim3(256), 2 * (A_HB + 9 * B_HB), stream, dd);
    } else if (d.in_g) {
        if (prec == 2) hipLaunchKernelGGL((deconv3x3s2_tile_kernel<2, true>), grid, dim3(256), 2 * (A_HB + 9 * B_HB), stream, dd);
        else if (prec == 1) hipLaunchKernelGGL((deconv3x3s2_tile_kernel<1, true>), grid, dim3(256), A_HB + 9 * B_HB, stream, dd);
        else hipLaunchKernelGGL((deconv3x3s2_tile_kernel<0, true>), grid, dim3(256), lds_f32, stream, dd);
    } else if (prec == 2) hipLaunchKernelGGL((deconv3x3s2_tile_kernel<2, false>), grid, dim3(256), 2 * (A_HB + 9 * B_HB), stream, dd);
    else if (prec == 1) hipLaunchKernelGGL((deconv3x3s2_tile_kernel<1, false>), grid, dim3(256), A_HB + 9 * B_HB, stream, dd);
    else hipLaunchKernelGGL((deconv3x3s2_tile_kernel<0, false>), grid, dim3(256), lds_f32, stream, dd);
    return PIVP_LAUNCH_STATUS();
}

}  // namespace pivp

PIVP_DEFINE_MAIN_PRIO_SETTER(deconv_tile)

#ifdef PIVP_DT_STAMPS
extern "C" int pivp_debug_dt_stamps(long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pivp_dt_stamps), sizeof(long long) * n) == hipSuccess ? 0 : -2;
}
#endif
