// ConvLSTM gate convolution, spatial-tile variant: the block's input patch is staged ONCE per 32 input channels and
// serves all 25 taps.
//
// igemm_f32.hip gathers a fresh A tile (BM anchors x 32 channels) for every (tap, channel chunk): 25 gathers of
// mostly the same pixels, each with its own address arithmetic, buffer loads and ds_writes -- a third of that
// kernel's staging work, which the ablations in profiles/r01/NOTES.md price at ~10 % of its run time.  Here a block
// owns a 4 x 16 patch of anchors of ONE image (64 anchors, as the 2x2-wave tile there) and 32 channels x 4 gates.  For
// each 32-channel chunk of the input it stages the (4+4) x (16+4) halo patch into LDS once ([160 pixels][36] floats,
// zero outside the image: the hardware bounds check of the buffer load returns 0) and runs the 25 taps against it: tap
// (ty, tx) only shifts the LDS address of the A fragment by (ty * 20 + tx) pixels.  Per chunk only the 16 KB weight
// tile is staged (4 loads + 4 ds_writes per thread instead of 6 + 6 with per-piece bounds logic), the next channel
// chunk's patch travels through registers during taps 19..23 and is written at the chunk boundary (single A
// buffer: 23 KB + 2 x 18.4 KB of weights = 60 KB, two blocks per CU).
// A 16-lane phase of a ds_read_b128 covers 16 consecutive patch pixels at a 144-B pitch: conflict-free.
// MFMA, accumulator layout, gate epilogue (shuffle-gather of the 4 gates, c/h update, optional gate activations for
// BPTT, LayerNorm partial) are those of igemm_f32_kernel<2, 2, 4, true>; results are bit-identical to it when the
// channel-chunk-major K order is taken into account (same products, different summation order).
// Shapes: 5x5 stride-1 "same" conv, W % 16 == 0, H % 4 == 0 (the 32x32 and 16x16 maps; 8x8 stays on the gather kernel).
#include <type_traits>

#include "pivp_kernels.h"

namespace pivp {

namespace {
constexpr int TP = 36;                 // LDS row pitch (floats)
constexpr int TH = 8, TW = 20;         // halo patch: (4 + 4) rows x (16 + 4) columns
constexpr int A_FL = TH * TW * TP;     // 5760 floats
__device__ __forceinline__ float t_sigmoid(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float t_tanh(float x) { return 2.0f * __frcp_rn(1.0f + __expf(-2.0f * x)) - 1.0f; }
}

// NCH: channels per block (x 4 gates = block columns).  32: wave tile 32 anchors x (4 gates x 16 channels), two MFMA tiles.
// 16: wave tile 32 anchors x (4 gates x 8 channels), one MFMA tile -- twice the blocks, for the 16x16 maps whose 256 blocks
// of 32 channels leave one block (one wave per SIMD) on every CU.
template <int NCH>
__global__ __launch_bounds__(256, 1) void convlstm_tile_kernel(const IgemmDesc d) {
    constexpr int B_FL = 4 * NCH * TP;          // weight tile: [4 gates x NCH channels][36]
    constexpr int TPW = NCH / 16;               // MFMA tiles per wave
    constexpr int CPW = NCH / 2;                // channels per wave (all 4 gates of a channel stay in one wave)
    constexpr int GPT = 32 / CPW;               // gates per MFMA tile
    constexpr int NB = NCH / 8;                 // weight float4 per thread and chunk
    extern __shared__ __attribute__((aligned(16))) float lds[];   // A | B0 | B1
    float* const At = lds;
    float* const Bt = lds + A_FL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int half = lane >> 5, l31 = lane & 31;
    const int C = d.C, n_nblk = C / NCH;
    const int H = d.Hin, W = d.Win;
    const int pw = W >> 4, ph = H >> 2;                // patches per row / column
    const int n_patch = d.B * ph * pw;
    // XCD-aware order, column-block major (as igemm_f32.hip)
    int lid = blockIdx.x;
    if ((gridDim.x & 7) == 0) lid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int nblk = lid / n_patch, patch = lid - nblk * n_patch;
    const int b = patch / (ph * pw), prem = patch - b * ph * pw;
    const int y0 = (prem / pw) * 4, x0 = (prem - (prem / pw) * pw) * 16;
    const int cin = d.c0 + d.c1, ncc = cin >> 5;
    const int nchunks = 25 * ncc;

    constexpr unsigned OOB = 0xC0000000u;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x0), 0, d.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.c1 ? d.x1 : d.x0), 0, d.c1 ? d.bytes1 : d.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, d.bytesw, 0x00020000);

    // ---- staging roles ---------------------------------------------------------------------------------------------
    // A patch: 160 pixels x 8 float4 = 1280 float4, 5 per thread; offsets (or the out-of-range marker) fixed per thread
    const int cvec = tid & 7;
    unsigned a_go0[5], a_go1[5];
    int a_lw[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int tp = (tid >> 3) + 32 * j;                  // patch pixel 0..159
        const int ty = tp / TW, tx = tp - ty * TW;
        const int iy = y0 - 2 + ty, ix = x0 - 2 + tx;
        const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const int pix = (b * H + iy) * W + ix;
        a_go0[j] = ok ? (unsigned)((pix * d.ld0 + cvec * 4) * 4) : OOB;
        a_go1[j] = ok ? (unsigned)((pix * d.ld1 + cvec * 4) * 4) : OOB;
        a_lw[j] = tp * TP + cvec * 4;
    }
    // B tile: [4 gates x NCH channels][32 k]: NB float4 per thread; tile row = gate * NCH + channel
    const int prow = tid >> 3;
    int b_go[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int row = prow + 32 * j, g = row / NCH, cl = row - g * NCH;
        b_go[j] = ((g * C + nblk * NCH + cl) * 32 + cvec * 4) * 4;
    }
    const int b_lw = prow * TP + cvec * 4;

    // two weight register sets: the tile of chunk i+2 is loaded during chunk i and written to LDS during chunk i+1, so a load
    // has a whole chunk (2-4k cycles) to arrive before its ds_write waits for it.  With one set (load in micro-steps 0-3, write
    // in 12-15 of the same chunk) the writes waited on L2 / MALL latency every chunk (igemm_wgrad.hip found the same: -9 %).
    f32x4 ra[5], rb[2][NB];
    // next weight chunk to load: tap l_tap of channel chunk l_cc (taps innermost)
    int l_tap = 0, l_cc = 0;
    int s_wbase = 0;
    auto next_b = [&]() {
        s_wbase = __builtin_amdgcn_readfirstlane((l_tap * (d.wcin >> 5) + l_cc) * d.N * 128);
        const bool w = ++l_tap == 25;
        l_tap = w ? 0 : l_tap;
        l_cc += w ? 1 : 0;
    };
    auto load_b = [&](auto SET, auto J) {
        constexpr int j = decltype(J)::value;
        if constexpr (j < NB) rb[decltype(SET)::value][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, b_go[j], s_wbase, 0));
    };
    auto store_b = [&](auto SET, auto J, int buf) {
        constexpr int j = decltype(J)::value;
        if constexpr (j < NB) *reinterpret_cast<f32x4*>(Bt + buf * B_FL + b_lw + 32 * j * TP) = rb[decltype(SET)::value][j];
    };
    auto load_a = [&](auto J, int cc) {     // piece J of channel chunk cc's patch
        constexpr int j = decltype(J)::value;
        const int ch = cc << 5;
        const bool first = ch < d.c0;
        const int soff = __builtin_amdgcn_readfirstlane((first ? ch : ch - d.c0) * 4);
        ra[j] = __builtin_bit_cast(f32x4, first ? __builtin_amdgcn_raw_buffer_load_b128(rs0, a_go0[j], soff, 0)
                                                : __builtin_amdgcn_raw_buffer_load_b128(rs1, a_go1[j], soff, 0));
    };
    auto store_a = [&]() {
#pragma unroll
        for (int j = 0; j < 5; ++j) *reinterpret_cast<f32x4*>(At + a_lw[j]) = ra[j];
    };

    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // fragment addresses (floats): A row of lane l31 = anchor (2*wm + l31/16, l31%16) of the patch, shifted by the tap
    const int a_lane = ((2 * wm + (l31 >> 4)) * TW + (l31 & 15)) * TP + 4 * half;
    int b_off[TPW];   // MFMA column l31 of tile t = gate t * GPT + l31 / CPW, channel wn * CPW + l31 % CPW
#pragma unroll
    for (int t = 0; t < TPW; ++t) b_off[t] = ((t * GPT + l31 / CPW) * NCH + wn * CPW + (l31 % CPW)) * TP + 4 * half;

    // ---- one chunk = tap c_tap of channel chunk c_cc: 16 micro-steps of 2 MFMAs ----------------------------------------
    int c_tap = 0, c_cc = 0;
    // LOAD: this chunk issues the weight loads of chunk +2 into register set SET; STORE: it writes set SET^1 (chunk +1) to the
    // other LDS buffer
    auto chunk = [&](auto LOAD, auto STORE, auto SET, int buf) {
        constexpr bool do_load = decltype(LOAD)::value, stage = decltype(STORE)::value;
        using OTHER = std::integral_constant<int, decltype(SET)::value ^ 1>;
        const int ty = c_tap / 5, tx = c_tap - ty * 5;                     // scalar
        const float* As = At + a_lane + (ty * TW + tx) * TP;
        const float* Bs = Bt + buf * B_FL;
        const bool a_next = stage && c_tap >= 19 && c_tap < 24 && c_cc + 1 < ncc;   // this chunk carries a piece of the next patch
        f32x4 fa[2], fb[2][TPW];
        fa[0] = *reinterpret_cast<const f32x4*>(As);
#pragma unroll
        for (int t = 0; t < TPW; ++t) fb[0][t] = *reinterpret_cast<const f32x4*>(Bs + b_off[t]);
        __builtin_amdgcn_sched_barrier(0);
        auto micro = [&](auto Q, auto S2) {
            constexpr int q = decltype(Q)::value, s2 = decltype(S2)::value, step = q * 4 + s2;
            constexpr int cur = q & 1, nxt = cur ^ 1;
            if constexpr (s2 == 0 && q < 3) {
                fa[nxt] = *reinterpret_cast<const f32x4*>(As + 8 * (q + 1));
#pragma unroll
                for (int t = 0; t < TPW; ++t) fb[nxt][t] = *reinterpret_cast<const f32x4*>(Bs + b_off[t] + 8 * (q + 1));
            }
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s2], fb[cur][t][s2], acc[t], 0, 0, 0);
            if constexpr (do_load && step < 4) load_b(SET, std::integral_constant<int, step>{});
            if constexpr (do_load && step == 4) next_b();                  // scalars of the loads of the next chunk
            if constexpr (stage && step == 5) {
                if (a_next) {                                              // uniform branch
                    switch (c_tap) {
                        case 19: load_a(std::integral_constant<int, 0>{}, c_cc + 1); break;
                        case 20: load_a(std::integral_constant<int, 1>{}, c_cc + 1); break;
                        case 21: load_a(std::integral_constant<int, 2>{}, c_cc + 1); break;
                        case 22: load_a(std::integral_constant<int, 3>{}, c_cc + 1); break;
                        default: load_a(std::integral_constant<int, 4>{}, c_cc + 1); break;
                    }
                }
            }
            if constexpr (stage && step >= 12) store_b(OTHER{}, std::integral_constant<int, step - 12>{}, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto qgroup = [&](auto Q) {
            micro(Q, std::integral_constant<int, 0>{}); micro(Q, std::integral_constant<int, 1>{});
            micro(Q, std::integral_constant<int, 2>{}); micro(Q, std::integral_constant<int, 3>{});
        };
        qgroup(std::integral_constant<int, 0>{}); qgroup(std::integral_constant<int, 1>{});
        qgroup(std::integral_constant<int, 2>{}); qgroup(std::integral_constant<int, 3>{});
    };

    // prologue: patch of channel chunk 0, weights of chunk 0 (to LDS) and of chunk 1 (stay in register set 1)
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    load_a(I0{}, 0); load_a(I1{}, 0); load_a(I2{}, 0); load_a(I3{}, 0); load_a(std::integral_constant<int, 4>{}, 0);
    next_b();
    load_b(S0{}, I0{}); load_b(S0{}, I1{}); load_b(S0{}, I2{}); load_b(S0{}, I3{});
    next_b();
    if (nchunks > 1) { load_b(S1{}, I0{}); load_b(S1{}, I1{}); load_b(S1{}, I2{}); load_b(S1{}, I3{}); }
    next_b();                                  // scalars of chunk 2 (loaded while chunk 0 is multiplied)
    store_a();
    store_b(S0{}, I0{}, 0); store_b(S0{}, I1{}, 0); store_b(S0{}, I2{}, 0); store_b(S0{}, I3{}, 0);
    __syncthreads();
    auto advance = [&]() {
        __syncthreads();
        if (++c_tap == 25) {                   // channel-chunk boundary: the next patch moves from registers to LDS
            c_tap = 0; ++c_cc;
            store_a();
            __syncthreads();
        }
    };
    // two chunks per trip (register sets 0 and 1 alternate statically: a run-time parity branch made hipcc fall back to
    // vmcnt(0) in front of every ds_write); nchunks = 25 * ncc, `it` stays even
    int it = 0;
    for (; it + 3 < nchunks; it += 2) {        // chunk it loads it+2 into set 0 and writes it+1 (set 1); chunk it+1 the reverse
        chunk(std::true_type{}, std::true_type{}, S0{}, 0); advance();
        chunk(std::true_type{}, std::true_type{}, S1{}, 1); advance();
    }
    const int rest = nchunks - it;             // 1, 2 or 3 chunks left
    if (rest == 3) {
        chunk(std::true_type{}, std::true_type{}, S0{}, 0); advance();
        chunk(std::false_type{}, std::true_type{}, S1{}, 1); advance();
        chunk(std::false_type{}, std::false_type{}, S0{}, 0);
    } else if (rest == 2) {
        chunk(std::false_type{}, std::true_type{}, S0{}, 0); advance();
        chunk(std::false_type{}, std::false_type{}, S0{}, 1);
    } else {
        chunk(std::false_type{}, std::false_type{}, S0{}, 0);
    }

    // ---- epilogue: gates, state update, optional gate activations and LayerNorm partial ------------------------------
    const int chl = wn * CPW + (l31 % CPW);
    const int ch = nblk * NCH + chl;
    const int grp = l31 / CPW;
    const float bj = d.bias[ch], bi = d.bias[C + ch], bf = d.bias[2 * C + ch] + 1.0f, bo = d.bias[3 * C + ch];
    float sv[16];
    unsigned own = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        sv[r] = 0.f;
        float g4[4];
#pragma unroll
        for (int G = 0; G < 4; ++G) g4[G] = __shfl(acc[G / GPT][r], (l31 % CPW) + CPW * (G % GPT) + 32 * half, 64);
        const int i = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;        // anchor within the block: patch (i / 16, i % 16)
        const int m = (b * H + y0 + (i >> 4)) * W + x0 + (i & 15);
        if (grp == (r % GPT)) {
            const size_t o = (size_t)m * C + ch;
            const float aj = t_tanh(g4[0] + bj), ai = t_sigmoid(g4[1] + bi);
            const float af = t_sigmoid(g4[2] + bf), ao = t_sigmoid(g4[3] + bo);
            const float cn = d.cstate_in[o] * af + ai * aj;
            d.cstate_out[o] = cn;
            const float hn = t_tanh(cn) * ao;
            d.hout[o] = hn;
            sv[r] = hn; own |= 1u << r;
            if (d.gates_out) {
                float* gp = d.gates_out + (size_t)m * 4 * C + ch;
                gp[0] = aj; gp[C] = ai; gp[2 * C] = af; gp[3 * C] = ao;
            }
        }
    }
    if (d.ln_part) {   // (count, mean, M2) of the block's h tile: two passes over registers, fixed order
        float s1 = 0.f, c1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if ((own >> r) & 1) { s1 += sv[r]; c1 += 1.f; }
        s1 = wave_sum(s1); c1 = wave_sum(c1);
        __syncthreads();
        if (lane == 0) { lds[wave] = s1; lds[4 + wave] = c1; }
        __syncthreads();
        const float cnt = (lds[4] + lds[5]) + (lds[6] + lds[7]);
        const float mean = ((lds[0] + lds[1]) + (lds[2] + lds[3])) / cnt;
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if ((own >> r) & 1) { const float dd = sv[r] - mean; q = fmaf(dd, dd, q); }
        q = wave_sum(q);
        if (lane == 0) lds[8 + wave] = q;
        __syncthreads();
        if (tid == 0) {
            float* p = d.ln_part + ((size_t)b * d.ln_nparts + (size_t)prem * n_nblk + nblk) * 4;
            p[0] = cnt; p[1] = mean; p[2] = (lds[8] + lds[9]) + (lds[10] + lds[11]); p[3] = 0.f;
        }
    }
}

bool convlstm_tile_ok(const IgemmDesc& d) {
    return d.Win % 16 == 0 && d.Hin % 4 == 0 && d.ksize == 5 && d.pad == 2 && d.in_step == 1 && d.C % 32 == 0;
}

template <int NCH>
static int launch_tile(const IgemmDesc& d, hipStream_t stream, int* ln_nparts) {
    constexpr int lds_bytes = (A_FL + 2 * 4 * NCH * TP) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convlstm_tile_kernel<NCH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        attr_set = true;
    }
    IgemmDesc dd = d;
    const int patches = (d.Hin / 4) * (d.Win / 16), nb = d.C / NCH;
    const int np = patches * nb;
    dd.ln_nparts = (d.ln_part && np <= d.ln_cap) ? np : 0;
    if (!dd.ln_nparts) dd.ln_part = nullptr;
    if (ln_nparts) *ln_nparts = dd.ln_nparts;
    hipLaunchKernelGGL(convlstm_tile_kernel<NCH>, dim3(d.B * patches * nb), dim3(256), lds_bytes, stream, dd);
    return PIVP_LAUNCH_STATUS();
}

// d validated by igemm_validate(d, true).  ln_nparts as in igemm_lstm.  nch: 0 = automatic, else 16 / 32 channels per block.
int convlstm_tile(const IgemmDesc& d, hipStream_t stream, int* ln_nparts, int nch) {
    PIVP_CHECK_ARG(convlstm_tile_ok(d) && (nch == 0 || nch == 16 || nch == 32));
    const long blocks32 = (long)d.B * (d.Hin / 4) * (d.Win / 16) * (d.C / 32);
    // 16-channel blocks only when 32-channel ones cannot give every CU a block: on the 16x16 maps at B = 32 two resident
    // 16-channel blocks per CU measured the same as one 32-channel block (lstm3/4/6: 83.8 / 110.5 / 162.6 vs 82.8 / 107.9 / 162.7 us)
    if (nch == 0) nch = blocks32 < 256 ? 16 : 32;
    return nch == 16 ? launch_tile<16>(d, stream, ln_nparts) : launch_tile<32>(d, stream, ln_nparts);
}

}  // namespace pivp
