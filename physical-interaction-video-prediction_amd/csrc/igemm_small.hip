// Small-problem variant of the implicit-GEMM convolution (igemm_f32.hip): 32 x 32 output tiles whose K is
// split over the block's four waves.
//
// The stride-2 convolutions enc1 / enc2 (TM:501-502) and the transposed convolution enc4 (TM:505) have M = 2048 ..
// 8192 anchors and N = 32 .. 128 columns at B = 32: with the 128 x (32..128) tiles of the main kernel they give 16 ..
// 64 workgroups on a 256-CU chip.  Here a workgroup owns a 32-anchor x 32-column tile (256 .. 1024 workgroups for
// the same layers, several resident per CU: 18 KB of LDS each), each K chunk of one tap x 32 input channels is
// staged once per block, and wave w multiplies the chunk's k-group w (8 of the 32 k) -- one ds_read_b128 per operand
// and four v_mfma_f32_32x32x2_f32 per chunk and wave.  The four partial tiles are summed through LDS in a fixed
// order (bitwise reproducible) and the bias / ReLU / accumulate epilogue writes float4 rows.
// Same descriptor, operand layouts, tap enumeration and out-of-image handling (buffer loads with an out-of-range
// offset return 0) as the main kernel.
#include <stdlib.h>
#include <type_traits>

#include "pivp_kernels.h"

namespace pivp {

namespace {
constexpr int SP = 36;   // LDS row pitch in floats: 144-B rows, conflict-free ds_read_b128 / ds_write_b128
constexpr int TILE = 32 * SP;
}

// NTB: 32-column tiles per block (the A fragment is reused NTB times; fewer, longer-running blocks).
template <int NTB, bool IN_LN = false, bool FUSE3 = false>
__global__ __launch_bounds__(256) void igemm_small_kernel(const IgemmDesc d) {
    static_assert(!FUSE3 || NTB == 2, "the fused 1x1 needs all 64 channels of a pixel in one block");
    const int bx = blockIdx.x, by = blockIdx.y, gdy = gridDim.y;
    PIVP_SET_MAIN_PRIO();
    constexpr int BN = 32 * NTB;
    constexpr int RP = BN + 4;                                    // row pitch of the partial-sum image
    constexpr int A_OFF = 0, B_OFF = 2 * TILE;                    // A0 A1 | B0 B1 (NTB tiles each)
    constexpr int STAGE_FLOATS = 2 * TILE + 2 * NTB * TILE, RED_FLOATS = 4 * 32 * RP;
    __shared__ __attribute__((aligned(16))) float lds[STAGE_FLOATS > RED_FLOATS ? STAGE_FLOATS : RED_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int n_nblk = d.n_mblk;                                  // d.N / BN, with the multiplier / shift that divide by it (igemm_small below)
    const int mblk = pivp_fdiv(bx, d.fd_mb_mul, d.fd_mb_sh), nblk = bx - mblk * n_nblk;
    const int m0 = mblk * 32;
    const int phase = by, py = phase >> 1, px = phase & 1;
    const bool deconv = d.deconv != 0;
    const int nty = deconv ? 1 + py : d.ksize, ntx = deconv ? 1 + px : d.ksize;
    const int ncc = (d.c0 + d.c1) >> 5;
    const int nchunks = nty * ntx * ncc;
    const int HWg = d.Hg * d.Wg;

    // staging role: one float4 of the A tile and one of the B tile per thread
    constexpr unsigned OOB = 0xC0000000u;
    const int cvec = tid & 7, prow = tid >> 3;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x0), 0, d.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.c1 ? d.x1 : d.x0), 0, d.c1 ? d.bytes1 : d.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, d.bytesw, 0x00020000);
    int a_off0 = 0, a_off1 = 0, a_iy0 = -(1 << 20), a_ix0 = 0, a_pix0 = 0;
    auto anchor = [&]() {          // (called behind the first chunk's weight loads)
        const int m = m0 + prow;
        if (m < d.M) {
            const int b = pivp_fdiv(m, d.fd_hw_mul, d.fd_hw_sh), rem = m - b * HWg;
            const int ay = pivp_fdiv(rem, d.fd_w_mul, d.fd_w_sh), ax = rem - ay * d.Wg;
            a_iy0 = ay * d.in_step; a_ix0 = ax * d.in_step;
            const int pix = b * d.Hin * d.Win + a_iy0 * d.Win + a_ix0;
            a_pix0 = pix;
            a_off0 = (pix * d.ld0 + cvec * 4) * 4;
            a_off1 = (pix * d.ld1 + cvec * 4) * 4;
        }
    };
    int b_goff[NTB];
#pragma unroll
    for (int t = 0; t < NTB; ++t) b_goff[t] = ((nblk * BN + t * 32 + prow) * 32 + cvec * 4) * 4;
    const int lds_w = prow * SP + cvec * 4;

    // LayerNorm-on-load (d.in_g): the tile's 32 anchors lie in ONE sample (the launcher checks Hg * Wg % 32 == 0); its statistics are merged
    // from the producer's partials by wave 0 and every staged float4 becomes (v - mean) * rstd * gamma + beta, the expression of
    // ln_apply_kernel (bit-identical to the two-launch form).  Out-of-image taps load 0 for v, gamma and beta alike: they stay 0.
    constexpr bool in_ln = IN_LN;
    __shared__ float in_stat[2];
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_ln ? d.in_g : d.x0), 0, in_ln ? d.Hin * d.Win * d.c0 * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_ln ? d.in_b : d.x0), 0, in_ln ? d.Hin * d.Win * d.c0 * 4 : 0, 0x00020000);
    if (IN_LN && tid < 64) {
        float mean, rstd;
        ln_merge_partials(d.in_part, pivp_fdiv(m0, d.fd_hw_mul, d.fd_hw_sh), d.in_np, d.in_eps, mean, rstd);
        if (tid == 0) {
            in_stat[0] = mean; in_stat[1] = rstd;
            const int bs = pivp_fdiv(m0, d.fd_hw_mul, d.fd_hw_sh);
            if (d.in_stat_out && nblk == 0 && m0 == bs * HWg) { d.in_stat_out[bs * 2] = mean; d.in_stat_out[bs * 2 + 1] = rstd; }      // kept for the backward pass
        }
    }
    int l_cc = 0, l_ty = 0, l_tx = 0;
    // two register sets: chunk i+2 is loaded while chunk i is multiplied and written to LDS at the end of chunk i+1
    // NS register sets (round 6: 4, before 2): chunk c travels in set c % NS, is loaded NS chunks ahead of its multiplies and written to LDS one chunk
    // ahead.  With two sets a chunk's loads had ~1.5 chunk times (~700 cycles) to come back from L2 / MALL -- less than their latency under load: the loop
    // ran at ~930 cycles per chunk (enc2's 18 chunks against enc1's 9: +3.5 us) for ~340 cycles of instructions and 256 of MFMAs.
    constexpr int NS = 4;
    f32x4 ras[NS], rbs[NS][NTB];
    f32x4 rgs[NS], rbe[NS];            // in_ln: gamma / beta of the staged float4
    int roo[NS];                       // in_ln with d.in_out (training plans): where the staged float4, normalised, is ALSO written (float index, or -1).  Each
                                       // input pixel of a stride-2 3x3 conv is the tap (1..2, 1..2) of exactly one anchor: that anchor's thread owns it.
    // PART 0: the whole chunk; 1: its weight tiles only (no advance); 2: the rest, then advance -- the first chunk's weights go out before the anchor's
    // address arithmetic (a block's prologue is priced by the instructions in front of its first load: 244 of this kernel's 960 before round 6)
    auto load_next = [&](auto SET, auto PART) {   // chunk (l_ty, l_tx, l_cc) -> register set SET, then advance
        constexpr int part = decltype(PART)::value;
        f32x4& ra = ras[decltype(SET)::value];
        f32x4 (&rb)[NTB] = rbs[decltype(SET)::value];
        int dy, dx, wi;
        if (deconv) {
            const int ky = py ? 2 * l_ty : 1, kx = px ? 2 * l_tx : 1;
            dy = (py + 1 - ky) >> 1; dx = (px + 1 - kx) >> 1; wi = ky * 3 + kx;
        } else {
            dy = l_ty - d.pad; dx = l_tx - d.pad; wi = l_ty * d.ksize + l_tx;
        }
        const int ch = l_cc << 5;
        const bool first = ch < d.c0;
        const int ld = first ? d.ld0 : d.ld1;
        const int delta = ((dy * d.Win + dx) * ld + (first ? ch : ch - d.c0)) * 4;
        const int wbase = (wi * (d.wcin >> 5) + l_cc) * (d.wN ? d.wN : d.N) * 128;
        if constexpr (part != 1) {
            const int iy = a_iy0 + dy, ix = a_ix0 + dx;
            const bool ok = (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
            const unsigned off = ok ? (unsigned)((first ? a_off0 : a_off1) + delta) : OOB;
            ra = __builtin_bit_cast(f32x4, first ? __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0)
                                                 : __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0));
            if constexpr (IN_LN) {         // element (iy, ix, channel) of the sample: [Hin*Win][c0]
                const bool own = d.in_out && nblk == 0 && ok && l_ty >= 1 && l_tx >= 1 && l_ty < nty;      // (not the requests past the last chunk)
                roo[decltype(SET)::value] = own ? (a_pix0 + dy * d.Win + dx) * d.in_out_ld + ch + cvec * 4 : -1;
                const unsigned goff = ok ? (unsigned)(((iy * d.Win + ix) * d.c0 + ch + cvec * 4) * 4) : OOB;
                rgs[decltype(SET)::value] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsg, goff, 0, 0));
                rbe[decltype(SET)::value] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, goff, 0, 0));
            }
        }
        if constexpr (part != 2) {
#pragma unroll
            for (int t = 0; t < NTB; ++t) rb[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, b_goff[t], wbase, 0));
        }
        if constexpr (part != 1) { if (++l_cc == ncc) { l_cc = 0; if (++l_tx == ntx) { l_tx = 0; ++l_ty; } } }
    };
    auto store_regs = [&](auto SET, int buf) {
        f32x4 ra = ras[decltype(SET)::value];
        const f32x4 (&rb)[NTB] = rbs[decltype(SET)::value];
        if constexpr (IN_LN) {
            const float mean = in_stat[0], rstd = in_stat[1];
            const f32x4 g = rgs[decltype(SET)::value], be = rbe[decltype(SET)::value];
#pragma unroll
            for (int e = 0; e < 4; ++e) ra[e] = (ra[e] - mean) * rstd * g[e] + be[e];
            if (d.in_out && roo[decltype(SET)::value] >= 0) *reinterpret_cast<f32x4*>(d.in_out + roo[decltype(SET)::value]) = ra;
        }
        *reinterpret_cast<f32x4*>(lds + A_OFF + buf * TILE + lds_w) = ra;
#pragma unroll
        for (int t = 0; t < NTB; ++t) *reinterpret_cast<f32x4*>(lds + B_OFF + (buf * NTB + t) * TILE + lds_w) = rb[t];
    };

    if (nchunks > 0) load_next(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});      // the first chunk's weight tiles
    __builtin_amdgcn_sched_barrier(0);
    anchor();
    // FUSE3: this wave's operands of the 1x1 that follows -- its 16 output columns' weights (16 k-steps x 4 k) and the per-sample bias
    // b3 + W3s . (action, state) -- are requested here, a whole K loop ahead of their use.  Wave w owns output columns 16 w .. 16 w + 15.
    float f3w[FUSE3 ? 16 : 1], f3c = 0.f;
    if constexpr (FUSE3) {
        const int col = 16 * wave + (lane & 15), kg = lane >> 4, bs = pivp_fdiv(m0, d.fd_hw_mul, d.fd_hw_sh);
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) f3w[s4] = d.f3_w[(4 * s4 + kg) * 64 + col];
        f3c = d.f3_b[col];
        if (d.f3_use_state) {
#pragma unroll
            for (int j = 0; j < 10; ++j) f3c = fmaf(j < 5 ? d.f3_action[bs * 5 + j] : d.f3_state[bs * 5 + j - 5], d.f3_w[(64 + j) * 64 + col], f3c);
        }
    }
    f32x16 acc[NTB];
#pragma unroll
    for (int t = 0; t < NTB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int frag = l31 * SP + 8 * wave + 4 * half;   // k-group `wave` of row l31

    if (nchunks > 0) {
        using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
        using P0 = std::integral_constant<int, 0>; using P2 = std::integral_constant<int, 2>;
        auto mma = [&](int buf) {
            const f32x4 fa = *reinterpret_cast<const f32x4*>(lds + A_OFF + buf * TILE + frag);
            f32x4 fb[NTB];
#pragma unroll
            for (int t = 0; t < NTB; ++t) fb[t] = *reinterpret_cast<const f32x4*>(lds + B_OFF + (buf * NTB + t) * TILE + frag);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int t = 0; t < NTB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s2], fb[t][s2], acc[t], 0, 0, 0);
        };
        using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
        // (requests past the last chunk are made all the same: their tap is past the tap set, so the weight offset is outside the pack's descriptor --
        // zeros, no traffic -- and the activation load is an in-image read nobody uses.  A run-time condition around a request makes hipcc's wait-count
        // pass assume nothing about the loads in flight: s_waitcnt vmcnt(0) in front of every LDS write, i.e. no prefetch at all.)
        load_next(S0{}, P2{});
        load_next(S1{}, P0{});
        load_next(S2{}, P0{});
        load_next(S3{}, P0{});
        if constexpr (IN_LN) __syncthreads();   // in_stat
        store_regs(S0{}, 0);
        __syncthreads();
        // chunk c: LDS buffer c & 1, register set c % 4.  Its trip: request chunk c + 4 into the set chunk c left when it was written to LDS, multiply,
        // write chunk c + 1 into the other buffer (every wave left that buffer's chunk c - 1 at the last barrier), barrier.
        // (no run-time condition inside a trip -- the write of a chunk past the end puts zeros into the buffer nobody reads again -- and none between the four
        // trips of the steady loop: every wait in it is a counted vmcnt)
        auto trip = [&](auto SET, auto NEXT) {
            constexpr int buf = decltype(SET)::value & 1;
            load_next(SET, P0{});
            mma(buf);
            store_regs(NEXT, buf ^ 1);
            __syncthreads();
        };
        int it = 0;
        for (; it + NS <= nchunks; it += NS) { trip(S0{}, S1{}); trip(S1{}, S2{}); trip(S2{}, S3{}); trip(S3{}, S0{}); }
        if (it < nchunks) trip(S0{}, S1{});
        if (it + 1 < nchunks) trip(S1{}, S2{});
        if (it + 2 < nchunks) trip(S2{}, S3{});
    }

    // sum of the four waves' partial tiles, then the epilogue on float4 rows
#pragma unroll
    for (int t = 0; t < NTB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
            lds[(wave * 32 + m) * RP + t * 32 + l31] = acc[t][r];
        }
    __syncthreads();
    const int m = m0 + prow;
    const bool valid = m < d.M;
    const int b = valid ? pivp_fdiv(m, d.fd_hw_mul, d.fd_hw_sh) : 0, rem = m - b * HWg;
    const int ay = pivp_fdiv(rem, d.fd_w_mul, d.fd_w_sh), ax = rem - ay * d.Wg;
    const int oy0 = deconv ? py : 0, ox0 = deconv ? px : 0;
    float* orow = d.out + ((size_t)(b * d.Hout + ay * d.out_step + oy0) * d.Wout + ax * d.out_step + ox0) * d.ldo + nblk * BN;
    float sv[4 * NTB];   // this thread's outputs, for the fused LayerNorm partial
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
        const int cl = t * 32 + cvec * 4;   // column within the block tile
        f32x4 v = *reinterpret_cast<const f32x4*>(lds + prow * RP + cl);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(lds + (w * 32 + prow) * RP + cl);
        if (d.bias) v += *reinterpret_cast<const f32x4*>(d.bias + nblk * BN + cl);
        if (d.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        if (valid) {
            if (d.accum) v += *reinterpret_cast<const f32x4*>(orow + cl);
            *reinterpret_cast<f32x4*>(orow + cl) = v;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) sv[t * 4 + e] = v[e];
    }
    if constexpr (FUSE3) {
        // e3 = relu(f3c + tile . W3x): [32 px x 64] x [64 x 64] on v_mfma_f32_16x16x4_f32, wave w its 16 output columns for both 16-pixel row tiles.
        // C = the bias, instruction s fed k = 4 s + lane group: the fmaf chain of enc3_state_kernel in its order (bias, smear terms, k ascending): bit-identical.
        __syncthreads();                             // the partial-sum image is dead: the tile (after bias / ReLU) takes its place, [32][68]
#pragma unroll
        for (int t = 0; t < NTB; ++t)
            *reinterpret_cast<f32x4*>(lds + prow * 68 + t * 32 + cvec * 4) = f32x4{sv[t * 4], sv[t * 4 + 1], sv[t * 4 + 2], sv[t * 4 + 3]};
        __syncthreads();
        const int li = lane & 15, kg = lane >> 4;
        f32x4 c3[2];
        c3[0] = f32x4{f3c, f3c, f3c, f3c}; c3[1] = c3[0];
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                c3[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[(16 * mt + li) * 68 + 4 * s4 + kg], f3w[s4], c3[mt], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mm = m0 + 16 * mt + 4 * kg + r;       // accumulator register r: pixel row 4 (lane >> 4) + r of the 16-row tile, column lane & 15
                if (mm < d.M) d.f3_out[(size_t)mm * 64 + 16 * wave + li] = fmaxf(c3[mt][r], 0.f);
            }
        const int bs3 = pivp_fdiv(m0, d.fd_hw_mul, d.fd_hw_sh);
        if (d.f3_state_out && m0 == bs3 * HWg && tid < 5) {      // current_state = Linear(state_action) (TM:730): once per sample
            const int bs = bs3;
            float v = d.f3_bcs[tid];
#pragma unroll
            for (int j = 0; j < 10; ++j) v = fmaf(d.f3_wcs[tid * 10 + j], j < 5 ? d.f3_action[bs * 5 + j] : d.f3_state[bs * 5 + j - 5], v);
            d.f3_state_out[bs * 5 + tid] = v;
        }
    }
    if (d.ln_part) {   // (count, mean, M2) of the block's outputs, two passes over registers, fixed summation order
        __syncthreads();   // the partial-sum image is dead
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 4 * NTB; ++i) s1 += valid ? sv[i] : 0.f;
        s1 = wave_sum(s1);
        const float c1 = wave_sum(valid ? (float)(4 * NTB) : 0.f);
        if (lane == 0) { lds[wave] = s1; lds[4 + wave] = c1; }
        __syncthreads();
        const float cnt = (lds[4] + lds[5]) + (lds[6] + lds[7]);
        const float mean = ((lds[0] + lds[1]) + (lds[2] + lds[3])) / cnt;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4 * NTB; ++i) { const float dd = sv[i] - mean; q = valid ? fmaf(dd, dd, q) : q; }
        q = wave_sum(q);
        if (lane == 0) lds[8 + wave] = q;
        __syncthreads();
        if (tid == 0) {
            const int bb = pivp_fdiv(m0, d.fd_hw_mul, d.fd_hw_sh);
            float* pp = d.ln_part + ((size_t)bb * d.ln_nparts + (((m0 - bb * HWg) >> 5) * n_nblk + nblk) * gdy + phase) * 4;
            pp[0] = cnt; pp[1] = mean; pp[2] = (lds[8] + lds[9]) + (lds[10] + lds[11]); pp[3] = 0.f;
        }
    }
}

// d has been validated by igemm_validate (igemm_f32.hip); additionally needs 16-B aligned out / bias rows.
int igemm_small(const IgemmDesc& d, hipStream_t stream, int* ln_nparts) {
    PIVP_CHECK_ARG(d.ldo % 4 == 0 && ((uintptr_t)d.out & 15) == 0 && (!d.bias || ((uintptr_t)d.bias & 15) == 0));
    const int mblk = (d.M + 31) / 32, nt = d.N / 32;
    // widest column block that still leaves >= 4 blocks per CU (enc4: 1 tile 14.2 us vs 2 tiles 15.3; enc5: 3 tiles 23.8 vs 1 tile
    // 25.7; enc6: 2 tiles 39.5 vs 1 tile 44.9)
    int ntb = 1;
    if (nt % 3 == 0 && (long)mblk * (nt / 3) * d.nphase >= 1024) ntb = 3;
    else if (nt % 2 == 0 && (long)mblk * (nt / 2) * d.nphase >= 1024) ntb = 2;
    if (d.f3_out) {      // the fused 1x1 behind this conv: 64 columns in one block, tiles inside one sample, LayerNorm-on-load form only
        PIVP_CHECK_ARG(d.in_g && nt == 2 && d.nphase == 1 && (d.Hg * d.Wg) % 32 == 0 && !d.accum && !d.ln_part);
        PIVP_CHECK_ARG(d.f3_w && d.f3_b && d.f3_action && d.f3_state && (!d.f3_state_out || (d.f3_wcs && d.f3_bcs)));
        ntb = 2;
    }
    dim3 grid(mblk * (nt / ntb), d.nphase);
    IgemmDesc dd = d;   // fused LayerNorm partials: one per block, when no tile straddles two samples
    const int hwg = d.Hg * d.Wg;
    const int np = (hwg / 32) * (nt / ntb) * d.nphase;
    dd.ln_nparts = (d.ln_part && hwg % 32 == 0 && np <= d.ln_cap) ? np : 0;
    if (!dd.ln_nparts) dd.ln_part = nullptr;
    if (ln_nparts) *ln_nparts = dd.ln_nparts;
    dd.n_mblk = nt / ntb;                                        // (here: the column blocks, the divisor of blockIdx.x)
    pivp_fastdiv((unsigned)(nt / ntb), &dd.fd_mb_mul, &dd.fd_mb_sh);
    pivp_fastdiv((unsigned)hwg, &dd.fd_hw_mul, &dd.fd_hw_sh);
    pivp_fastdiv((unsigned)d.Wg, &dd.fd_w_mul, &dd.fd_w_sh);
    if (d.in_g) {      // LayerNorm-on-load (see IgemmDesc::in_g): one sample per tile, one source, a plain conv
        PIVP_CHECK_ARG(igemm_in_ln_ok(d) && d.in_b && d.in_part && d.in_np > 0);
        // the write-back of the normalised input (training plans): every input pixel must be the tap (1..2, 1..2) of one anchor -- the 3x3 stride-2 pad-1 conv
        PIVP_CHECK_ARG(!d.in_out || (d.ksize == 3 && d.pad == 1 && d.in_step == 2 && d.Hin == 2 * d.Hg && d.Win == 2 * d.Wg && d.in_out_ld >= d.c0 && d.in_out_ld % 4 == 0 &&
                                     ((uintptr_t)d.in_out & 15) == 0 && (long long)d.B * d.Hin * d.Win * d.in_out_ld < (1LL << 31)));
        if (d.f3_out) { hipLaunchKernelGGL((igemm_small_kernel<2, true, true>), grid, dim3(256), 0, stream, dd); return PIVP_LAUNCH_STATUS(); }
        if (ntb == 3) hipLaunchKernelGGL((igemm_small_kernel<3, true>), grid, dim3(256), 0, stream, dd);
        else if (ntb == 2) hipLaunchKernelGGL((igemm_small_kernel<2, true>), grid, dim3(256), 0, stream, dd);
        else hipLaunchKernelGGL((igemm_small_kernel<1, true>), grid, dim3(256), 0, stream, dd);
        return PIVP_LAUNCH_STATUS();
    }
    if (ntb == 3) hipLaunchKernelGGL(igemm_small_kernel<3>, grid, dim3(256), 0, stream, dd);
    else if (ntb == 2) hipLaunchKernelGGL(igemm_small_kernel<2>, grid, dim3(256), 0, stream, dd);
    else hipLaunchKernelGGL(igemm_small_kernel<1>, grid, dim3(256), 0, stream, dd);
    return PIVP_LAUNCH_STATUS();
}

// geometry the LayerNorm-on-load form serves: a conv (not the 4-phase transposed form) on one source whose rows are exactly the c0
// channels, output tiles of 32 anchors inside one sample
bool igemm_in_ln_ok(const IgemmDesc& d) {
    return !d.deconv && d.nphase == 1 && d.c1 == 0 && d.ld0 == d.c0 && (d.Hg * d.Wg) % 32 == 0 && d.M == d.B * d.Hg * d.Wg &&
           (long long)d.Hin * d.Win * d.c0 * 4 < (1LL << 31);
}

}  // namespace pivp

PIVP_DEFINE_MAIN_PRIO_SETTER(igemm_small)
