// Implicit-GEMM convolution on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One kernel template serves every dense contraction of the per-timestep program
// (reference call sites: src/models/train_model.py, "TM"):
//   * the seven ConvLSTM 5x5 gate convolutions with the gate math fused into the epilogue
//     (TM:262-272: concat(x,h) -> conv5x5 -> split j,i,f,o -> c,h update)      [95% of FLOPs]
//   * the 3x3 stride-2 convolutions enc1/enc2 (TM:501-502)
//   * the 3x3 stride-2 transposed convolutions enc4/enc5/enc6 (TM:505-507), run as their four
//     sub-pixel phases (1/2/2/4 taps) so no zero-stuffed input is ever touched.
//
// GEMM view: M = anchors (b, ay, ax) of an anchor grid, N = output channels, K = taps x Cin.
// A[m][k] is gathered on the fly from NHWC activations (zero outside the image); B is the weight
// matrix, pre-packed K-inner as [tap][Cin/32][N][32] so that a 32-deep K chunk of any column is
// 128 contiguous bytes.  fp32 in / fp32 accumulate: the MFMA result is bit-for-bit a k-ordered
// fmaf chain, so parity with the fp32/fp64 oracle is an ordering question only.
//
// Tiling: 256 threads = 4 waves arranged WM x WN.  A wave owns 32 anchors x (32/WN channels x all
// of its gates/columns); a block owns BM = 32*WM anchors x BN columns.  K is consumed in chunks of
// one tap x 32 input channels, staged global -> registers -> LDS (two LDS buffers, one barrier per
// chunk): the buffer loads of chunk i+1 are issued at the top of chunk i and written to LDS at
// its end.
//   A tile [BM][36] and B tile [BN][36] floats: 144-B rows make every per-lane ds_read_b128
//   (4 consecutive k of one row) conflict-free within its 16-lane group.
// What it took to keep ONE wave per SIMD (all the parallelism a B = 32 rollout offers) on the
// matrix pipe, in the order the measurements found it (profiles/r01/NOTES.md):
//   * operand fragments by ds_read_b128 from K-inner tiles (5 LDS reads per 16 MFMAs);
//   * no table lookups in the loop: taps are decoded with scalar counters (a kernarg byte-array
//     lookup compiled to a VECTOR load + vmcnt(0), i.e. two memory round trips per chunk);
//   * raw buffer loads, out-of-image taps get an out-of-range offset and the hardware returns 0:
//     no branches, 32-bit offsets, and the whole chunk body is ONE basic block;
//   * that block's staging instructions are spread over the MFMA gaps (sched_group_barrier).
// ConvLSTM: the block's 128 columns are the 4 gates (j,i,f,o) of 32 channels.  With WN = 1 a wave's
// four accumulators are the four gates of the same channels; with WN = 2 / 4 a 32-column tile holds
// 2 / 4 gates of 16 / 8 channels and the epilogue gathers the four gates of a channel with wave
// shuffles.  The 4C-wide gate tensor is never written.  NTB = 2 is the square 64 x 64 tile: 4 gates of 16
// channels per block, 2 x 2 waves, each wave tile the 4 gates of 8 channels (GPT = gates per wave tile = 4).
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>

#include "pivp_kernels.h"

namespace pivp {

constexpr int IG_P = 36;  // LDS row pitch in floats (A and B tiles)

template <int WM, int NTB, int KG = 1>
constexpr int ig_lds_bytes() { return KG * 2 * (32 * WM + 32 * NTB) * IG_P * 4; }

// sigmoid / tanh of the gate epilogue through v_exp_f32 / v_rcp_f32: |error| <= ~2e-7 absolute, an order
// below the fp32 accumulation noise of the K = 1600..4800 dot products in front of them.
// 1 / d as v_rcp_f32 plus one Newton step (3 instructions, error well under 1 ulp) instead of an IEEE division: __frcp_rn expands to
// ~10 instructions (div_scale x2, rcp, 4 fma, div_fmas, div_fixup), 5 times per cell.  The bare 1-ulp v_rcp_f32 is not enough for the
// parity path: through 9 recurrent steps and the STP warp of white-noise frames it moved tests/test_gpu_model.py's STP case past its gate.
__device__ __forceinline__ float fast_rcp(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.0f), r, r);
}
// (x is clamped at -87: below that exp(-x) is +inf, v_rcp_f32 returns 0 and the Newton step's inf * 0 is a NaN -- found in round 4 by a range test
// of the split-precision kernels with pre-activations of -150, where this kernel wrote NaN cells; sigmoid(-87) = 1.6e-38)
__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.0f + __expf(-fmaxf(x, -87.0f))); }
// tanh: 2 / (1 + exp(-2x)) - 1 carries ~1.2e-7 of ABSOLUTE error at every x (the rounding of a value near 1), i.e. many ulps of a small
// tanh; behind a LayerNorm that divides by the ~0.2 spread of h this was a third of the whole path's error against the float64 oracle
// (scripts/gate_math_study.py: hidden1 3.7e-7 rms with libm's tanh, 4.7e-7 with that formula) and what put the STP fixture over 1e-4.
// Here: |x| < 0.55: x + x^3 P(x^2), a degree-4 least-squares fit (3.3e-8 absolute, < 1 ulp relative); beyond: 1 - 2 / (1 + exp(2|x|)).
__device__ __forceinline__ float fast_tanh(float x) {
    const float a = fminf(fabsf(x), 15.0f);   // tanh(15) rounds to 1; keeps exp finite for the Newton step of fast_rcp
    const float p = a * a;
    float q = fmaf(p, -0.006149096414446831f, 0.02097311243414879f);
    q = fmaf(p, q, -0.053824927657842636f);
    q = fmaf(p, q, 0.13332274556159973f);
    q = fmaf(p, q, -0.3333330452442169f);
    const float small = fmaf(a * p, q, a);
    const float big = fmaf(-2.0f, fast_rcp(1.0f + __expf(2.0f * a)), 1.0f);
    return copysignf(a < 0.55f ? small : big, x);
}

#ifdef PIVP_F32_STAMPS   // per-block phase stamps (constant-rate 100 MHz counter) of the kernel: scripts/f32_stamps.py
// second half of the array: the shader-cycle counter (s_memtime) at the same points; cycles / wall = the clock the chip holds in that phase
__device__ long long pivp_f32_stamps[2048 * 8];
#define F32_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 2048 && blockIdx.y == 0 && blockIdx.z == 0) { \
    pivp_f32_stamps[blockIdx.x * 4 + (i)] = (long long)wall_clock64(); pivp_f32_stamps[2048 * 4 + blockIdx.x * 4 + (i)] = (long long)clock64(); } } while (0)
#else
#define F32_STAMP(i)
#endif

// (Rounds 1-4 carried timing-only ablations of this kernel -- no loads / no LDS stores / no barrier / no address math -- as a template parameter;
// their numbers are in profiles/r01 .. r03/NOTES.md, the code in the history.)
// KG = 2 (ConvLSTM on small maps, lstm5's 8 x 8): the block is TWO groups of 4 waves that each run HALF of the K chunks of the same
// output tile through their own LDS buffers, in lockstep (same barriers); group 1 then hands its accumulators to group 0 through LDS
// and group 0 runs the epilogue.  With M = 2048 anchors there are only 256 tiles of 32 x 128: one block per CU, one wave per SIMD,
// MFMA pipe busy 0.62.  Measured: lstm5 110.6 -> 117 TFLOP/s only (rollout 8.69 -> 8.66 ms): the tile is not short of waves but of L2
// bandwidth -- 256 blocks x 150 chunks x (4 KB of A + 16 KB of B) = 614 MB per launch in ~80 us, 7.7 TB/s; a 32-row tile uses every
// weight byte for 32 MACs.  A square 64 x 64 tile (16 channels x 4 gates) would move 20 % less; not built.
template <int WM, int WN, int NTB, bool LSTM, int KG = 1>
__global__ __launch_bounds__(256 * KG, (KG == 1 && LSTM && WM < 4) ? 2 : 1) void igemm_f32_kernel(const IgemmDesc d) {
    F32_STAMP(0);
    PIVP_SET_MAIN_PRIO();
    static_assert(WM * WN == 4, "4 waves");
    static_assert(KG == 1 || (KG == 2 && LSTM), "the in-block K split serves the ConvLSTM tile only");
    static_assert(!LSTM || NTB == 4, "ConvLSTM blocks own 4 gates x 32 channels");
    constexpr int BM = 32 * WM;
    constexpr int BN = 32 * NTB;
    constexpr int CB = 8 * NTB;   // LSTM: channels per block (its BN columns are the 4 gates of CB channels): 32, or 16 for the square 64 x 64 tile
    constexpr int CPW = LSTM ? CB / WN : 32 / WN;  // LSTM: channels per wave
    constexpr int GPT = 32 / CPW; // LSTM: gates inside one 32-column MFMA tile of a wave (1, 2 or 4); with CB = 32 this is WN
    constexpr int TPW = LSTM ? 4 / GPT : NTB / WN;
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    constexpr int A_FLOATS = BM * IG_P, B_FLOATS = BN * IG_P;
    const int gid = KG > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;     // K group of this wave
    float* const lds = lds_all + gid * 2 * (A_FLOATS + B_FLOATS);                           // the group's own A / B buffers

    const int tid = KG > 1 ? (int)threadIdx.x & 255 : (int)threadIdx.x;                     // thread index inside the group
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int phase = (int)gridDim.y - 1 - (int)blockIdx.y;   // transposed conv: the 4-tap phase is dispatched first, the 1-tap phase fills the tail
    const int n_nblk = LSTM ? (d.C / CB) : (d.N / BN);
    // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (each with its own 4 MB L2), so block b runs
    // on XCD b % 8.  Give XCD k the k-th CONTIGUOUS eighth of the logical tile list, ordered column-block-major: its L2
    // then holds one column block's weights (<= 2.5 MB for every layer here) and a contiguous band of anchors whose
    // 5x5 halos overlap each other, instead of every XCD streaming all weights and all of the image.
    int lid = blockIdx.x;
    if ((gridDim.x & 7) == 0) lid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int n_mblk = d.n_mblk;                       // gridDim.x / n_nblk, and the multiplier / shift that divide by it (launch_igemm)
    const int nblk = pivp_fdiv(lid, d.fd_mb_mul, d.fd_mb_sh);
    const int mblk = lid - nblk * n_mblk;
    const int m0 = mblk * BM;
    const int cin = d.c0 + d.c1;
    const int ncc = cin >> 5;
    const int py = phase >> 1, px = phase & 1;
    // tap set: conv K x K (rows ky, cols kx) or, for the transposed 3x3 s2 conv, the taps of output parity
    // (py, px): oy = 2*iy - 1 + ky  =>  ky = 1 (py = 0) or ky in {0, 2} (py = 1), iy = a + (py + 1 - ky)/2.
    const bool deconv = LSTM ? false : d.deconv != 0;
    const int ksz = LSTM ? 5 : d.ksize, pad = LSTM ? 2 : d.pad;      // the gate convolution is 5 x 5, pad 2 (igemm_validate): constants there
    const int nty_all = deconv ? 1 + py : ksz;
    const int ntx = deconv ? 1 + px : ksz;
    // optional split of K over blockIdx.z (data gradients of small-M layers): contiguous ranges of the chunk sequence
    // (kernel row, kernel column, 32-channel chunk), equal to within one chunk; partial sums are atomically added into a pre-zeroed output
    const int ksplit = LSTM ? 1 : (int)gridDim.z;                      // (the gate kernels never split K over the grid)
    const int nchunks_tot = nty_all * ntx * ncc;
    const int z_begin = ksplit > 1 ? (int)((unsigned)(nchunks_tot * (int)blockIdx.z) / (unsigned)ksplit) : 0;          // (<= 25 * 32 chunks x 64 splits: 32 bits)
    const int z_end = ksplit > 1 ? (int)((unsigned)(nchunks_tot * ((int)blockIdx.z + 1)) / (unsigned)ksplit) : nchunks_tot;
    const int nchunks_all = z_end - z_begin;
    const int nchunks = nchunks_all / KG;               // (KG = 2: the launcher only takes this form for an even chunk count)
    const int chunk0 = z_begin + gid * nchunks;         // this block's / group's first chunk
    const int HWg = d.Hg * d.Wg;

    // ---- staging roles ---------------------------------------------------------------------
    constexpr int NA = (BM * 8 + 255) / 256;  // float4 per thread for the A tile
    constexpr int NB = BN * 8 / 256;          // float4 per thread for the B tile
    constexpr unsigned OOB = 0xC0000000u;     // beyond num_records of any descriptor: the load returns 0
    const int cvec = tid & 7;                 // float4 within the 32-float chunk row
    const int prow = tid >> 3;                // 0..31
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x0), 0, d.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.c1 ? d.x1 : d.x0), 0, d.c1 ? d.bytes1 : d.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, d.bytesw, 0x00020000);
    int a_off0[NA], a_off1[NA];   // anchor pixel's byte offset in either source
    unsigned a_mask[NA];          // bit (ty * 5 + tx): tap (ty, tx) of this anchor reads inside the image (ksize <= 5)
    auto anchors = [&]() {        // (called in the prologue, behind the first chunk's weight loads)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int m = m0 + prow + 32 * j;
        if (m < d.M && prow + 32 * j < BM) {
            const int b = pivp_fdiv(m, d.fd_hw_mul, d.fd_hw_sh);
            const int rem = m - b * HWg;
            const int ay = pivp_fdiv(rem, d.fd_w_mul, d.fd_w_sh);
            const int ax = rem - ay * d.Wg;
            const int iy0 = ay * d.in_step, ix0 = ax * d.in_step;
            const int pix = b * d.Hin * d.Win + iy0 * d.Win + ix0;
            a_off0[j] = (pix * d.ld0 + cvec * 4) * 4;
            a_off1[j] = (pix * d.ld1 + cvec * 4) * 4;
            // validity is separable: row ty is inside for all columns or none.  Two masks here replace two adds, two
            // compares and an s_and per gathered piece and chunk in the loop.
            unsigned colm = 0, m25 = 0;
            if (deconv) {
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const int dxx = (px + 1 - (px ? 2 * t : 1)) >> 1;
                    if (t < ntx && (unsigned)(ix0 + dxx) < (unsigned)d.Win) colm |= 1u << t;
                }
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const int dyy = (py + 1 - (py ? 2 * t : 1)) >> 1;
                    if (t < nty_all && (unsigned)(iy0 + dyy) < (unsigned)d.Hin) m25 |= colm << (5 * t);
                }
            } else {
                // tap t of a row / column is inside iff pad - x0 <= t <= extent - 1 + pad - x0: one contiguous run of bits per axis; the rows' run is
                // spread to bits 0, 5, .., 20 and multiplied with the columns' 5 bits (no carries: the products do not overlap)
                auto run = [](int lo, int hi) -> unsigned { return hi >= lo ? ((2u << hi) - 1u) & ~((1u << lo) - 1u) : 0u; };
                colm = run(max(pad - ix0, 0), min(d.Win - 1 + pad - ix0, ntx - 1));
                const int rlo = max(pad - iy0, 0), rhi = min(d.Hin - 1 + pad - iy0, nty_all - 1);
                const unsigned rows = rhi >= rlo ? 0x108421u & ((2u << (5 * rhi)) - 1u) & ~((1u << (5 * rlo)) - 1u) : 0u;
                m25 = colm * rows;
            }
            a_mask[j] = m25;
        } else {
            a_off0[j] = a_off1[j] = 0;
            a_mask[j] = 0;            // never in range
        }
    }
    };
    int b_goff[NB];                           // byte offset of this thread's float4 inside a [N][32] weight chunk
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int r = prow + 32 * j;
        const int col = LSTM ? (r / CB) * d.C + nblk * CB + (r % CB) : nblk * BN + r;
        b_goff[j] = (col * 32 + cvec * 4) * 4;
    }
    const int lds_wa = (prow * IG_P + cvec * 4);          // this thread's write slot inside the A / B tile (floats)

    // two register sets: the tiles of chunk i+2 are loaded during chunk i and written to LDS during chunk i+1 (under load an
    // L2 / MALL round trip is longer than one chunk: profiles/r01/NOTES.md)
    f32x4 ras[2][NA];
    f32x4 rbs[2][NB];

    // scalar state of the NEXT chunk to load: channel chunk l_cc of tap (l_ty, l_tx); no division, no branch
    int l_cc = 0, l_ty = 0, l_tx = 0;
    if (KG > 1 || ksplit > 1) {
        const int tap = pivp_fdiv(chunk0, d.fd_cc_mul, d.fd_cc_sh);
        l_cc = chunk0 - tap * ncc; l_ty = LSTM ? tap / 5 : tap / ntx; l_tx = tap - l_ty * ntx;
    }
    int s_delta = 0, s_ld = 0, s_wbase = 0;
    unsigned s_bit = 0;
    bool s_first = true;
    // scalar prelude of one chunk's loads in three parts (tap decode; activation offset; weight offset + counters): inside the K loop they
    // run in three consecutive micro-steps that carry no staging piece, so that no MFMA waits behind the whole ~25-instruction chain of
    // dependent scalar multiplies.  (The timing-only build with the address math alone is 4.8 us slower on lstm1's 102; spreading the
    // chain returned 0.3 % of the seven layers, same box: 798.7 / 801.0 -> 797.0 / 797.9 us -- the other wave of the SIMD already filled it.)
    int t_dy = 0, t_dx = 0, t_wi = 0, t_cbase = 0;
    auto stage_a = [&]() {
        if (deconv) {
            const int ky = py ? 2 * l_ty : 1, kx = px ? 2 * l_tx : 1;
            t_dy = (py + 1 - ky) >> 1; t_dx = (px + 1 - kx) >> 1; t_wi = ky * 3 + kx;
        } else {
            t_dy = l_ty - pad; t_dx = l_tx - pad; t_wi = l_ty * ksz + l_tx;
        }
        s_bit = __builtin_amdgcn_readfirstlane(1u << (l_ty * 5 + l_tx));
        const int ch = l_cc << 5;
        s_first = ch < d.c0;
        s_ld = s_first ? d.ld0 : d.ld1;
        t_cbase = s_first ? ch : ch - d.c0;
    };
    auto stage_b = [&]() {
        // (readfirstlane: carried from one chunk to the next these values lose their "uniform" proof, and a divergent
        // soffset turns every weight load into a waterfall loop)
        s_delta = __builtin_amdgcn_readfirstlane(((t_dy * d.Win + t_dx) * s_ld + t_cbase) * 4);   // bytes, relative to the anchor pixel
    };
    auto stage_c = [&]() {
        s_wbase = __builtin_amdgcn_readfirstlane((t_wi * (d.wcin >> 5) + l_cc) * (d.wN ? d.wN : d.N) * 128);       // bytes; the weight keeps all its Cin chunks
        ++l_cc;
        const bool w0 = l_cc == ncc;
        l_cc = w0 ? 0 : l_cc;
        l_tx += w0 ? 1 : 0;
        const bool w1 = l_tx == ntx;
        l_tx = w1 ? 0 : l_tx;
        l_ty += w1 ? 1 : 0;
    };
    auto stage_begin = [&]() { stage_a(); stage_b(); stage_c(); };
    auto load_piece = [&](auto SET, auto J) {        // piece j < NA: A-tile load j; NA <= j < NA+NB: B-tile load j-NA
        constexpr int j = decltype(J)::value;
        f32x4 (&ra)[NA] = ras[decltype(SET)::value];
        f32x4 (&rb)[NB] = rbs[decltype(SET)::value];
        if constexpr (j < NA) {
            const bool ok = (a_mask[j] & s_bit) != 0;
            unsigned off = ok ? (unsigned)((s_first ? a_off0[j] : a_off1[j]) + s_delta) : OOB;
            ra[j] = __builtin_bit_cast(f32x4, s_first ? __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0)
                                                      : __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0));
        } else if constexpr (j < NA + NB) {
            rb[j - NA] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, b_goff[j - NA], s_wbase, 0));
        }
    };
    auto store_piece = [&](auto SET, auto J, int buf) {
        constexpr int j = decltype(J)::value;
        const f32x4 (&ra)[NA] = ras[decltype(SET)::value];
        const f32x4 (&rb)[NB] = rbs[decltype(SET)::value];
        if constexpr (j < NA) {
            if (BM >= 32 * (j + 1)) *reinterpret_cast<f32x4*>(lds + buf * A_FLOATS + lds_wa + 32 * j * IG_P) = ra[j];
        } else if constexpr (j < NA + NB) {
            *reinterpret_cast<f32x4*>(lds + 2 * A_FLOATS + buf * B_FLOATS + lds_wa + 32 * (j - NA) * IG_P) = rb[j - NA];
        }
    };

    // ConvLSTM: the K = 1600..4800 sum of an output runs as NACC interleaved chains (chunk parity, and for the narrow tiles also the
    // parity of the 8-channel k-group inside a chunk), added pairwise in front of the gate math.  One fp32 chain that long carried
    // ~1.25x the rounding error of the blocked BLAS sum the reference's NumPy path uses; the STP warp of white-noise frames turns
    // that into pixels (tests/test_gpu_model.py: stp_b2_t4 at the 1e-4 gate).  Costs 16 accumulator registers per extra chain.
    constexpr int NACC = LSTM ? (TPW == 1 ? 4 : TPW == 2 ? 2 : 1) : 1;
    f32x16 accs[NACC][TPW];       // (cleared in the prologue, while the first chunk's loads are in flight)
    f32x16 (&acc)[TPW] = accs[0];

    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int a_off = (wm * 32 + l31) * IG_P + 4 * half;
    int b_off[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        int row;
        if (LSTM) row = (t * GPT + l31 / CPW) * CB + wn * CPW + (l31 % CPW);   // gate-major rows of the block tile
        else      row = (wn * TPW + t) * 32 + l31;
        b_off[t] = 2 * A_FLOATS + row * IG_P + 4 * half;
    }

    // One chunk = 16 micro-steps (k-group q, k-step s) of TPW MFMAs each.  The fragments of k-group q+1 are read
    // from LDS ahead of group q's MFMAs; when STAGE, micro-step i < NA+NB is followed by the i-th buffer load of
    // the NEXT chunk and micro-step 16-(NA+NB)+i by its ds_write, so no stretch without an MFMA is longer than
    // one load's address arithmetic.  sched_barrier(0) pins this order (left alone, hipcc sinks the loads to
    // just above their ds_writes and exposes the whole L2 round trip every chunk).
    constexpr int NST = NA + NB;
    static_assert(NST <= 8, "staging pieces must fit the 16 micro-steps twice");
    // The scalar prelude of a chunk's loads (~25 SALU) only has one MFMA's shadow to hide in (issue is in order), so it
    // runs in a micro-step that carries no staging piece when there is one (step NST, for the chunk AFTER the one whose
    // loads follow); tiles that use all 8+8 staging steps keep it in front of the first load.
    constexpr int PRE = NST < 8 ? NST : 0;
    // (A barrier after micro-step 11 with the next chunk's first fragments read behind it was measured and made no
    // difference on any layer: the barrier stays at the chunk's end.)
    f32x4 fa[2];
    f32x4 fb[2][TPW];
    auto read_frag0 = [&](int slot) {
        fa[0] = *reinterpret_cast<const f32x4*>(lds + slot * A_FLOATS + a_off);
#pragma unroll
        for (int t = 0; t < TPW; ++t) fb[0][t] = *reinterpret_cast<const f32x4*>(lds + slot * B_FLOATS + b_off[t]);
    };
    // LOAD: this chunk issues the loads of chunk +2 into register set SET; STORE: it writes set SET^1 (chunk +1) to the other
    // LDS buffer
    auto chunk = [&](auto LOAD, auto STORE, auto SET, auto PAR, int buf) {   // PAR: parity of the chunk's index (accumulator chain)
        constexpr bool do_load = decltype(LOAD)::value;
        constexpr bool do_store = decltype(STORE)::value;
        using OTHER = std::integral_constant<int, decltype(SET)::value ^ 1>;
        const float* As = lds + buf * A_FLOATS + a_off;
        const float* Bs = lds + buf * B_FLOATS;
        read_frag0(buf);
        __builtin_amdgcn_sched_barrier(0);
        auto micro = [&](auto Q, auto S2) {
            constexpr int q = decltype(Q)::value, s2 = decltype(S2)::value, step = q * 4 + s2;
            constexpr int cur = q & 1, nxt = cur ^ 1;
            constexpr int ci = NACC == 1 ? 0 : NACC == 2 ? decltype(PAR)::value : 2 * decltype(PAR)::value + (q & 1);
            if constexpr (s2 == 0 && q < 3) {
                fa[nxt] = *reinterpret_cast<const f32x4*>(As + 8 * (q + 1));
#pragma unroll
                for (int t = 0; t < TPW; ++t) fb[nxt][t] = *reinterpret_cast<const f32x4*>(Bs + b_off[t] + 8 * (q + 1));
            }
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                accs[ci][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s2], fb[cur][t][s2], accs[ci][t], 0, 0, 0);
            if constexpr (do_load && PRE == 0 && step == 0) stage_begin();
            if constexpr (do_load && PRE > 0 && step == PRE) stage_a();
            if constexpr (do_load && PRE > 0 && step == PRE + 1) stage_b();
            if constexpr (do_load && PRE > 0 && step == PRE + 2) stage_c();
            if constexpr (do_load && step < NST) load_piece(SET, std::integral_constant<int, step>{});
            if constexpr (do_store && step >= 16 - NST) store_piece(OTHER{}, std::integral_constant<int, step - (16 - NST)>{}, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto qgroup = [&](auto Q) {
            micro(Q, std::integral_constant<int, 0>{}); micro(Q, std::integral_constant<int, 1>{});
            micro(Q, std::integral_constant<int, 2>{}); micro(Q, std::integral_constant<int, 3>{});
        };
        qgroup(std::integral_constant<int, 0>{}); qgroup(std::integral_constant<int, 1>{});
        qgroup(std::integral_constant<int, 2>{}); qgroup(std::integral_constant<int, 3>{});
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    auto load_all = [&](auto SET) {
        load_piece(SET, std::integral_constant<int, 0>{}); load_piece(SET, std::integral_constant<int, 1>{});
        load_piece(SET, std::integral_constant<int, 2>{}); load_piece(SET, std::integral_constant<int, 3>{});
        load_piece(SET, std::integral_constant<int, 4>{}); load_piece(SET, std::integral_constant<int, 5>{});
        load_piece(SET, std::integral_constant<int, 6>{}); load_piece(SET, std::integral_constant<int, 7>{});
    };
    auto store_all = [&](auto SET, int buf) {
        store_piece(SET, std::integral_constant<int, 0>{}, buf); store_piece(SET, std::integral_constant<int, 1>{}, buf);
        store_piece(SET, std::integral_constant<int, 2>{}, buf); store_piece(SET, std::integral_constant<int, 3>{}, buf);
        store_piece(SET, std::integral_constant<int, 4>{}, buf); store_piece(SET, std::integral_constant<int, 5>{}, buf);
        store_piece(SET, std::integral_constant<int, 6>{}, buf); store_piece(SET, std::integral_constant<int, 7>{}, buf);
    };
    auto sync = [&]() { __syncthreads(); };
    auto load_b = [&](auto SET) {       // the weight pieces of a chunk: pieces NA .. NA + NB - 1
        if constexpr (NB > 0) load_piece(SET, std::integral_constant<int, NA + 0>{});
        if constexpr (NB > 1) load_piece(SET, std::integral_constant<int, NA + 1>{});
        if constexpr (NB > 2) load_piece(SET, std::integral_constant<int, NA + 2>{});
        if constexpr (NB > 3) load_piece(SET, std::integral_constant<int, NA + 3>{});
    };
    auto load_a = [&](auto SET) {
        if constexpr (NA > 0) load_piece(SET, std::integral_constant<int, 0>{});
        if constexpr (NA > 1) load_piece(SET, std::integral_constant<int, 1>{});
        if constexpr (NA > 2) load_piece(SET, std::integral_constant<int, 2>{});
        if constexpr (NA > 3) load_piece(SET, std::integral_constant<int, 3>{});
    };
    static_assert(NA <= 4 && NB <= 4, "load_a / load_b unroll four pieces each");

    // The prologue in the order that gets the first loads out soonest (round 6: ~880 instructions ran in front of the first tile load -- divisions, tap masks,
    // 64-bit addresses, the accumulators' clears; with two waves per SIMD in the same phase that was the 3-4 us in front of the K loop): the first
    // chunk's weight pieces need only the block's column and the thread's slot, so they go first; the anchors' offsets and tap masks are computed
    // under their round trip; then the anchors' loads, bias and c_{t-1}; the accumulators are cleared while all of those are in flight.
    if (nchunks > 0) { stage_begin(); load_b(S0{}); }
    __builtin_amdgcn_sched_barrier(0);
    anchors();
    if (nchunks > 0) load_a(S0{});
    __builtin_amdgcn_sched_barrier(0);

    // ConvLSTM: bias and c_{t-1} of the cells this lane will update are requested here, in front of the K loop.  Read in the
    // epilogue they cost one exposed HBM round trip per accumulator row, 16 in a row (measured on the bf16 kernel, where the
    // epilogue was longer than the tap loop).  Lane (grp, channel) updates rows r = k * GPT + grp of its wave tile (see the epilogue).
    constexpr int OWNR = LSTM ? 16 / GPT : 1;
    float cpre[OWNR];
    float bj = 0.f, bi = 0.f, bf = 0.f, bo = 0.f;
    if constexpr (LSTM) {
        // (through a buffer descriptor, addressed as the epilogue's stores: one 32-bit lane offset, the cell's row as a scalar offset, rows past M
        // read 0 -- their cells are dropped by the stores' descriptor.  The 64-bit address of each of the 8-16 loads was 40 % of them.)
        const int C = d.C, ch = nblk * CB + wn * CPW + (l31 % CPW), grp = l31 / CPW;
        bj = d.bias[ch]; bi = d.bias[C + ch]; bf = d.bias[2 * C + ch] + 1.0f; bo = d.bias[3 * C + ch];
        const __amdgpu_buffer_rsrc_t rsci = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.cstate_in), 0, d.M * C * 4, 0x00020000);
        const int voc = ((m0 + wm * 32 + 4 * half + (GPT > 1 ? grp : 0)) * C + ch) * 4;
#pragma unroll
        for (int k = 0; k < OWNR; ++k) {
            const int srow = GPT == 1 ? (k & 3) + 8 * (k >> 2) : GPT == 2 ? 2 * (k & 1) + 8 * (k >> 1) : 8 * k;
            cpre[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsci, voc, srow * C * 4, 0));
        }
    }

#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int n = 0; n < TPW; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) accs[a][n][r] = 0.f;
    // prologue: chunk 0 straight into buffer 0, chunk 1 into register set 1
    if (nchunks > 0) {
        store_all(S0{}, 0);
        if (nchunks > 1) { stage_begin(); load_all(S1{}); }
        if constexpr (PRE > 0) stage_begin();   // parameters of chunk 2, loaded while chunk 0 is consumed
        __syncthreads();
        F32_STAMP(1);
        // two chunks per trip: the register sets alternate statically (a run-time parity branch makes hipcc wait vmcnt(0) in
        // front of every ds_write); `it` stays even
        int it = 0;
        for (; it + 3 < nchunks; it += 2) {
            chunk(std::true_type{}, std::true_type{}, S0{}, S0{}, 0); sync();
            chunk(std::true_type{}, std::true_type{}, S1{}, S1{}, 1); sync();
        }
        const int rest = nchunks - it;             // 1, 2 or 3 chunks left
        if (rest == 3) {
            chunk(std::true_type{}, std::true_type{}, S0{}, S0{}, 0); sync();
            chunk(std::false_type{}, std::true_type{}, S1{}, S1{}, 1); sync();
            chunk(std::false_type{}, std::false_type{}, S0{}, S0{}, 0);
        } else if (rest == 2) {
            chunk(std::false_type{}, std::true_type{}, S0{}, S0{}, 0); sync();
            chunk(std::false_type{}, std::false_type{}, S0{}, S1{}, 1);
        } else {
            chunk(std::false_type{}, std::false_type{}, S0{}, S0{}, 0);
        }
    }
    F32_STAMP(2);
    if constexpr (NACC > 1) {   // join the chains pairwise, fixed order
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (NACC == 2) accs[0][t][r] = accs[0][t][r] + accs[1][t][r];
                else accs[0][t][r] = (accs[0][t][r] + accs[1][t][r]) + (accs[2][t][r] + accs[3][t][r]);
            }
    }

    if constexpr (KG > 1) {   // group 1 hands its half of the K sum to group 0: [register][thread] floats through the (dead) tile buffers
        __syncthreads();
        float* xch = lds_all;
        if (gid == 1) {
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[(t * 16 + r) * 256 + tid] = accs[0][t][r];
        }
        __syncthreads();
        if (gid == 1) {       // group 0 alone runs the epilogue; group 1 only keeps it company at the statistics' three barriers
            if (LSTM && d.ln_part) { __syncthreads(); __syncthreads(); __syncthreads(); }
            return;
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) accs[0][t][r] += xch[(t * 16 + r) * 256 + tid];
    }

    // ---- epilogue --------------------------------------------------------------------------
    // (count, mean, M2) of the block's output tile, two passes over the values held in registers, fixed summation
    // order (bitwise reproducible); written as one LayerNorm partial of sample b.
    auto tile_stats = [&](const float* v, auto ownmask, auto NV, int b, int slot) {
        constexpr int nv = decltype(NV)::value;
        float s1 = 0.f, c1 = 0.f;
#pragma unroll
        for (int i = 0; i < nv; ++i)
            if ((ownmask >> i) & 1) { s1 += v[i]; c1 += 1.f; }
        s1 = wave_sum(s1); c1 = wave_sum(c1);
        __syncthreads();   // every wave is past its last LDS tile read: the tiles are dead, reuse their space
        if (lane == 0) { lds[wave] = s1; lds[4 + wave] = c1; }
        __syncthreads();
        const float cnt = (lds[4] + lds[5]) + (lds[6] + lds[7]);
        const float mean = ((lds[0] + lds[1]) + (lds[2] + lds[3])) / cnt;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < nv; ++i)
            if ((ownmask >> i) & 1) { const float dd = v[i] - mean; q = fmaf(dd, dd, q); }
        q = wave_sum(q);
        if (lane == 0) lds[8 + wave] = q;
        __syncthreads();
        if (tid == 0) {
            float* p = d.ln_part + ((size_t)b * d.ln_nparts + slot) * 4;
            p[0] = cnt; p[1] = mean; p[2] = (lds[8] + lds[9]) + (lds[10] + lds[11]); p[3] = 0.f;
        }
    };
    if constexpr (LSTM) {
        // Accumulator row r of a lane is one anchor, its column (gate t * GPT + grp, channel): the 4 gates of a cell sit in the GPT
        // lanes lane ^ (x * CPW) and the 4 / GPT tiles.  Lane grp takes rows r = k * GPT + grp: it keeps its own gate of that row
        // and receives the others from its partners in GPT - 1 xor-shuffles per tile, each partner sending the row its receiver
        // owns.  Every lane updates one cell per k (no idle lanes; GPT = 1 needs no exchange).  Register arrays are indexed
        // statically; per-lane choices are select chains.
        const int C = d.C;
        const int ch = nblk * CB + wn * CPW + (l31 % CPW);
        const int grp = l31 / CPW;
        auto pick = [&](const float (&v)[GPT], int idx) -> float {
            if constexpr (GPT == 1) {
                return v[0];
            } else if constexpr (GPT == 2) {
                return idx ? v[1] : v[0];
            } else {
                const float lo = (idx & 1) ? v[1] : v[0], hi = (idx & 1) ? v[3] : v[2];
                return (idx & 2) ? hi : lo;
            }
        };
        // Stores go through buffer descriptors: the lane-dependent part of a cell's address (row mlane, channel) is ONE 32-bit offset,
        // the cell-dependent part (row k * GPT + grp -> srow(k) rows further) a scalar offset, and rows past M fall outside the
        // descriptor and are dropped by the hardware -- no 64-bit address arithmetic and no exec-mask region per cell.
        const int M = d.M;
        const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(d.cstate_out, 0, M * C * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsh = __builtin_amdgcn_make_buffer_rsrc(d.hout, 0, M * C * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(d.gates_out ? d.gates_out : d.hout, 0, d.gates_out ? M * C * 16 : 0, 0x00020000);
        const int mlane = m0 + wm * 32 + 4 * half + (GPT > 1 ? grp : 0);
        const int vo = (mlane * C + ch) * 4, vog = (mlane * 4 * C + ch) * 4;
        float sv[OWNR];        // this lane's h values, for the fused LayerNorm statistics
        unsigned own = 0;
#pragma unroll
        for (int k = 0; k < OWNR; ++k) {
            float g4[4];
#pragma unroll
            for (int t = 0; t < 4 / GPT; ++t) {
                float rows[GPT], val[GPT];
#pragma unroll
                for (int g = 0; g < GPT; ++g) rows[g] = acc[t][k * GPT + g];
                val[0] = pick(rows, grp);
#pragma unroll
                for (int x = 1; x < GPT; ++x) val[x] = __shfl_xor(pick(rows, grp ^ x), x * CPW, 64);
#pragma unroll
                for (int g = 0; g < GPT; ++g) g4[t * GPT + g] = pick(val, g ^ grp);
            }
            // row r = k * GPT + grp of the wave tile is anchor (r & 3) + 8 (r >> 2) (+ 4 half): srow = the part that does not depend on grp
            const int srow = GPT == 1 ? (k & 3) + 8 * (k >> 2) : GPT == 2 ? 2 * (k & 1) + 8 * (k >> 1) : 8 * k;
            const int so = srow * C * 4;                     // scalar
            const float aj = fast_tanh(g4[0] + bj), ai = fast_sigmoid(g4[1] + bi);
            const float af = fast_sigmoid(g4[2] + bf), ao = fast_sigmoid(g4[3] + bo);
            const float cn = cpre[k] * af + ai * aj;
            const float hn = fast_tanh(cn) * ao;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cn), rsc, vo, so, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hn), rsh, vo, so, 0);
            const bool in = mlane + srow < M;
            sv[k] = in ? hn : 0.f;
            own |= in ? 1u << k : 0u;
            if (d.gates_out) {   // training: keep the gate activations for BPTT, [pixel][gate][C]
                const int sog = so * 4;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, aj), rsg, vog, sog, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ai), rsg, vog, sog + C * 4, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, af), rsg, vog, sog + C * 8, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ao), rsg, vog, sog + C * 12, 0);
            }
        }
        if (d.ln_part) {
            const int b = pivp_fdiv(m0, d.fd_hw_mul, d.fd_hw_sh);
            tile_stats(sv, own, std::integral_constant<int, OWNR>{}, b, ((m0 - b * HWg) / BM) * n_nblk + nblk);
        }
    } else if (!deconv && d.out_step == 1 && d.Hout == d.Hg && d.Wout == d.Wg && !d.bias && !d.relu && !d.accum && !d.ln_part &&
               (long long)d.M * d.ldo * 4 < (1LL << 31)) {
        // The ConvLSTM data gradient's epilogue (plain stride-1 conv: output pixel = anchor, no bias / ReLU / statistics): as the gate epilogue above, the
        // lane-dependent part of an element's address is ONE 32-bit offset of a buffer descriptor, the row of the accumulator register a scalar offset,
        // rows past M fall outside the descriptor and are dropped -- no divisions, no 64-bit address arithmetic, no exec-mask region per row.  (The general
        // form below divides twice per accumulator row: blocks spent 15-16 us between their last MFMA and their last store on lstm1 / lstm7 at B = 32 against
        // the gate epilogue's 4.6, which is all of the 8-11 us per launch by which the data gradient trailed the forward kernel: profiles/r06/NOTES.md 2.)
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, d.M * d.ldo * 4, 0x00020000);
        const int vo = ((m0 + wm * 32 + 4 * half) * d.ldo + nblk * BN + wn * TPW * 32 + l31) * 4;
        const int row4 = __builtin_amdgcn_readfirstlane(d.ldo * 4);
        if (ksplit > 1) {
            if (nchunks > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int t = 0; t < TPW; ++t)
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[t][r], rso, vo + t * 128, ((r & 3) + 8 * (r >> 2)) * row4, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const float v = acc[t][r];      // (a copy: __builtin_bit_cast applied to the vector ELEMENT acc[t][r] reads element 0 for every r with this hipcc)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rso, vo + t * 128, ((r & 3) + 8 * (r >> 2)) * row4, 0);
                }
        }
    } else {
        const int oy0 = deconv ? py : 0, ox0 = deconv ? px : 0;
        float sv[16 * TPW];
        unsigned long long own = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < d.M) {
                const int b = pivp_fdiv(m, d.fd_hw_mul, d.fd_hw_sh);
                const int rem = m - b * HWg;
                const int ay = pivp_fdiv(rem, d.fd_w_mul, d.fd_w_sh);
                const int ax = rem - ay * d.Wg;
                const int opix = (ay * d.out_step + oy0) * d.Wout + ax * d.out_step + ox0;
                const size_t o = ((size_t)b * d.Hout * d.Wout + opix) * d.ldo;
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const int col = nblk * BN + (wn * TPW + t) * 32 + l31;
                    if (ksplit > 1) {
                        if (nchunks > 0) atomicAdd(d.out + o + col, acc[t][r]);
                    } else {
                        float v = acc[t][r] + (d.bias ? d.bias[col] : 0.f);
                        if (d.relu) v = fmaxf(v, 0.f);
                        if (d.accum) v += d.out[o + col];
                        d.out[o + col] = v;
                        sv[t * 16 + r] = v; own |= 1ull << (t * 16 + r);
                    }
                }
            }
        }
        if (d.ln_part) {
            const int b = pivp_fdiv(m0, d.fd_hw_mul, d.fd_hw_sh);
            tile_stats(sv, own, std::integral_constant<int, 16 * TPW>{}, b,
                       (((m0 - b * HWg) / BM) * n_nblk + nblk) * (int)gridDim.y + phase);
        }
    }
#ifdef PIVP_F32_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the block's stores have left
    F32_STAMP(3);
#endif
}

template <int WM, int WN, int NTB, bool LSTM, int KG = 1>
static int launch_igemm(const IgemmDesc& d, hipStream_t stream, int ksplit = 1, int* ln_nparts = nullptr) {
    constexpr int BM = 32 * WM, BN = 32 * NTB;
    const int n_nblk = LSTM ? (d.C / (8 * NTB)) : (d.N / BN);
    const int mblk = (d.M + BM - 1) / BM;
    constexpr int lds_bytes = ig_lds_bytes<WM, NTB, KG>();
    static PerDeviceOnce once;
    if (pivp_ensure_dyn_lds(once, reinterpret_cast<const void*>(&igemm_f32_kernel<WM, WN, NTB, LSTM, KG>), lds_bytes) != PIVP_OK) return PIVP_ERR_LAUNCH;
    dim3 grid(mblk * n_nblk, d.nphase, ksplit);
    IgemmDesc dd = d;
    dd.n_mblk = mblk;
    pivp_fastdiv((unsigned)mblk, &dd.fd_mb_mul, &dd.fd_mb_sh);
    pivp_fastdiv((unsigned)(d.Hg * d.Wg), &dd.fd_hw_mul, &dd.fd_hw_sh);
    pivp_fastdiv((unsigned)d.Wg, &dd.fd_w_mul, &dd.fd_w_sh);
    pivp_fastdiv((unsigned)((d.c0 + d.c1) >> 5), &dd.fd_cc_mul, &dd.fd_cc_sh);
    // fused LayerNorm partials: only when no tile straddles two samples and the caller's buffer holds them
    const int hwg = d.Hg * d.Wg;
    const int np = (hwg / BM) * n_nblk * d.nphase;
    dd.ln_nparts = (d.ln_part && ksplit == 1 && hwg % BM == 0 && np <= d.ln_cap) ? np : 0;
    if (!dd.ln_nparts) dd.ln_part = nullptr;
    if (ln_nparts) *ln_nparts = dd.ln_nparts;
    hipLaunchKernelGGL((igemm_f32_kernel<WM, WN, NTB, LSTM, KG>), grid, dim3(256 * KG), lds_bytes, stream, dd);
    return PIVP_LAUNCH_STATUS();
}

int igemm_validate(const IgemmDesc& d, bool lstm) {
    PIVP_CHECK_ARG(d.x0 && d.w && d.c0 > 0 && d.c0 % 32 == 0 && d.c1 % 32 == 0 && d.c1 >= 0);
    PIVP_CHECK_ARG(d.c1 == 0 || d.x1);
    PIVP_CHECK_ARG(d.ld0 >= d.c0 && d.ld0 % 4 == 0 && (d.c1 == 0 || (d.ld1 >= d.c1 && d.ld1 % 4 == 0)));
    PIVP_CHECK_ARG(d.B > 0 && d.Hin > 0 && d.Win > 0 && d.Hg > 0 && d.Wg > 0 && d.in_step >= 1);
    PIVP_CHECK_ARG(d.M == d.B * d.Hg * d.Wg);
    PIVP_CHECK_ARG(d.deconv ? (d.nphase == 4 && d.in_step == 1 && d.out_step == 2) : (d.nphase == 1 && d.ksize >= 1 && d.ksize <= 5 && d.pad >= 0));   // 5x5 tap bitmask
    PIVP_CHECK_ARG(d.bytes0 > 0 && d.bytesw > 0 && (d.c1 == 0 || d.bytes1 > 0));
    PIVP_CHECK_ARG(d.wcin >= d.c0 + d.c1 && d.wcin % 32 == 0);
    if (lstm) {
        PIVP_CHECK_ARG(d.C > 0 && d.C % 32 == 0 && d.N == 4 * d.C && d.bias && d.cstate_in && d.cstate_out && d.hout);
        PIVP_CHECK_ARG(d.nphase == 1 && d.in_step == 1 && d.Hg == d.Hin && d.Wg == d.Win && d.ksize == 5 && d.pad == 2 && !d.deconv);
        PIVP_CHECK_ARG((long long)d.M * d.C * 4 < (1LL << 31));         // c / h through 32-bit buffer offsets
    } else {
        PIVP_CHECK_ARG(d.out && d.N % 32 == 0 && d.N >= 32 && d.ldo >= d.N);
        PIVP_CHECK_ARG(d.out_step >= 1 && d.Hout > 0 && d.Wout > 0);
        PIVP_CHECK_ARG((d.Hg - 1) * d.out_step + (d.deconv ? 1 : 0) < d.Hout && (d.Wg - 1) * d.out_step + (d.deconv ? 1 : 0) < d.Wout);
    }
    return PIVP_OK;
}

// Tile choice: the largest block tile that still gives every one of the 256 CUs a block.
// variant 0 = auto, 1 = 4x1 waves (BM 128), 2 = 2x2 (BM 64), 3 = 1x4 (BM 32).
int igemm_lstm(const IgemmDesc& d, hipStream_t stream, int variant, int* ln_nparts) {
    int rc = igemm_validate(d, true);
    if (rc != PIVP_OK) return rc;
    if (variant == 0) {
        // Two resident blocks per CU (64-row tile, 240 VGPRs) beat the 128-row tile (320 VGPRs: one block, one wave per SIMD) at every grid
        // size: at M = 131072 / 32768 (config 5's maps, scripts/bench_lstm_layers.py 128) 139.9 against 132.3 TFLOP/s over the seven layers,
        // so the 128-row tile is never picked (variant 1 stays for the tests)
        const long nb = d.C / 32;
        if ((long)(d.M / 64) * nb >= 512) variant = 2;
        else if ((long)(d.M / 64) * nb >= 256) variant = 4;      // one 64-row block per CU: as two K groups (falls back to variant 2 on an odd chunk count)
        else variant = 3;
    }
    // (A square 64 x 64 tile, 16 channels x 4 gates per block, 20 % less operand traffic per MAC -- round 4's variants 5 / 6 -- was slower on every
    // layer, profiles/r04/NOTES.md 3: in the history.)
    // (128-row tiles as ONE 8-wave block per CU, two K groups -- a third less L2 traffic, half the staging per MFMA -- were measured on the
    // layers that fill the chip: lstm1 126 -> 119.5 TF, lstm7 135.5 -> 132, rollout 8.67 -> 8.86 ms; 46 VGPRs spill under the 256 cap.  Not kept.)
    switch (variant) {
        case 4:   // 64-row tile as two K groups of 4 waves (two waves per SIMD where the grid gives every CU one block): lstm4 106.6 -> 104.6 us,
                  // lstm6 153.6 -> 150.2 at B = 32; needs an even chunk count (lstm3 has 75; round 6 gave group 1 a null chunk -- tap 25 loads zeros -- so that lstm3 could take this form: 122 TFLOP/s either way, not kept)
            if (((25 * ((d.c0 + d.c1) >> 5)) & 1) == 0) return launch_igemm<2, 2, 4, true, 2>(d, stream, 1, ln_nparts);
            return launch_igemm<2, 2, 4, true>(d, stream, 1, ln_nparts);
        case 1: return launch_igemm<4, 1, 4, true>(d, stream, 1, ln_nparts);
        case 2: return launch_igemm<2, 2, 4, true>(d, stream, 1, ln_nparts);
        case 3: {
            // the in-block K split needs an even number of chunks (25 taps x (c0 + c1) / 32: always even when the channel count is a multiple of 64)
            const int ncc = (d.c0 + d.c1) >> 5;
            if (((25 * ncc) & 1) == 0) return launch_igemm<1, 4, 4, true, 2>(d, stream, 1, ln_nparts);
            return launch_igemm<1, 4, 4, true>(d, stream, 1, ln_nparts);
        }
    }
    return PIVP_ERR_BADARG;
}

// Plain conv / transposed conv: pick the block tile that fills the 256 CUs.  Blocks = (M/BM) * (N/BN) * phases.
// Long-K data gradients (the ConvLSTM's 5x5 over 4C channels).  The output may be produced by K-split blocks with atomic adds into a
// pre-zeroed destination, so both the tile AND the split are free: pick the pair with the least modelled time
//   blocks per CU x tile area x (taps per split + c) / tile efficiency,
// c = the per-block prologue / epilogue in taps' worth of time: 3 when a block has its CU to itself (and its one wave per SIMD then
// runs at 0.8), 1.5 with two resident blocks, 1 with three or more (their fixed phases hide under each other's MFMAs); resident
// blocks per CU are bounded by the tile's registers and LDS.  At M = 2048..8192 the block count, not the tile shape, decides.  Splits are contiguous ranges of the chunk
// sequence, so any split count is balanced.  Measured at B = 32 (scripts/dgrad_ks_scan.sh, us per launch, best of all forced splits
// = the model's choice): lstm1/2 128x64 / 2 splits 113.7 (64x64 unsplit 117.8), lstm3 128x96 / 8 90.4, lstm4 64x128 / 4 109.3 (by kernel
// rows, 5 splits: 123.6), lstm5 64x64 / 8 91.4 (5: 97.0), lstm6 128x96 / 4 159.7 (128x64 / 5: 182.1), lstm7 64x128 unsplit 210.6.
// Returns false when the descriptor is not such a conv.  ks == 1: plain stores, the destination need not be zeroed.
static bool dgrad_choice(const IgemmDesc& d, int& bt, int& bks) {
    if (!(d.ksplit_ok && !d.deconv && !d.bias && !d.relu && !d.accum && d.ksize * d.ksize * ((d.c0 + d.c1) / 32) > 40)) return false;
    const int nt = d.N / 32;
    struct Tile { int wm, wn, ntb; double eff; int resident; };
    static const Tile tiles[] = {{2, 2, 2, 0.80, 4}, {4, 1, 1, 0.70, 3}, {4, 1, 2, 0.90, 2}, {4, 1, 3, 0.95, 2}, {4, 1, 4, 1.00, 1}, {2, 2, 4, 0.95, 2}, {1, 4, 4, 0.80, 3}};
    static const double fixed_taps[] = {3.0, 1.5, 1.0, 1.0};
    static const double alone[] = {0.90, 1.0, 1.0, 1.0};      // one wave per SIMD hides none of its own waits (0.80 until round 6: see the split penalty below)
    const int cus = pivp_cu_count();
    bt = -1; bks = 1;
    double bcost = 1e300;
    const int nchunks = d.ksize * d.ksize * ((d.c0 + d.c1) / 32);
    // a split output is cleared first and then takes ks atomic adds per element: what that costs grows with the rows (lstm1 / lstm2 at B = 32, 32,768 rows x 64
    // columns: 128 x 64 unsplit 107.4 us against 112 in two splits once the epilogue stopped dividing -- r06_c13; before, the split hid a 15-us epilogue)
    const double split_penalty = 1.01 + 0.14 * (d.M >= 32768 ? 1.0 : d.M / 32768.0);
    for (int t = 0; t < 7; ++t) {
        if (nt % tiles[t].ntb) continue;
        const int bm = 32 * tiles[t].wm, bn = 32 * tiles[t].ntb;
        const long mb = (d.M + bm - 1) / bm, nb = d.N / bn;
        for (int ks = 1; ks <= 10 && ks * 8 <= nchunks; ++ks) {          // >= 8 chunks per split
            const long blocks = mb * nb * ks;
            const long per_cu = (blocks + cus - 1) / cus;
            const int res = (int)(per_cu < tiles[t].resident ? per_cu : tiles[t].resident);
            const double taps = (double)(d.ksize * d.ksize) / ks;       // taps of one split
            const double cost = (double)per_cu * bm * bn * (taps + fixed_taps[res - 1]) / (tiles[t].eff * alone[res - 1]) * (ks > 1 ? split_penalty : 1.0);
            if (cost < bcost) { bcost = cost; bt = t; bks = ks; }
        }
    }
    return bt >= 0;
}
int igemm_conv_ksplit(const IgemmDesc& d) {
    int bt, bks;
    return dgrad_choice(d, bt, bks) ? bks : 1;
}

int igemm_conv(const IgemmDesc& d, hipStream_t stream, int* ln_nparts) {
    int rc = igemm_validate(d, false);
    if (rc != PIVP_OK) return rc;
    const int nt = d.N / 32;
    // transposed conv on maps that tile into 8 x 16 input patches: all four parities per block (deconv_tile.hip), unless the grid would be tiny
    if (deconv_tile_ok(d) && (long)d.B * (d.Hin / 8) * (d.Win / 16) * nt >= 16)
        return deconv_tile(d, stream, ln_nparts, d.bf16);
    const long full = (long)((d.M + 127) / 128) * d.nphase;   // blocks with BM = 128 and the whole N in one block
    int bt = -1, bks = 1;
    if (dgrad_choice(d, bt, bks)) {
        switch (bt) {
            case 0: return launch_igemm<2, 2, 2, false>(d, stream, bks, ln_nparts);
            case 1: return launch_igemm<4, 1, 1, false>(d, stream, bks, ln_nparts);
            case 2: return launch_igemm<4, 1, 2, false>(d, stream, bks, ln_nparts);
            case 3: return launch_igemm<4, 1, 3, false>(d, stream, bks, ln_nparts);
            case 4: return launch_igemm<4, 1, 4, false>(d, stream, bks, ln_nparts);
            case 5: return launch_igemm<2, 2, 4, false>(d, stream, bks, ln_nparts);
            case 6: return launch_igemm<1, 4, 4, false>(d, stream, bks, ln_nparts);
        }
        return PIVP_ERR_BADARG;
    }
    // Short K (the 3x3 stride-2 convs and the transposed convs: 2..18 chunks) or too few big tiles to fill the chip:
    // 32-row tiles with K split over the waves (igemm_small.hip).  Measured at B = 32 (scripts/bench_tail_ops.py):
    // enc1 15.0 -> 6.6 us, enc2 21.4 -> 9.7, enc4 19.3 -> 14.2, enc5 33.7 -> 23.8, enc6 47.6 -> 39.5.
    {
        const bool can = d.ldo % 4 == 0 && ((uintptr_t)d.out & 15) == 0 && (!d.bias || ((uintptr_t)d.bias & 15) == 0);
        const long big = full * ((nt + 3) / 4);
        const int chunks = (d.deconv ? 4 : d.ksize * d.ksize) * ((d.c0 + d.c1) / 32);
        if (can && (big < 128 || chunks <= 40)) {
            return igemm_small(d, stream, ln_nparts);
        }
    }
    if (nt > 4) {                                             // wide outputs: several column blocks
        if (nt % 4 == 0) return full * (nt / 4) >= 256 ? launch_igemm<4, 1, 4, false>(d, stream, 1, ln_nparts) : launch_igemm<2, 2, 4, false>(d, stream, 1, ln_nparts);
        if (nt % 3 == 0) return launch_igemm<4, 1, 3, false>(d, stream, 1, ln_nparts);
        if (nt % 2 == 0) return full * (nt / 2) >= 256 ? launch_igemm<4, 1, 2, false>(d, stream, 1, ln_nparts) : launch_igemm<2, 2, 2, false>(d, stream, 1, ln_nparts);
        return launch_igemm<4, 1, 1, false>(d, stream, 1, ln_nparts);
    }
    if (full >= 256) {
        switch (nt) {
            case 1: return launch_igemm<4, 1, 1, false>(d, stream, 1, ln_nparts);
            case 2: return launch_igemm<4, 1, 2, false>(d, stream, 1, ln_nparts);
            case 3: return launch_igemm<4, 1, 3, false>(d, stream, 1, ln_nparts);
            case 4: return launch_igemm<4, 1, 4, false>(d, stream, 1, ln_nparts);
        }
        return PIVP_ERR_BADARG;
    }
    if (nt == 4 && full * 2 < 256) return launch_igemm<1, 4, 4, false>(d, stream, 1, ln_nparts);   // BM 32
    if (nt == 4) return launch_igemm<2, 2, 4, false>(d, stream, 1, ln_nparts);                      // BM 64
    if (nt == 2 && full * 2 >= 128) return launch_igemm<2, 2, 2, false>(d, stream, 1, ln_nparts);  // BM 64
    return launch_igemm<4, 1, 1, false>(d, stream, 1, ln_nparts);                                    // BN 32: N/32 column blocks
}

}  // namespace pivp

#ifdef PIVP_F32_STAMPS
extern "C" int pivp_debug_f32_stamps(long long* out, int n) {   // n <= 2048 * 8 values: [block][entry, loop start, loop end, stores done] in 10 ns ticks, then the same in shader cycles
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pivp::pivp_f32_stamps), sizeof(long long) * n) == hipSuccess ? 0 : -2;
}
#endif

PIVP_DEFINE_MAIN_PRIO_SETTER(igemm_f32)
