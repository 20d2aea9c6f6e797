// Weight gradient of the stride-2 3x3 convs (enc1, enc2: TM:491-497) and transposed convs (enc4, enc5, enc6: TM:505-507), all nine taps
// from ONE staging of the operands.
//
// Both are the same sum over ANCHORS a of a small map (conv: output pixels, transposed conv: input pixels) and the pixels p = 2a - 1 + k
// of the map twice as large:      dW[tap k][ci][n] += sum_a  X[..][ci] * dY[..][n],      X = big / dY = small for the conv, the other way for
// the transposed conv.  igemm_wgrad_kernel gives every tap its own block, so each operand tile is fetched nine times (enc6 at B = 32:
// 151 MB through L2 per launch for 2.4 GFLOP, ~50 us).  Here a block takes a chunk of 4 x 8 anchors, stages the chunk's small tile
// [32][CS] and the 9 x 17 patch of the big map [153][CB] once -- global_load_lds_dwordx4 straight into a double-buffered LDS image, no
// registers, no ds_write -- and its waves run all nine taps against it on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: rows = ci,
// columns = n, k = anchors; a fragment is one conflict-free ds_read_b32 per lane from the pixel-major image, and a tap only moves the
// patch pixel).  A wave owns NS of the block's TI x TJ x 9 (32 x 32 tile, tap) units.  One barrier per chunk.
// The reduction runs over anchors AND timesteps (WgradDesc::tcount): the BPTT sweep hands a whole batch of timesteps to one launch, so
// a block's epilogue -- its units' accumulators into the block's own slot of WgradDesc::part, in fragment order (coalesced) -- is paid
// once per batch.  wgrad3x3s2_reduce sums the slots in a fixed order into the packed gradient: the result does not depend on scheduling.
#include "pivp_kernels.h"

namespace pivp {

namespace {
__device__ float g_w3_zero[4];      // what out-of-map patch pixels (row -1 / column -1 of the big map) and the image's padding lanes load

constexpr int W3_PW = 17, W3_PH = 9, W3_PPIX = W3_PW * W3_PH;     // patch of a 4 x 8 anchor chunk
constexpr int W3_TARGET_BLOCKS = 256;
constexpr int W3_BIAS_TAIL = 128;      // floats behind a block's [unit][16][64] accumulators: its column sums of dY (bias gradient), <= 96 used

struct W3Shape { int deconv, TI, TJ, NW; };
// which instance takes a layer (0 = none): transposed 96 -> 96 (enc5) as one 3 x 3-tile block of twelve waves (seven units each, six for the last three),
// transposed convs on multiples of 64 channels (enc6, enc4) as 2 x 2-tile blocks of twelve waves (one kernel row of one tile each), convs (enc1, enc2) as
// one-tile blocks of three waves (tiny layers: parallelism first).  Waves per block at B = 32, nine timesteps per launch (r05_c17 .. r05_c19): enc6 4 / 8 / 12
// waves 265 / 244 / 235 us, enc4 76 / 69 / 67, enc5 8 / 12 waves 159 / 152, enc1 and enc2 3 / 9 waves 30 / 29.
inline int w3_instance(const WgradDesc& d) {
    if (d.ksize != 3 || d.stride != 2 || d.pad != 1 || d.c1 != 0 || d.cin != d.c0 || d.wcin != d.cin) return 0;
    if (d.Hg % 4 || d.Wg % 8 || d.cin % 32 || d.N % 32 || d.ld0 % 4 || d.ldy % 4) return 0;
    if (d.deconv) {
        if (d.Hy != 2 * d.Hx || d.Wy != 2 * d.Wx || d.Hg != d.Hx || d.Wg != d.Wx) return 0;
        if (d.cin == 96 && d.N == 96) return 2;
        if (d.cin % 64 == 0 && d.N % 64 == 0) return 1;
        return 0;
    }
    if (d.Hx != 2 * d.Hy || d.Wx != 2 * d.Wy || d.Hg != d.Hy || d.Wg != d.Wy) return 0;
    return 3;
}
inline W3Shape w3_shape(int inst) {
    return inst == 1 ? W3Shape{1, 2, 2, 12} : inst == 2 ? W3Shape{1, 3, 3, 12} : W3Shape{0, 1, 1, 3};
}
// grid: a function of the descriptor's ONE-timestep geometry alone, so every launch of a sweep, the partial buffer and the reduction agree
inline void w3_grid(const WgradDesc& d, const W3Shape& sh, int& nblk, int& nsplit) {
    nblk = (d.cin / (32 * sh.TI)) * (d.N / (32 * sh.TJ));
    const int cpt = d.B * (d.Hg / 4) * (d.Wg / 8);
    nsplit = W3_TARGET_BLOCKS / nblk;
    if (nsplit > cpt) nsplit = cpt;
    if (nsplit < 1) nsplit = 1;
}
}  // namespace

template <bool DECONV, int TI, int TJ, int NW>
__global__ __launch_bounds__(NW * 64, 1) void wgrad3x3s2_kernel(const WgradDesc d) {
    constexpr int CS = (DECONV ? TI : TJ) * 32, CB = (DECONV ? TJ : TI) * 32;     // floats per pixel of the small tile / the big patch
    constexpr int BIG4 = W3_PPIX * CB / 4, SM4 = 32 * CS / 4, TOT4 = BIG4 + SM4;  // float4 of one chunk's image: patch | small tile
    constexpr int KL = (TOT4 + NW * 64 - 1) / (NW * 64);                          // DMAs per thread and chunk
    constexpr int BUF = KL * NW * 64 * 4;                                          // floats per LDS buffer (whole instructions)
    constexpr int NU = TI * TJ * 9, NS = (NU + NW - 1) / NW;
    constexpr bool CONTIG = NU % NW == 0 && 9 % NS == 0;    // a wave's units are NS consecutive taps of ONE (ci, n) tile: its small fragment is shared by the slots
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Hs = d.Hg, Ws = d.Wg, Hb = 2 * Hs, Wb = 2 * Ws;
    const int nnb = d.N / (32 * TJ);
    // Workgroups go to the 8 XCDs in turn (linear id mod 8).  The grid is (output blocks, P pixel splits); an XCD's P / 8 x nblk workgroups take P / 8
    // NEIGHBOURING positions of every stretch of P chunks (see the loop) and all output blocks of each: they share operand patches and halo rows in one L2.
    const int nblk = gridDim.x, P = gridDim.y, L = blockIdx.y * nblk + blockIdx.x;
    const int oblk = P % 8 == 0 ? (L >> 3) % nblk : (int)blockIdx.x;
    const int spos = P % 8 == 0 ? (L & 7) * (P / 8) + (L >> 3) / nblk : (int)blockIdx.y;
    const int nb = oblk % nnb, cb = oblk / nnb;
    const int ci0 = cb * TI * 32, n0 = nb * TJ * 32;
    const char* const sm_p = reinterpret_cast<const char*>(DECONV ? d.x0 : d.dy) + (DECONV ? ci0 : n0) * 4;
    const char* const bg_p = reinterpret_cast<const char*>(DECONV ? d.dy : d.x0) + (DECONV ? n0 : ci0) * 4;
    const int ld_s = DECONV ? d.ld0 : d.ldy, ld_b = DECONV ? d.ldy : d.ld0;
    const long long ts_s = DECONV ? d.ts_x0 : d.ts_dy, ts_b = DECONV ? d.ts_dy : d.ts_x0;
    const int cw = Ws / 8, cpi = (Hs / 4) * cw, cpt = d.B * cpi;
    const int tcount = d.tcount > 1 ? d.tcount : 1;
    const int nchunks = cpt * tcount;
    // Chunks are dealt round-robin: in step tau the P splits work on chunks tau * P .. + P - 1, a contiguous stretch of the maps (blocks that each walked
    // their own contiguous slice sat 1 MB apart at every moment).
    const int c_begin = spos, c_end = nchunks;

    // ---- staging roles: DMA k of this wave fills image float4 [(k * NW + wave) * 64 + lane] -------------------------------------------
    int rel[KL];                                  // byte offset of the lane's source from the chunk's patch origin / first anchor
    unsigned m_big = 0, m_top = 0, m_left = 0, m_ok = 0;
#pragma unroll
    for (int k = 0; k < KL; ++k) {
        const int q = (k * NW + wave) * 64 + lane;
        rel[k] = 0;
        if (q < BIG4) {
            const int pix = q / (CB / 4), c4 = q - pix * (CB / 4), pr = pix / W3_PW, pc = pix - pr * W3_PW;
            rel[k] = ((pr * Wb + pc) * ld_b + c4 * 4) * 4;
            m_big |= 1u << k; m_ok |= 1u << k;
            if (pr == 0) m_top |= 1u << k;
            if (pc == 0) m_left |= 1u << k;
        } else if (q < TOT4) {
            const int q2 = q - BIG4, a = q2 / (CS / 4), c4 = q2 - a * (CS / 4);
            rel[k] = (((a >> 3) * Ws + (a & 7)) * ld_s + c4 * 4) * 4;
            m_ok |= 1u << k;
        }
    }
    const char* const zero = reinterpret_cast<const char*>(g_w3_zero);
    auto issue = [&](int c, int buf) {
        const int tj = c / cpt, r = c - tj * cpt, b = r / cpi, ci = r - b * cpi, cr = ci / cw, cc = ci - cr * cw;     // block-uniform
        const int ar0 = cr * 4, ac0 = cc * 8;
        const char* sb = sm_p + (long long)tj * ts_s + ((long long)(b * Hs + ar0) * Ws + ac0) * ld_s * 4;
        const char* bb = bg_p + (long long)tj * ts_b + ((long long)(b * Hb + 2 * ar0 - 1) * Wb + 2 * ac0 - 1) * ld_b * 4;
        const unsigned oob = (ar0 == 0 ? m_top : 0u) | (ac0 == 0 ? m_left : 0u);      // this lane's DMAs that fall off the big map
#pragma unroll
        for (int k = 0; k < KL; ++k) {
            const char* src = ((m_big >> k) & 1) ? bb + rel[k] : sb + rel[k];
            if (((oob | ~m_ok) >> k) & 1) src = zero;
            float* dst = lds + buf * BUF + (k * NW + wave) * 256;       // wave-uniform; the DMA adds lane * 16 bytes
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    // ---- this wave's units -----------------------------------------------------------------------------------------------------------
    int s_addr[NS], b_addr[NS], unit[NS];         // fragment addresses (floats) inside a buffer; the unit's canonical index ij * 9 + tap
    bool last_ok = true;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        int ij, tap;
        if constexpr (CONTIG) { ij = wave / (9 / NS); tap = (wave % (9 / NS)) * NS + s; }
        else {
            int u = s * NW + wave;
            if (u >= NU) { u = 0; last_ok = false; }       // (only the last slot can run past the units)
            ij = u / 9; tap = u - ij * 9;
        }
        const int i = ij % TI, j = ij / TI, ky = tap / 3, kx = tap - ky * 3;
        unit[s] = ij * 9 + tap;
        s_addr[s] = BIG4 * 4 + half * CS + l31 + (DECONV ? i : j) * 32;
        b_addr[s] = 2 * half * CB + l31 + (ky * W3_PW + kx) * CB + (DECONV ? j : i) * 32;
    }
    f32x16 acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
    // Bias gradient = column sums of dY, by the blocks of channel block 0, straight from the staged image: the transposed conv's patch rows 1..8 x
    // columns 1..16 are the 8 x 16 dY pixels 2a, 2a + 1 of the chunk's anchors (every dY pixel belongs to exactly one chunk); the conv's dY is the
    // small tile.  Thread = (dY column, a group of those pixels).
    constexpr int CY = DECONV ? CB : CS, BG = NW * 64 / CY, BPIX = DECONV ? 128 : 32;
    const bool do_bias = d.db != nullptr && cb == 0 && tid < BG * CY;      // (block-uniform but for the thread bound)
    const int bcol = tid % CY, bgrp = tid / CY;
    float bsum = 0.f;
    auto bias = [&](int buf) {
        const float* L = lds + buf * BUF;
        for (int p = bgrp; p < BPIX; p += BG)
            bsum += DECONV ? L[((1 + (p >> 4)) * W3_PW + 1 + (p & 15)) * CB + bcol] : L[BIG4 * 4 + p * CS + bcol];
    };
    auto mma = [&](int buf) {
        const float* L = lds + buf * BUF;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {         // anchors 2 kk + half: row kk / 4, column 2 (kk % 4) + half of the chunk
            const int so = 2 * kk * CS, bo = ((2 * (kk / 4)) * W3_PW + 4 * (kk % 4)) * CB;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (!CONTIG && s == NS - 1 && !last_ok) continue;
                const float sv = L[s_addr[s] + so], bv = L[b_addr[s] + bo];
                const float a = DECONV ? sv : bv, y = DECONV ? bv : sv;
                acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, y, acc[s], 0, 0, 0);
            }
        }
    };

    if (c_begin < c_end) {
        issue(c_begin, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int buf = 0;
        for (int c = c_begin; c < c_end; c += P) {
            if (c + P < c_end) issue(c + P, buf ^ 1);      // (issued behind the first two k-steps instead, so that the address arithmetic runs under MFMAs in
            mma(buf);                                      // flight: enc6 242 us against 235, r05_c20)
            if (do_bias) bias(buf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's part of chunk c + 1 has landed ...
            __syncthreads();                                      // ... everybody's has, and nobody still reads chunk c
            buf ^= 1;
        }
    }

    // ---- epilogue: the block's slot of the partial buffer, [unit][16 accumulator registers][64 lanes] -----------------------------------
    float* const slot = d.part + ((size_t)spos * nblk + oblk) * (NU * 1024 + W3_BIAS_TAIL);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (!CONTIG && s == NS - 1 && !last_ok) continue;
        float* q = slot + unit[s] * 1024 + lane;
        if (d.part_overwrite) {
#pragma unroll
            for (int r = 0; r < 16; ++r) q[r * 64] = acc[s][r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) q[r * 64] += acc[s][r];
        }
    }
    // the column sums: the pixel groups meet through LDS (free behind the loop's last barrier), one plain sum per column and block into the slot's tail
    // (one atomic per thread into db, the first version: 1,024-1,536 adds per address and launch, ~50 us of serialised L2 atomics)
    if (d.db != nullptr && cb == 0) {
        if (tid < BG * CY) lds[bgrp * CY + bcol] = bsum;
        __syncthreads();
        if (tid < CY) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < BG; ++g) v += lds[g * CY + tid];
            float* q = slot + NU * 1024 + tid;
            *q = d.part_overwrite ? v : *q + v;
        }
    }
}

// dw (and db) += the sum over the pixel splits of the blocks' slots, in a fixed order.  Block = 32 consecutive elements of one output block's
// [unit][16][64] slot x 8 groups of splits; the eight partial sums meet through LDS.
__global__ __launch_bounds__(256) void wgrad3x3s2_reduce_kernel(const WgradDesc d, int TI, int TJ, int nblk, int nsplit) {
    __shared__ float red[8][32];
    const int NU = TI * TJ * 9, SLOT = NU * 1024 + W3_BIAS_TAIL;
    const int blk = blockIdx.y, el = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const int nnb = d.N / (32 * TJ), nb = blk % nnb, cb = blk / nnb;
    const bool is_bias = (int)blockIdx.x >= NU * 32;              // the last TJ blocks of a row: the slots' bias tails (32 columns each)
    if (is_bias && (cb != 0 || d.db == nullptr)) return;
    const int e = is_bias ? NU * 1024 + ((int)blockIdx.x - NU * 32) * 32 + el : blockIdx.x * 32 + el;
    const size_t stride = (size_t)nblk * SLOT;
    const float* src = d.part + (size_t)blk * SLOT + e;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int sp = sg;
    for (; sp + 24 < nsplit; sp += 32) {
        a0 += src[(size_t)sp * stride]; a1 += src[(size_t)(sp + 8) * stride];
        a2 += src[(size_t)(sp + 16) * stride]; a3 += src[(size_t)(sp + 24) * stride];
    }
    for (; sp < nsplit; sp += 8) a0 += src[(size_t)sp * stride];
    red[sg][el] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sg) return;
    float v = red[0][el];
#pragma unroll
    for (int g = 1; g < 8; ++g) v += red[g][el];
    if (is_bias) { d.db[nb * TJ * 32 + e - NU * 1024] += v; return; }
    const int u = e >> 10, r = (e >> 6) & 15, lane = e & 63, l31 = lane & 31, half = lane >> 5;
    const int ij = u / 9, tap = u - ij * 9, i = ij % TI, j = ij / TI;
    const int ci = (cb * TI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, n = (nb * TJ + j) * 32 + l31;
    d.dw[(((size_t)tap * (d.wcin >> 5) + (ci >> 5)) * d.N + n) * 32 + (ci & 31)] += v;
}

bool wgrad3x3s2_ok(const WgradDesc& d) { return w3_instance(d) != 0; }

long long wgrad3x3s2_part_floats(const WgradDesc& d) {
    const int inst = w3_instance(d);
    if (!inst) return 0;
    const W3Shape sh = w3_shape(inst);
    int nblk, nsplit;
    w3_grid(d, sh, nblk, nsplit);
    return (long long)nblk * nsplit * (sh.TI * sh.TJ * 9 * 1024 + W3_BIAS_TAIL);
}

template <bool DECONV, int TI, int TJ, int NW>
static int launch_w3(const WgradDesc& d, int nblk, int nsplit, hipStream_t s) {
    constexpr int CS = (DECONV ? TI : TJ) * 32, CB = (DECONV ? TJ : TI) * 32;
    constexpr int TOT4 = W3_PPIX * CB / 4 + 32 * CS / 4, KL = (TOT4 + NW * 64 - 1) / (NW * 64);
    constexpr int lds_bytes = 2 * KL * NW * 64 * 16;
    static_assert(lds_bytes <= 160 * 1024, "two chunk images must fit the CU's LDS");
    static PerDeviceOnce once;
    const int rc = pivp_ensure_dyn_lds(once, reinterpret_cast<const void*>(&wgrad3x3s2_kernel<DECONV, TI, TJ, NW>), lds_bytes);
    if (rc != PIVP_OK) return rc;
    hipLaunchKernelGGL((wgrad3x3s2_kernel<DECONV, TI, TJ, NW>), dim3(nblk, nsplit), dim3(NW * 64), lds_bytes, s, d);
    return PIVP_LAUNCH_STATUS();
}

int wgrad3x3s2(const WgradDesc& d, hipStream_t s) {
    const int inst = w3_instance(d);
    PIVP_CHECK_ARG(inst && d.x0 && d.dy && d.part);
    const W3Shape sh = w3_shape(inst);
    int nblk, nsplit;
    w3_grid(d, sh, nblk, nsplit);
    return inst == 1 ? launch_w3<true, 2, 2, 12>(d, nblk, nsplit, s) : inst == 2 ? launch_w3<true, 3, 3, 12>(d, nblk, nsplit, s)
                                                                                 : launch_w3<false, 1, 1, 3>(d, nblk, nsplit, s);
}

int wgrad3x3s2_reduce(const WgradDesc& d, hipStream_t s) {
    const int inst = w3_instance(d);
    PIVP_CHECK_ARG(inst && d.part && d.dw);
    const W3Shape sh = w3_shape(inst);
    int nblk, nsplit;
    w3_grid(d, sh, nblk, nsplit);
    hipLaunchKernelGGL(wgrad3x3s2_reduce_kernel, dim3(sh.TI * sh.TJ * 9 * 32 + sh.TJ, nblk), dim3(256), 0, s, d, sh.TI, sh.TJ, nblk, nsplit);
    return PIVP_LAUNCH_STATUS();
}

}  // namespace pivp
