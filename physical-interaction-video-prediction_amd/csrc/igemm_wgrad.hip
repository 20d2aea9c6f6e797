// Weight gradients of every MFMA convolution, and the weight re-pack the data-gradient needs.
//
// wgrad:  dW[tap][ci][n] += sum over anchors a of  X[pixA(a, tap)][ci] * dY[pixB(a, tap)][n]
//   conv (ConvLSTM 5x5 p2, 3x3 s2 p1):  anchors = output pixels, pixA = a*stride - pad + k, pixB = a
//   transposed 3x3 s2 p1 (TM:505-507):  anchors = input pixels,  pixA = a,                  pixB = 2a - 1 + k
// i.e. a GEMM dW_tap = X_tap^T (Cin x M) . dY (M x N) whose reduction runs over pixels.  On the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32): A[i = ci][k = pixel], B[k = pixel][j = n]; both operands are read from pixel-major LDS
// tiles [32 pixels][channels] with one conflict-free ds_read_b32 per lane.  A block owns one tap, 64 input
// channels x 128 output columns, and a slice of the pixels; slices are combined with fp32 atomic adds straight into
// the K-inner packed gradient (same layout as the weight, so Adam is elementwise).  Gradients therefore accumulate
// across blocks, timesteps and calls until the host clears them (Chainer: cleargrads + backward, TM:950).
#include <type_traits>

#include "pivp_kernels.h"

// (Timing-only ablations of wgrad5x5_kernel, round 2/3, in the history: no global loads / LDS stores in the loop 222 us, also no LDS reads 208,
// loads without stores 221, stores without loads 217 against the full kernel's 265 on lstm7 -- loads and stores cost nothing on their own and 43 us
// together, i.e. what costs is LDS contents that CHANGE between chunks; scripts/wgrad_data_dependence.py: the full kernel is 13 % faster whenever
// one of its operands is all zeros, at an unchanged 2.39 GHz.)

#ifdef PIVP_WG_STAMPS   // per-block phase stamps of the ConvLSTM weight-gradient kernels (scripts/wgrad_stamps.py): [block][entry, loop start, loop end, done] in
// 10 ns ticks (constant-rate counter), then the same points on the shader-cycle counter: cycles / wall = the clock the chip holds in that phase
__device__ long long pivp_wg_stamps[2048 * 8];
#define WG_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < 2048) { const int sb_ = blockIdx.y * gridDim.x + blockIdx.x; \
    pivp_wg_stamps[sb_ * 4 + (i)] = (long long)wall_clock64(); pivp_wg_stamps[2048 * 4 + sb_ * 4 + (i)] = (long long)clock64(); } } while (0)
#else
#define WG_STAMP(i)
#endif

namespace pivp {

constexpr int WG_PIX = 32;     // pixels (GEMM K) per chunk
constexpr int WG_CI = 64;      // input channels per block
constexpr int WG_XP = WG_CI + 4;   // LDS pitches (floats); +4 keeps 16-B alignment of rows

// NT: 32-column MFMA tiles per wave; the block covers 64 input channels x 64*NT output columns.  NT = 1 for layers with at most
// 64 output channels (enc1, enc2, enc6): with 128 columns half of the waves multiplied zero padding.
template <int NT>
__global__ __launch_bounds__(256, 1) void igemm_wgrad_kernel(const WgradDesc d) {
    constexpr int WG_N = 64 * NT;      // output columns per block
    constexpr int WG_YP = WG_N + 4;
    constexpr int YL = 16 * NT;        // float4 lanes per dY row
    constexpr int YRP = 256 / YL;      // dY rows per staging pass
    constexpr int NYL = WG_PIX / YRP;  // passes
    __shared__ __attribute__((aligned(16))) float xs[2][WG_PIX * WG_XP];
    __shared__ __attribute__((aligned(16))) float ys[2][WG_PIX * WG_YP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;             // wave tile: 32 ci x 32*NT n
    const int l31 = lane & 31, half = lane >> 5;
    const int ncb = (d.cin + WG_CI - 1) / WG_CI, nnb = (d.N + WG_N - 1) / WG_N;
    int bid = blockIdx.x;
    const int nb = bid % nnb; bid /= nnb;
    const int cb = bid % ncb; bid /= ncb;
    const int tap = bid;                                  // 0 .. ksize*ksize-1
    const int ky = tap / d.ksize, kx = tap - ky * d.ksize;
    const int ci0 = cb * WG_CI, n0 = nb * WG_N;
    const int nsplit = gridDim.y, split = blockIdx.y;
    const int cpt = (d.M + WG_PIX - 1) / WG_PIX;          // chunks per timestep
    const int tcount = d.tcount > 1 ? d.tcount : 1;
    const int nchunks_total = cpt * tcount;
    const int c_begin = (int)((long)nchunks_total * split / nsplit), c_end = (int)((long)nchunks_total * (split + 1) / nsplit);
    const int HWg = d.Hg * d.Wg;
    // offsets of the two operands relative to the anchor
    const int ady = d.deconv ? 0 : ky - d.pad, adx = d.deconv ? 0 : kx - d.pad;          // X side (after anchor*sa)
    const int bdy = d.deconv ? ky - 1 : 0, bdx = d.deconv ? kx - 1 : 0;                   // dY side (after anchor*sb)
    const int sa = d.deconv ? 1 : d.stride, sb = d.deconv ? 2 : 1;
    constexpr unsigned OOB = 0xC0000000u;

    // staging: X tile 32 pix x 64 ci = 512 float4 (2 per thread), dY tile 32 pix x 128 n = 1024 float4 (4 per thread)
    const int xr = tid >> 4, xc = (tid & 15) * 4;        // rows xr, xr+16
    const int yr = tid / YL, yc = (tid % YL) * 4;        // rows yr, yr+YRP, ...
    // this thread's 4 X columns live in x0 (channels [0,c0)) or x1 ([c0,c0+c1)); a 64-channel block may straddle
    // the two, so both descriptors are read and the one that does not hold the columns gets an out-of-range offset
    // (hardware returns 0) -- the sum is the value, no per-lane descriptor select.
    const int xci = ci0 + xc;
    const bool xcol_ok = xci < d.cin;
    const bool in0 = xci < d.c0;
    const bool ycol_ok = (n0 + yc) < d.N;
    f32x4 rxs[2][2], rYs[2][NYL];   // two register sets: chunk c+2 is loaded during chunk c, written to LDS at the end of chunk c+1
    // anchor coordinates without per-load divisions: when chunks never straddle a sample and the anchor-grid width is a
    // multiple or a divisor of 32 (every map of this model), row r of a chunk sits at a fixed (row, column) offset from the
    // chunk's first anchor.  The first version divided twice per load: ~420 of the ~600 instructions between two chunks'
    // 32 MFMAs.
    const bool fast = HWg % WG_PIX == 0 && (d.Wg % WG_PIX == 0 || WG_PIX % d.Wg == 0);
    int xdr[2], xdx[2], ydr[NYL], ydx[NYL];
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int r = xr + 16 * j; xdr[j] = d.Wg >= WG_PIX ? 0 : r / d.Wg; xdx[j] = r - xdr[j] * d.Wg; }
#pragma unroll
    for (int j = 0; j < NYL; ++j) { const int r = yr + YRP * j; ydr[j] = d.Wg >= WG_PIX ? 0 : r / d.Wg; ydx[j] = r - ydr[j] * d.Wg; }
    auto issue = [&](auto SET, int chunk) {
        f32x4 (&rx)[2] = rxs[decltype(SET)::value];
        f32x4 (&rY)[NYL] = rYs[decltype(SET)::value];
        // (timestep of the batch, chunk inside it): block-uniform; that timestep's descriptors are rebuilt here (scalar work)
        const int tj = __builtin_amdgcn_readfirstlane(tcount > 1 ? chunk / cpt : 0);
        chunk -= tj * cpt;
        const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(d.x0) + (long long)tj * d.ts_x0), 0, d.bytes0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(d.c1 ? d.x1 : d.x0) + (long long)tj * (d.c1 ? d.ts_x1 : d.ts_x0)), 0,
            d.c1 ? d.bytes1 : d.bytes0, 0x00020000);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(d.dy) + (long long)tj * d.ts_dy), 0, d.bytesy, 0x00020000);
        const int m0 = __builtin_amdgcn_readfirstlane(chunk * WG_PIX);
        const int b0 = m0 / HWg, rem0 = m0 - b0 * HWg, ay0 = rem0 / d.Wg, ax0 = rem0 - ay0 * d.Wg;   // wave-uniform
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = chunk * WG_PIX + xr + 16 * j;
            unsigned off0 = OOB, off1 = OOB;
            if (m < d.M && xcol_ok) {
                int b, ay, ax;
                if (fast) { b = b0; ay = ay0 + xdr[j]; ax = ax0 + xdx[j]; }
                else { b = m / HWg; const int rem = m - b * HWg; ay = rem / d.Wg; ax = rem - ay * d.Wg; }
                const int iy = ay * sa + ady, ix = ax * sa + adx;
                if ((unsigned)iy < (unsigned)d.Hx && (unsigned)ix < (unsigned)d.Wx) {
                    const int pix = (b * d.Hx + iy) * d.Wx + ix;
                    if (in0) off0 = (unsigned)((pix * d.ld0 + xci) * 4);
                    else     off1 = (unsigned)((pix * d.ld1 + xci - d.c0) * 4);
                }
            }
            f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx0, off0, 0, 0));
            if (d.c1) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx1, off1, 0, 0));
            rx[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NYL; ++j) {
            const int m = chunk * WG_PIX + yr + YRP * j;
            unsigned off = OOB;
            if (m < d.M && ycol_ok) {
                int b, ay, ax;
                if (fast) { b = b0; ay = ay0 + ydr[j]; ax = ax0 + ydx[j]; }
                else { b = m / HWg; const int rem = m - b * HWg; ay = rem / d.Wg; ax = rem - ay * d.Wg; }
                const int oy = ay * sb + bdy, ox = ax * sb + bdx;
                if ((unsigned)oy < (unsigned)d.Hy && (unsigned)ox < (unsigned)d.Wy)
                    off = (unsigned)((((b * d.Hy + oy) * d.Wy + ox) * d.ldy + n0 + yc) * 4);
            }
            rY[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, off, 0, 0));
        }
    };
    auto store = [&](auto SET, int buf) {
        const f32x4 (&rx)[2] = rxs[decltype(SET)::value];
        const f32x4 (&rY)[NYL] = rYs[decltype(SET)::value];
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(&xs[buf][(xr + 16 * j) * WG_XP + xc]) = rx[j];
#pragma unroll
        for (int j = 0; j < NYL; ++j) *reinterpret_cast<f32x4*>(&ys[buf][(yr + YRP * j) * WG_YP + yc]) = rY[j];
    };
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // Bias gradient = column sums of dY, on the side: the dY values a wave feeds its MFMAs pass through its registers anyway.  A conv's
    // dY tile is the same for all taps: the blocks of tap 0 / channel block 0 see every dY pixel of their slice once.  The transposed conv
    // gathers dY at 2a - 1 + k: taps (1,1), (1,2), (2,1), (2,2) visit each of the four output parities exactly once and never leave the
    // map.  (A bias_grad_kernel launch per conv and timestep before: 44 launches per train step.)
    const bool do_bias = d.db != nullptr && cb == 0 && wm == 0 &&
                         (d.deconv ? (tap == 4 || tap == 5 || tap == 7 || tap == 8) : tap == 0);
    float bsum[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bsum[t] = 0.f;
    if (c_begin < c_end) {
        using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
        auto mma = [&](int buf) {
            const float* X = &xs[buf][wm * 32 + l31];
            const float* Y = &ys[buf][wn * 32 * NT + l31];
#pragma unroll
            for (int k = 0; k < WG_PIX; k += 2) {
                const float a = X[(k + half) * WG_XP];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float yv = Y[(k + half) * WG_YP + 32 * t];
                    bsum[t] += yv;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, yv, acc[t], 0, 0, 0);
                }
            }
        };
        issue(S0{}, c_begin);
        store(S0{}, 0);
        if (c_begin + 1 < c_end) issue(S1{}, c_begin + 1);
        __syncthreads();
        int c = c_begin;
        for (; c + 1 < c_end; c += 2) {          // chunks c (LDS buffer 0) and c+1 (buffer 1)
            if (c + 2 < c_end) issue(S0{}, c + 2);
            mma(0);
            store(S1{}, 1);
            __syncthreads();
            if (c + 3 < c_end) issue(S1{}, c + 3);
            mma(1);
            if (c + 2 < c_end) store(S0{}, 0);
            __syncthreads();
        }
        if (c < c_end) mma(0);
    }
    if (do_bias) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float v = bsum[t] + __shfl_xor(bsum[t], 32, 64);
            const int n = n0 + wn * 32 * NT + t * 32 + l31;
            if (half == 0 && n < d.N) atomicAdd(d.db + n, v);
        }
    }
    // acc[t][r]: row i = ci (= (r&3) + 8*(r>>2) + 4*half), column j = n (= l31)
    if (d.part) {      // this block's own [64][WG_N] slot: plain read-modify-write, rows of 32 lanes are 128 contiguous bytes
        float* pt = d.part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (64 * WG_N);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* q = pt + (wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * WG_N + wn * 32 * NT + t * 32 + l31;
                *q = d.part_overwrite ? acc[t][r] : *q + acc[t][r];
            }
        return;
    }
    // direct path: fp32 atomics into the packed gradient [tap][ci/32][n][ci%32]
    const int wtap = tap;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + wn * 32 * NT + t * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (n < d.N && ci < d.cin) {
                float* g = d.dw + (((size_t)wtap * (d.wcin >> 5) + (ci >> 5)) * d.N + n) * 32 + (ci & 31);
                atomicAdd(g, acc[t][r]);
            }
        }
    }
}

// dw += sum over the pixel splits of the per-block partial tiles (see WgradDesc::part).  Thread = one element of one tile (grid: element
// chunks x tiles); the nsplit loads of a thread are independent and, across a wave, contiguous.  (One block per tile, the first
// version, took 154 us for 9-18 blocks.)
__global__ __launch_bounds__(256) void igemm_wgrad_reduce_kernel(const WgradDesc d, int wg_n, int nsplit) {
    const int tiles = gridDim.y, tile = blockIdx.y;
    const int ncb = (d.cin + WG_CI - 1) / WG_CI, nnb = (d.N + wg_n - 1) / wg_n;
    int bid = tile;
    const int nb = bid % nnb; bid /= nnb;
    const int cb = bid % ncb; bid /= ncb;
    const int tap = bid;
    const int tile_floats = 64 * wg_n;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= tile_floats) return;
    const int row = e / wg_n, col = e - row * wg_n;
    const int ci = cb * WG_CI + row, n = nb * wg_n + col;
    if (ci >= d.cin || n >= d.N) return;
    const float* src = d.part + (size_t)tile * tile_floats + e;
    const size_t stride = (size_t)tiles * tile_floats;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int sp = 0;
    for (; sp + 3 < nsplit; sp += 4) {
        a0 += src[(size_t)sp * stride]; a1 += src[(size_t)(sp + 1) * stride];
        a2 += src[(size_t)(sp + 2) * stride]; a3 += src[(size_t)(sp + 3) * stride];
    }
    for (; sp < nsplit; ++sp) a0 += src[(size_t)sp * stride];
    d.dw[(((size_t)tap * (d.wcin >> 5) + (ci >> 5)) * d.N + n) * 32 + (ci & 31)] += (a0 + a1) + (a2 + a3);
}

// ---------------------------------------------------------------------------------------------------------
// Fast path for the ConvLSTM weight gradient (5x5, stride 1, pad 2): 93 % of all wgrad flops.
//   * one WAVE = one worker on 32-pixel chunks (32 consecutive pixels of image rows): it stages, for kernel row ky,
//     an X strip with a 2-pixel x-halo ([R rows][SW+4][32 ci]) and the dY tile ([32 pixels][32*NTW n]) into its own
//     double-buffered LDS region, then runs the 5 taps kx = 0..4 of that kernel row against the SAME dY fragments:
//     16 k-steps x 5 taps x NTW tiles = 80*NTW MFMAs per chunk for ~12 16-byte loads per lane, and no block barrier
//     anywhere in the main loop (a wave only ever reads what it wrote itself);
//   * the 4 waves of a block work on the same (ky, 32 ci, 32*NTW n) tile over interleaved chunks; their accumulators are
//     summed through LDS and the block issues ONE set of atomics, transposed through LDS so that a wave-instruction adds
//     two contiguous 128-B segments of the K-inner packed gradient (full atomic rate) instead of 64 scattered words.
// ---------------------------------------------------------------------------------------------------------
// (Round 2 also spread the staging over the chunk's 16 k-steps, one load and one ds_write per k-step in the shadow of its ten MFMAs, as
// igemm_f32.hip does: 159.6 vs 159.8 us per launch, no gain -- and 213 us with a run-time `if (c + 8 < c_end)` around the pieces, which
// makes hipcc drain vmcnt in front of every ds_write.  Interleaving at MFMA granularity (sched_group_barrier: 1 MFMA, 3 VALU, ten times) gave
// 160.0 us.  The pipe's idle third (MFMA busy 0.65) is not the staging's issue slots.  The block form of issue() / store() stays.)
// Two blocks per CU: with NTW = 2 a wave carries 160 accumulator registers (344 VGPRs in all) and a block ~100 KB of LDS, so every SIMD
// holds ONE wave and each of its waits is exposed (MFMA busy 0.58-0.65).  NTW = 1 halves both (180 VGPRs, 70-80 KB): two blocks = two
// waves per SIMD, for 1.36x the operand traffic per MFMA.  Per launch at B = 32 (us, NTW 2 -> 1): lstm1/2 130 -> 119, lstm3 99 -> 90,
// lstm4 130 -> 118, lstm5 101 -> 92, lstm6 183 -> 169, lstm7 238 -> 224; train step 31.0 -> 30.7 ms.  (Keeping NTW = 2 and dropping to one
// LDS buffer + one staging register set per wave instead spilled 76-97 VGPRs: 32.3 -> 35.0 ms.)  A second round of blocks costs 8-9 us
// per launch (prologue + block reduction + atomics + the first loads' latency): the grid stays at one round of resident blocks.
// (Round 4 also built THREE resident blocks per CU -- one LDS buffer and one staging register set per wave, <= 168 registers, 42 KB of LDS: slower on
// every layer, profiles/r04/NOTES.md 5; and the NTW = 2 form stayed selectable until round 5.  Both are in the history.)
template <int SW>   // SW: pixels of one image row inside a chunk (min(W, 32))
__global__ __launch_bounds__(256, 2) void wgrad5x5_kernel(const WgradDesc d) {
    constexpr int NTW = 1;                      // 32-column tiles per wave
    constexpr int R = 32 / SW;                  // image rows per chunk
    constexpr int SP = R * (SW + 4);            // strip pixels
    constexpr int XP = 32, YP = 32 * NTW;       // LDS row lengths (floats): lane-contiguous reads, no padding needed
    constexpr int WBUF = SP * XP + 32 * YP;     // floats per wave per buffer
    constexpr int NBUF = 2;                     // LDS buffers per wave
    constexpr int NACC = 5 * NTW;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    WG_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int ncb = d.cin >> 5, nnb = d.N / (32 * NTW);
    int bid = blockIdx.x;
    const int nb = bid % nnb; bid /= nnb;
    const int cb = bid % ncb; bid /= ncb;
    const int ky = bid;                          // 0..4
    const int ci0 = cb * 32, n0 = nb * 32 * NTW;
    const int HWg = d.Hg * d.Wg, Wd = d.Wg, Hd = d.Hg;
    const int cpt = d.M / 32;                     // chunks per timestep
    const int tcount = d.tcount > 1 ? d.tcount : 1;
    const int nchunks_total = cpt * tcount;
    const int nsplit = gridDim.y, split = blockIdx.y;
    const int c_begin = (int)((long)nchunks_total * split / nsplit), c_end = (int)((long)nchunks_total * (split + 1) / nsplit);
    constexpr unsigned OOB = 0xC0000000u;
    const bool from0 = ci0 < d.c0;               // a 32-channel block never straddles the two sources (c0 % 32 == 0)
    const char* xbase = reinterpret_cast<const char*>(from0 ? d.x0 : d.x1);
    const long long xts = from0 ? d.ts_x0 : d.ts_x1;
    const int xbytes = from0 ? d.bytes0 : d.bytes1;
    const int xld = from0 ? d.ld0 : d.ld1;
    const int xc0 = from0 ? ci0 : ci0 - d.c0;
    float* wbase = sm + wave * NBUF * WBUF;

    // staging roles inside a wave: X strip: 8 lanes per strip pixel (float4 of 4 channels), 8 pixels per pass
    constexpr int NXP = (SP + 7) / 8;
    // dY tile: YP/4 lanes per pixel, 64/(YP/4) pixels per pass
    constexpr int YL = YP / 4, YPP = 64 / YL, NYP = 32 / YPP;
    const int xl = lane & 7, xq = lane >> 3;
    const int yl = lane % YL, yq = lane / YL;
    // two register sets: the loads of chunk c+8 are issued at the top of chunk c and written to LDS at the END of chunk c+4, so
    // they have two chunks (~20k cycles) to arrive; with one set (issue at the top, ds_write at the end of the same chunk) the
    // ds_writes waited for loads every chunk: the ablations price that wait at 43 of lstm7's 265 us.
    f32x4 rxs[2][NXP], rys[2][NYP];
    auto issue = [&](auto SET, int chunk) {
        f32x4 (&rx4)[NXP] = rxs[decltype(SET)::value];
        f32x4 (&ry4)[NYP] = rys[decltype(SET)::value];
        // chunk -> (timestep of the batch, chunk inside it); wave-uniform.  The descriptors of that timestep's operands are rebuilt
        // here (scalar work): byte strides between timesteps can exceed a descriptor's 32-bit offsets.
        const int tj = __builtin_amdgcn_readfirstlane(tcount > 1 ? chunk / cpt : 0);
        const int p0 = (chunk - tj * cpt) * 32;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xbase + (long long)tj * xts), 0, xbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(d.dy) + (long long)tj * d.ts_dy), 0, d.bytesy, 0x00020000);
        const int b = p0 / HWg, rem = p0 - b * HWg, y0 = rem / Wd, x0 = rem - y0 * Wd;
#pragma unroll
        for (int j = 0; j < NXP; ++j) {
            const int sp = xq + 8 * j;                     // strip pixel
            unsigned off = OOB;
            if (sp < SP) {
                const int r = sp / (SW + 4), xx = sp - r * (SW + 4);
                const int iy = y0 + r + ky - 2, ix = x0 + xx - 2;
                if ((unsigned)iy < (unsigned)Hd && (unsigned)ix < (unsigned)Wd)
                    off = (unsigned)((((b * Hd + iy) * Wd + ix) * xld + xc0 + xl * 4) * 4);
            }
            rx4[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < NYP; ++j) {
            const int pix = yq + YPP * j;
            const unsigned off = (unsigned)(((p0 + pix) * d.ldy + n0 + yl * 4) * 4);
            ry4[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, off, 0, 0));
        }
    };
    auto store = [&](auto SET, int buf) {
        f32x4 (&rx4)[NXP] = rxs[decltype(SET)::value];
        f32x4 (&ry4)[NYP] = rys[decltype(SET)::value];
        float* xs = wbase + buf * WBUF;
        float* ys = xs + SP * XP;
#pragma unroll
        for (int j = 0; j < NXP; ++j) {
            const int sp = xq + 8 * j;
            if (sp < SP) *reinterpret_cast<f32x4*>(xs + sp * XP + xl * 4) = rx4[j];
        }
#pragma unroll
        for (int j = 0; j < NYP; ++j) *reinterpret_cast<f32x4*>(ys + (yq + YPP * j) * YP + yl * 4) = ry4[j];
    };
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float bsum[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) bsum[t] = 0.f;
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    int c = c_begin + wave;
    if (c < c_end) { issue(S0{}, c); store(S0{}, 0); }
    if (c + 4 < c_end) issue(S1{}, c + 4);
    int buf = 0;
    auto iter = [&](auto SET) {     // chunk c from LDS buffer `buf`; SET = register set that receives chunk c+8
        using OTHER = std::integral_constant<int, decltype(SET)::value ^ 1>;
        if (c + 8 < c_end) issue(SET, c + 8);
        const float* xs = wbase + buf * WBUF + l31;
        const float* ys = xs - l31 + SP * XP + l31;
        // operands of k-step s2+1 are read from LDS BEFORE the 5*NTW MFMAs of k-step s2 are issued, and the order is pinned:
        // left to itself hipcc reads each operand right in front of its MFMA and waits lgkmcnt(0) twice per k-step
        // (two exposed LDS round trips per 10 MFMAs: the kernel ran at 70 TFLOP/s)
        float av[2][5], bv[2][NTW];   // (bsum: running column sums of the dY values this lane feeds the MFMAs = bias gradient)
        auto read_step = [&](auto S2, int slot) {
            constexpr int s2 = decltype(S2)::value;
            constexpr int sidx0 = (2 * s2 / SW) * (SW + 4) + (2 * s2 % SW);   // strip index of (pixel 2*s2, kx = 0)
#pragma unroll
            for (int t = 0; t < NTW; ++t) bv[slot][t] = ys[(2 * s2 + half) * YP + t * 32];
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) av[slot][kx] = xs[(sidx0 + half + kx) * XP];
        };
        auto kstep = [&](auto S2) {
            constexpr int s2 = decltype(S2)::value, cur = s2 & 1;
            if constexpr (s2 + 1 < 16) read_step(std::integral_constant<int, s2 + 1>{}, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NTW; ++t) bsum[t] += bv[cur][t];
#pragma unroll
            for (int kx = 0; kx < 5; ++kx)
#pragma unroll
                for (int t = 0; t < NTW; ++t)
                    acc[kx * NTW + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][kx], bv[cur][t], acc[kx * NTW + t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        read_step(std::integral_constant<int, 0>{}, 0);
        kstep(std::integral_constant<int, 0>{}); kstep(std::integral_constant<int, 1>{}); kstep(std::integral_constant<int, 2>{});
        kstep(std::integral_constant<int, 3>{}); kstep(std::integral_constant<int, 4>{}); kstep(std::integral_constant<int, 5>{});
        kstep(std::integral_constant<int, 6>{}); kstep(std::integral_constant<int, 7>{}); kstep(std::integral_constant<int, 8>{});
        kstep(std::integral_constant<int, 9>{}); kstep(std::integral_constant<int, 10>{}); kstep(std::integral_constant<int, 11>{});
        kstep(std::integral_constant<int, 12>{}); kstep(std::integral_constant<int, 13>{}); kstep(std::integral_constant<int, 14>{});
        kstep(std::integral_constant<int, 15>{});
        if (c + 4 < c_end) store(OTHER{}, buf ^ 1);
        buf ^= 1;
        c += 4;
    };
    WG_STAMP(1);
    while (c < c_end) {
        iter(S0{});
        if (c < c_end) iter(S1{});
    }
    WG_STAMP(2);
    // ---- bias gradient: the blocks of kernel row 2 / channel block 0 have fed every dY element of their pixel range through
    // the MFMAs exactly once; lane (n, half) holds the sum over its half's pixels ---------------------------------------
    if (d.db && ky == 2 && cb == 0) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const float v = bsum[t] + __shfl_xor(bsum[t], 32, 64);
            if (half == 0) atomicAdd(d.db + n0 + t * 32 + l31, v);
        }
    }
    // ---- block reduction of the 4 workers + transposed atomics -------------------------------------------------
    // LDS image of one worker's result: [tap kx][tile t][n 0..31][ci 0..31 (+1 pad)].  The n rows are 33 floats apart: with a
    // 32-float pitch the 32 lanes of an accumulator register (same ci, n = lane) hit one bank, a 32-way conflict on every one of the
    // 160 stores and loads of each pass.
    __syncthreads();                               // every wave is done with its staging buffers
    constexpr int IT = 32 * 33;                    // floats per 32 x 32 tile image
    constexpr int IMG = NACC * IT;
    auto put = [&](float* img) {
#pragma unroll
        for (int t = 0; t < NACC; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) img[t * IT + l31 * 33 + (r & 3) + 8 * (r >> 2) + 4 * half] = acc[t][r];
    };
    auto add = [&](const float* img) {
#pragma unroll
        for (int t = 0; t < NACC; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] += img[t * IT + l31 * 33 + (r & 3) + 8 * (r >> 2) + 4 * half];
    };
    if (wave >= 2) put(sm + (wave - 2) * IMG);
    __syncthreads();
    if (wave < 2) add(sm + wave * IMG);
    __syncthreads();
    if (wave == 1) put(sm);
    __syncthreads();
    if (wave == 0) { add(sm); }
    __syncthreads();
    if (wave == 0) put(sm);
    __syncthreads();
    // packed gradient [tap][wcin/32][N][32]: for tap (ky,kx), tile t: rows n0 + t*32 + n, 32 contiguous ci each
    for (int i = tid; i < NACC * 1024; i += 256) {
        const int t = i >> 10, rem = i & 1023, n = rem >> 5, ci = rem & 31;
        const int kx = t / NTW, tt = t - kx * NTW;
        const int tap = ky * 5 + kx;
        float* g = d.dw + (((size_t)tap * (d.wcin >> 5) + cb) * d.N + n0 + tt * 32 + n) * 32 + ci;
        // (these contiguous atomics cost 2 % of the kernel, 160 -> 157 us on lstm7 without them, unlike the generic kernel's scattered ones,
        // 85 -> 31 us, which is why only that one got per-block partial sums)
        atomicAdd(g, sm[t * IT + n * 33 + ci]);
    }
#ifdef PIVP_WG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WG_STAMP(3);
#endif
}

template <int SW>
static int launch_wgrad5x5(const WgradDesc& d, hipStream_t s) {
    constexpr int NTW = 1, OCC = 2;      // 32-column tiles per wave, resident blocks per CU (the kernel's launch bounds)
    constexpr int R = 32 / SW, SP = R * (SW + 4);
    constexpr int WBUF = SP * 32 + 32 * 32 * NTW;
    constexpr int IMG = 5 * NTW * 32 * 33;
    constexpr int NBUF = 2;
    constexpr int lds_floats = (4 * NBUF * WBUF > 2 * IMG) ? 4 * NBUF * WBUF : 2 * IMG;
    constexpr int lds_bytes = lds_floats * 4;
    static PerDeviceOnce once;
    if (pivp_ensure_dyn_lds(once, reinterpret_cast<const void*>(&wgrad5x5_kernel<SW>), lds_bytes) != PIVP_OK) return PIVP_ERR_LAUNCH;
    const int tiles = 5 * (d.cin / 32) * (d.N / (32 * NTW));
    const int chunks = d.M / 32 * (d.tcount > 1 ? d.tcount : 1);
    // One block per CU is resident (~100 KB of LDS), so the grid is sized to whole rounds of the chip's CUs: the first
    // version asked for "about 512" blocks and got 520-600, i.e. a third round that ran 8-88 blocks on 256 CUs (lstm7: 264 us
    // for 171 us of MFMA work).  Take the fewest rounds (1..3) whose last round is at least 90 % full.
    const int cus = pivp_cu_count() * OCC;   // resident blocks
    int nsplit = 1;
    double best = 0.0;
    for (int r = 1; r <= 3; ++r) {
        int ns = (cus * r) / tiles;
        if (ns > chunks / 16) ns = chunks / 16;          // >= 4 chunks per wave
        if (ns < 1) ns = 1;
        const long blocks = (long)tiles * ns;
        const double fill = (double)blocks / (double)(((blocks + cus - 1) / cus) * cus);
        if (fill > best + 0.02) { best = fill; nsplit = ns; }
        if (best >= 0.9) break;
    }
    hipLaunchKernelGGL((wgrad5x5_kernel<SW>), dim3(tiles, nsplit), dim3(256), lds_bytes, s, d);
    return PIVP_LAUNCH_STATUS();
}

static bool takes_fast_path(const WgradDesc& d) {
    return !d.deconv && d.ksize == 5 && d.pad == 2 && d.stride == 1 && d.M % 32 == 0 && d.N % 64 == 0 &&
           (d.Wg == 8 || d.Wg == 16 || d.Wg % 32 == 0) && d.Hx == d.Hy && d.Wx == d.Wy;
}
// grid of the generic kernel: a function of the descriptor alone, so a launch, its partial buffer and its reduction agree
static void generic_grid(const WgradDesc& d, int& wg_n, int& tiles, int& nsplit) {
    wg_n = d.N <= 64 ? 64 : 128;
    const int ncb = (d.cin + WG_CI - 1) / WG_CI, nnb = (d.N + wg_n - 1) / wg_n;
    tiles = d.ksize * d.ksize * ncb * nnb;
    const int chunks = (d.M + WG_PIX - 1) / WG_PIX;      // of ONE timestep: a batched launch, a single one and the reduction must agree on the planes
    // direct path: every block ends with a tile of atomics (64 x 128), so no more pixel splits than fill the chip twice (3 blocks fit a
    // CU) and >= 8 chunks (256 pixels) per block (enc4 with 4: 63 -> 94 us, atomics); kept for the partial-sum path, where a split
    // costs 16-32 KB of traffic instead
    // (round 5, call 15: a target of 256 / 384 / 1024 blocks instead of 512 gives config 3 11.50 / 11.27 / 11.27 ms against 11.18-11.22,
    // fp32 train 28.02 against 27.85 with 256: 512 stays)
    nsplit = (512 + tiles - 1) / tiles;
    if (nsplit > chunks / 8) nsplit = chunks / 8;
    if (nsplit < 1) nsplit = 1;
}

int igemm_wgrad(const WgradDesc& d, hipStream_t s, int* bias_done) {
    if (bias_done) *bias_done = 0;
    PIVP_CHECK_ARG(d.x0 && d.dy && d.dw && d.c0 > 0 && d.c0 % 32 == 0 && d.c1 >= 0 && d.c1 % 32 == 0 && (d.c1 == 0 || d.x1));
    PIVP_CHECK_ARG(d.cin == d.c0 + d.c1 && d.wcin >= d.cin && d.wcin % 32 == 0 && d.N > 0 && d.N % 32 == 0);
    PIVP_CHECK_ARG(d.M == d.B * d.Hg * d.Wg && d.M > 0 && d.ksize >= 1 && d.ksize <= 7);
    PIVP_CHECK_ARG(d.bytes0 > 0 && d.bytesy > 0 && (d.c1 == 0 || d.bytes1 > 0));
    if (d.part && wgrad5x5p_ok(d)) {        // ConvLSTM 5x5, partial-slot form (round 6): no atomics; the column sums ride along
        if (bias_done) *bias_done = d.db ? 1 : 0;
        return wgrad5x5p(d, s);
    }
    if (takes_fast_path(d)) {
        if (bias_done) *bias_done = d.db ? 1 : 0;
        return d.Wg == 8 ? launch_wgrad5x5<8>(d, s) : d.Wg == 16 ? launch_wgrad5x5<16>(d, s) : launch_wgrad5x5<32>(d, s);
    }
    if (d.part && wgrad3x3s2_ok(d)) {      // stride-2 3x3: all nine taps from one staging (wgrad3x3s2.hip); the column sums ride along
        if (bias_done) *bias_done = d.db ? 1 : 0;
        return wgrad3x3s2(d, s);
    }
    int wg_n, tiles, nsplit;
    generic_grid(d, wg_n, tiles, nsplit);
    // the column sums ride along when the tap set qualifies: a 3x3 conv, or the transposed 3x3 s2 conv (see the kernel)
    const bool bias_here = d.db && d.ksize == 3 && (!d.deconv || (d.Hy == 2 * d.Hx && d.Wy == 2 * d.Wx));
    WgradDesc dd = d;
    if (!bias_here) dd.db = nullptr;
    if (bias_done) *bias_done = bias_here ? 1 : 0;
    if (wg_n == 64) hipLaunchKernelGGL(igemm_wgrad_kernel<1>, dim3(tiles, nsplit), dim3(256), 0, s, dd);
    else hipLaunchKernelGGL(igemm_wgrad_kernel<2>, dim3(tiles, nsplit), dim3(256), 0, s, dd);
    return PIVP_LAUNCH_STATUS();
}

long long igemm_wgrad_part_floats(const WgradDesc& d) {
    if (wgrad5x5p_ok(d)) return wgrad5x5p_part_floats(d);
    if (takes_fast_path(d)) return 0;
    if (wgrad3x3s2_ok(d)) return wgrad3x3s2_part_floats(d);
    int wg_n, tiles, nsplit;
    generic_grid(d, wg_n, tiles, nsplit);
    return (long long)tiles * nsplit * 64 * wg_n;
}

int igemm_wgrad_reduce(const WgradDesc& d, hipStream_t s) {
    PIVP_CHECK_ARG(d.part && d.dw);
    if (wgrad5x5p_ok(d)) return wgrad5x5p_reduce(d, s);
    PIVP_CHECK_ARG(!takes_fast_path(d));
    if (wgrad3x3s2_ok(d)) return wgrad3x3s2_reduce(d, s);
    int wg_n, tiles, nsplit;
    generic_grid(d, wg_n, tiles, nsplit);
    hipLaunchKernelGGL(igemm_wgrad_reduce_kernel, dim3((64 * wg_n + 255) / 256, tiles), dim3(256), 0, s, d, wg_n, nsplit);
    return PIVP_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------------
// Re-pack for the data gradient: W packed [tap][Cin/32][N][32]  ->  Wt packed [tap'][N/32][Cin][32] with
// tap' = flipped tap when `flip` (stride-1 conv: dX = conv(dY, W flipped, in/out swapped)) or the same tap
// (stride-2 conv <-> transposed conv are exact adjoints of each other on the same tap index).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void repack_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                               int taps, int cin, int N, int flip) {
    __shared__ float t[32][33];
    const int tap = blockIdx.z, cc = blockIdx.y, nc = blockIdx.x;   // 32 ci x 32 n tile
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = w + (((size_t)tap * (cin >> 5) + cc) * N + nc * 32) * 32;      // [n 0..31][ci 0..31]
    for (int r = ty; r < 32; r += 8) t[r][tx] = src[r * 32 + tx];                      // t[n][ci]
    __syncthreads();
    const int tap2 = flip ? taps - 1 - tap : tap;
    float* dst = wt + (((size_t)tap2 * (N >> 5) + nc) * cin + cc * 32) * 32;           // [ci 0..31][n 0..31]
    for (int r = ty; r < 32; r += 8) dst[r * 32 + tx] = t[tx][r];
}

int repack_transpose(const float* w, float* wt, int taps, int cin, int N, int flip, hipStream_t s) {
    PIVP_CHECK_ARG(w && wt && taps > 0 && cin > 0 && cin % 32 == 0 && N > 0 && N % 32 == 0);
    hipLaunchKernelGGL(repack_transpose_kernel, dim3(N / 32, cin / 32, taps), dim3(256), 0, s, w, wt, taps, cin, N, flip);
    return PIVP_LAUNCH_STATUS();
}

}  // namespace pivp

#ifdef PIVP_WG_STAMPS
extern "C" int pivp_debug_wg_stamps(long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pivp_wg_stamps), sizeof(long long) * n) == hipSuccess ? 0 : -2;
}
#endif
