// ConvLSTM weight gradient with bf16 operands and fp32 accumulation (the bf16 precision mode's form of wgrad5x5_kernel,
// igemm_wgrad.hip):   dW[tap][ci][n] += sum over pixels m of  X[m + tap][ci] * dG[m][n]      (5x5, stride 1, pad 2)
// i.e. per tap a GEMM whose REDUCTION runs over pixels.  On v_mfma_f32_32x32x16_bf16 both operands then need 8 consecutive
// k = 8 pixels per lane for one row (n) / column (ci), while the activations are pixel-major [pixel][channel].  gfx950's
// transposing LDS read does that for free: ds_read_b64_tr_b16 takes, per group of 16 lanes, a block of 4 image rows (pixels)
// x 16 columns (channels) and hands lane i column i of the 4 rows, so two of them deliver the 8 k values of an MFMA operand
// straight from the pixel-major image -- and a tap only shifts the pixel (row) address, so no alignment case arises.
//
// A block owns one kernel row ky (5 taps kx), 32 input channels, 128 gate columns (wave w: columns 32 w .. 32 w + 31) and a
// slice of the pixel tiles; MFMA rows = n, columns = ci, so that the epilogue's atomic adds into the K-inner packed gradient
// [tap][ci / 32][n][32] are 128-B contiguous per half-wave.  Per tile of 128 pixels (8 x 16, or two 8 x 8 images) it stages
// dG [128 px][128 n] and the X strip of kernel row ky [8 rows][16 + 4 columns][32 ci] as bf16 (fp32 in HBM, rounded on the way
// in; out-of-image pixels = the zeros of an out-of-range buffer load) and runs 8 k-steps x 5 taps against the same dG
// fragments.  The next tile's loads are in flight in registers while the current one is multiplied.
// Slices meet in dW by fp32 atomic adds (gradients accumulate until the host clears them, as in the fp32 kernel).
#include <stdlib.h>
#include <type_traits>

#include "pivp_kernels.h"

namespace pivp {

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Row pitches for the transposing reads.  The 32 lanes of a ds_read_b64_tr_b16 phase (banks = dword address mod 64) cover 4 consecutive
// image rows x 32 columns = 4 x 64 B, so they are conflict-free iff the row pitch is 16 dwords (mod 64): 256 + 64 B for the dG image, and
// exactly the 64 B of a strip pixel's 32 channels (no padding) for the X strip.  The first version padded both by 16 B (pitches of 68
// and 20 dwords): rows 4 dwords / 20 dwords apart overlap in the banks, two passes per read (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE =
// 0.50, profiles/r03).  The staging writes (8 B per lane, a row's lanes contiguous) are conflict-free at any pitch.
constexpr int GP = 320;                // dG image row pitch (bytes): 128 bf16 + 64
constexpr int XP = 64;                 // X strip pixel pitch (bytes): 32 bf16
constexpr int G_BYTES = 128 * GP;      // 40,960
constexpr int XPIX = 192;              // strip pixels allocated (8 x 20 = 160, or 2 x 8 x 12 = 192)
constexpr int X_BYTES = XPIX * XP;     // 12,288

__device__ __forceinline__ unsigned wpack2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// two fp32 -> packed bf16 (round to nearest even); a, b become the remainders v - bf16(v), exact in fp32 (the pieces of the three-piece form)
__device__ __forceinline__ unsigned wpack2_rest(float& a, float& b) {
    const unsigned p2 = wpack2(a, b);
    a -= __builtin_bit_cast(float, p2 << 16); b -= __builtin_bit_cast(float, p2 & 0xffff0000u);
    return p2;
}
// 4 rows x 16 columns of 16-bit elements, transposed: this lane's column, the 4 rows (see the header)
template <int OFF>
__device__ __forceinline__ bf16x4 lds_read_tr(unsigned addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
}  // namespace

template <int TW>   // tile = 8 x 16 pixels of one image (TW = 16) or 8 x 8 pixels of two images (TW = 8)
__global__ __launch_bounds__(256, 2) void wgrad5x5_bf16_kernel(const WgradDesc d, int tiles_per_split) {
    constexpr int tw = TW;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // dG image | X strip
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = d.Hx, W = d.Wx, N = d.N;
    const int ncb = d.cin >> 5;
    // block -> (ky, channel block, column block); blockIdx.y = pixel slice
    int bx = blockIdx.x;
    const int ky = bx % 5; bx /= 5;
    const int cb = bx % ncb, nb = bx / ncb;
    constexpr int ti_n = tw == 16 ? 1 : 2;
    constexpr int SWC = tw + 4;                               // strip columns
    const int tpr = W / tw, tpi = (H / 8) * tpr;
    const int n_tiles = (d.B / ti_n) * tpi;
    const int t_begin = blockIdx.y * tiles_per_split;
    const int t_end = min(n_tiles, t_begin + tiles_per_split);

    const int ch0 = cb * 32;                                  // first of the block's 32 channels of concat(x0, x1)
    const bool src0 = ch0 < d.c0;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src0 ? d.x0 : d.x1), 0, src0 ? d.bytes0 : d.bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.dy), 0, d.bytesy, 0x00020000);
    const int ldx = src0 ? d.ld0 : d.ld1, cho = src0 ? ch0 : ch0 - d.c0;
    constexpr unsigned OOB = 0xC0000000u;

    // ---- staging roles ----------------------------------------------------------------------------------------------------
    // dG tile: 128 px x 128 n = 4096 float4: thread -> (px = tid / 32 + 8 j, n4 = tid % 32), j < 16
    // X strip: XPIX px x 8 float4: thread -> (px = tid / 8 + 32 j, c4 = tid % 8), j < 6
    f32x4 rg[16], rx[6];
    auto tile_geom = [&](int t, int& b0, int& y0, int& x0) {
        b0 = (t / tpi) * ti_n;
        const int trem = t - (t / tpi) * tpi;
        y0 = (trem / tpr) * 8; x0 = (trem - (trem / tpr) * tpr) * tw;
    };
    auto load_tile = [&](int t) {
        int b0, y0, x0;
        tile_geom(t, b0, y0, x0);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = (tid >> 5) + 8 * j;                 // anchor of the tile
            const int ti = tw == 16 ? 0 : i >> 6, ay = tw == 16 ? i >> 4 : (i >> 3) & 7, ax = tw == 16 ? i & 15 : i & 7;
            const int m = ((b0 + ti) * H + y0 + ay) * W + x0 + ax;
            rg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsy, (unsigned)((m * d.ldy + nb * 128 + (tid & 31) * 4) * 4), 0, 0));
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int p = (tid >> 3) + 32 * j;                // strip pixel: [image][row][column]
            const int ti = p / (8 * SWC), pr = p - ti * (8 * SWC);
            const int py = pr / SWC, px = pr - py * SWC;
            const int iy = y0 + py + ky - 2, ix = x0 + px - 2;
            const bool ok = ti < ti_n && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const unsigned off = ok ? (unsigned)(((((b0 + ti) * H + iy) * W + ix) * ldx + cho + (tid & 7) * 4) * 4) : OOB;
            rx[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0));
        }
    };
    // bias gradient = column sums of dG: the blocks of kernel row 2 / channel block 0 see every dG pixel of their slice exactly once, in
    // fp32, on its way into LDS
    const bool do_bias = d.db != nullptr && ky == 2 && cb == 0;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    auto store_tile = [&]() {
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < 16; ++j) bsum += rg[j];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            uint2 v;
            v.x = wpack2(rg[j][0], rg[j][1]); v.y = wpack2(rg[j][2], rg[j][3]);
            *reinterpret_cast<uint2*>(lds + ((tid >> 5) + 8 * j) * GP + (tid & 31) * 8) = v;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            uint2 v;
            v.x = wpack2(rx[j][0], rx[j][1]); v.y = wpack2(rx[j][2], rx[j][3]);
            *reinterpret_cast<uint2*>(lds + G_BYTES + ((tid >> 3) + 32 * j) * XP + (tid & 7) * 8) = v;
        }
    };

    // ---- fragment addresses ---------------------------------------------------------------------------------------------------
    // transposing read j (0, 1) of k-step s: lane (group g = lane / 16, q = (lane % 16) / 4, p = lane % 4) supplies the address of
    // pixel k = 16 s + 8 (g / 2) + 4 j + q, columns 16 (g % 2) + 4 p .. + 3 of the operand's 32.  The lane-dependent part goes into
    // one base register per operand, (s, j) and the tap into the instruction's offset field.
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned a_base = lds0 + (8 * (g >> 1) + q) * GP + (wave * 32 + 16 * (g & 1) + 4 * p) * 2;     // dG: this wave's 32 columns
    // X strip pixel of k, tap kx: TW 16: row s, column 8 (g / 2) + 4 j + q + kx; TW 8: image s / 4, row 2 (s % 4) + g / 2, column 4 j + q + kx
    const unsigned b_base = lds0 + G_BYTES + (TW == 16 ? 8 * (g >> 1) + q : (g >> 1) * SWC + q) * XP + (16 * (g & 1) + 4 * p) * 2;
    auto a_off = [](int s, int j) constexpr { return (16 * s + 4 * j) * GP; };
    auto b_off = [](int s, int j, int kx) constexpr {
        return (TW == 16 ? s * SWC + 4 * j + kx : ((s >> 2) * 8 + 2 * (s & 3)) * SWC + 4 * j + kx) * XP;
    };

    f32x16 acc[5];
#pragma unroll
    for (int kx = 0; kx < 5; ++kx)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kx][r] = 0.f;

    auto kstep = [&](auto S) {
        constexpr int s = decltype(S)::value;
        bf16x4 a0 = lds_read_tr<a_off(s, 0)>(a_base), a1 = lds_read_tr<a_off(s, 1)>(a_base);
        bf16x4 b00 = lds_read_tr<b_off(s, 0, 0)>(b_base), b10 = lds_read_tr<b_off(s, 1, 0)>(b_base);
        bf16x4 b01 = lds_read_tr<b_off(s, 0, 1)>(b_base), b11 = lds_read_tr<b_off(s, 1, 1)>(b_base);
        bf16x4 b02 = lds_read_tr<b_off(s, 0, 2)>(b_base), b12 = lds_read_tr<b_off(s, 1, 2)>(b_base);
        bf16x4 b03 = lds_read_tr<b_off(s, 0, 3)>(b_base), b13 = lds_read_tr<b_off(s, 1, 3)>(b_base);
        bf16x4 b04 = lds_read_tr<b_off(s, 0, 4)>(b_base), b14 = lds_read_tr<b_off(s, 1, 4)>(b_base);
        // (inline asm: the waits are explicit and tied to the fragments so that no MFMA moves above them)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b00), "+v"(b10), "+v"(b01), "+v"(b11),
                     "+v"(b02), "+v"(b12), "+v"(b03), "+v"(b13), "+v"(b04), "+v"(b14));
        const bf16x8 fa = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_shufflevector(b00, b10, 0, 1, 2, 3, 4, 5, 6, 7), acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_shufflevector(b01, b11, 0, 1, 2, 3, 4, 5, 6, 7), acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_shufflevector(b02, b12, 0, 1, 2, 3, 4, 5, 6, 7), acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_shufflevector(b03, b13, 0, 1, 2, 3, 4, 5, 6, 7), acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_shufflevector(b04, b14, 0, 1, 2, 3, 4, 5, 6, 7), acc[4], 0, 0, 0);
    };
    if (t_begin < t_end) load_tile(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();                                       // every wave is done with the previous tile's images
        store_tile();
        __syncthreads();
        if (t + 1 < t_end) load_tile(t + 1);                   // in flight while this tile is multiplied
        kstep(std::integral_constant<int, 0>{}); kstep(std::integral_constant<int, 1>{});
        kstep(std::integral_constant<int, 2>{}); kstep(std::integral_constant<int, 3>{});
        kstep(std::integral_constant<int, 4>{}); kstep(std::integral_constant<int, 5>{});
        kstep(std::integral_constant<int, 6>{}); kstep(std::integral_constant<int, 7>{});
    }

    // ---- epilogue: accumulator row = n (8 (r / 4) + 4 (lane / 32) + r % 4 of the wave's 32), column = ci (lane % 32) --------------------
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) {
        float* base = d.dw + ((size_t)((ky * 5 + kx) * (d.wcin >> 5) + cb) * N + nb * 128 + wave * 32) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            atomicAdd(base + ((r & 3) + 8 * (r >> 2) + 4 * half) * 32, acc[kx][r]);
        }
    }
    if (do_bias) {   // thread (tid / 32, n4 = tid % 32) holds the sums of columns 4 n4 .. 4 n4 + 3 over its pixels: lanes l and l + 32 pair up
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = bsum[e] + __shfl_xor(bsum[e], 32, 64);
            if (half == 0) atomicAdd(d.db + nb * 128 + l31 * 4 + e, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// All 25 taps per block, timesteps batched (round 3).  The kernel above gives a block ONE kernel row (5 taps) of a 32-channel x 128-column
// slice, so every 128-pixel tile of dG is fetched (fp32, 64 KB) and converted by 5 x cin/32 blocks: 215 MB of L2 traffic per lstm1 launch
// for 13.4 GFLOP, 62 FLOP per byte -- it runs at the rate the operands arrive, not at the matrix cores' (profiles/r03: 75 us per launch,
// 0.07 of the bf16 peak), and a third of each launch is the 20,480 atomics every block ends with.  Here a block of EIGHT waves owns all 25
// taps of a 32-channel x 64-column slice (50 MFMA tiles of 32 n x 32 ci: wave w takes column half w & 1 and taps (w >> 1) + 4 i, 7 or 6
// accumulator tiles = 112 registers), fed from one dG tile [128 px][64 n] and one X PATCH [8 + 4 rows][16 + 4 columns][32 ci] (two images of
// 8 x 8 + halo on the 8-wide maps): 210 FLOP per fetched byte.  The weight gradient sums over pixels AND timesteps, so a launch takes a
// batch of timesteps (WgradDesc::tcount, operands at signed byte strides): more tiles per block, and the partial sums of a block leave it
// once per batch.  Pixel splits meet in dW by fp32 atomic adds in full-rate shape (two contiguous 128-B rows per wave-instruction).
// LDS images are [px][32 columns] with 64-B rows: the 32 lanes of a transposing read's phase cover 4 consecutive rows = the 64 banks.
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int W25_GH = 128 * 64 + 64;     // bytes of one 32-column half of the dG tile (+64: the two halves' staging writes use different banks)
constexpr int W25_G_BYTES = 2 * W25_GH;   // 16,512
constexpr int W25_XPIX = 320;             // patch pixels staged (12 x 20 = 240, or 2 x 12 x 12 = 288; 5 passes of 64)
constexpr int W25_X_BYTES = W25_XPIX * 64;
}  // namespace

// PCS = 2: every operand as TWO FP16 pieces (hi = fp16(v), lo = fp16(v - hi): 22 bits) and three MFMAs per product (lo * hi, hi * lo, hi * hi), the form of
// the fp16x3 precision mode: fp32-grade sums at a sixth of the matrix-pipe time of the fp32 kernel (wgrad5x5_kernel), which matters beyond this kernel's own
// duration because the sweep is bound by the matrix-pipe work of BOTH streams.  dG is staged times a power of two taken from the largest |value| of the
// batch (WgradDesc::dy_absmax), the activations as they are (LayerNorm outputs, h, ReLU outputs: fp16's range); one accumulator per tile -- three
// roundings per 16 products, where the fp32 MFMA rounds eight times.  LDS: two planes of each image, 74 KB.
// PCS = 3: THREE BF16 pieces (hi + mid + lo = v exactly, fp32's exponent range: no scale) and the six products of weight >= 2^-16 (the bf16x6 mode's form):
// three planes of each image (111 KB), the taps of a k-step in two passes so that their fragments fit the registers.
// NW = 4 (round 5; PCS = 1 only): FOUR waves own all 25 taps of a 32-channel x 32-column slice (wave w: taps w + 4 i), one block per CU.  One wave per
// SIMD (122 + 112 registers) leaves half of every SIMD's registers to whatever the main stream runs: the sweep's small kernels (gate backward, LayerNorm
// sums, 3x3 convs) CO-RESIDE with this side-stream kernel instead of time-slicing whole CUs with the 8-wave form's blocks (profiles/r05/NOTES.md 3: the
// bf16 step is bound by CU time, not by the matrix pipe).  Config 3 train step 11.45 -> 11.29-11.33 ms (A/B/A/B in one call).  The bf16 gate convs' 8-wave
// blocks (2 x 149-221 registers per SIMD) still do not fit beside it; capping this kernel at 208 registers spills 41-56 of them.
template <int TW, int PCS = 1, int NW = 8>
__global__ __launch_bounds__(64 * NW, 1) void wgrad25_bf16_kernel(const WgradDesc d, int tiles_per_split) {
    static_assert(NW == 8 || (NW == 4 && PCS == 1), "four waves: the plain bf16 form");
    constexpr int tw = TW;
    constexpr int NCOL = NW == 8 ? 64 : 32;                   // gate columns per block
    constexpr int GB = NW == 8 ? W25_G_BYTES : W25_GH, XB = W25_X_BYTES;         // bytes of one plane of the dG tile / of the X patch
    constexpr int ti_n = tw == 16 ? 1 : 2;
    constexpr int PWC = tw + 4;                               // patch columns
    constexpr int NPIX = ti_n * 12 * PWC;                     // 240 / 288
    constexpr int XPP = 8 * NW;                               // patch pixels per staging pass (8 float4 each)
    constexpr int NXJ = (NPIX + XPP - 1) / XPP;               // staging passes of the patch
    constexpr int GL = NCOL / 4;                              // float4 lanes per dG row (16 / 8): 32 pixels per pass either way
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // dG halves | X patch
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = d.Hx, W = d.Wx, N = d.N;
    const int ncb = d.cin >> 5;
    const int cb = blockIdx.x % ncb, nb = blockIdx.x / ncb;   // 32 input channels x NCOL gate columns
    const int tpr = W / tw, tpi = (H / 8) * tpr;
    const int n_tiles = (d.B / ti_n) * tpi;                   // per timestep
    const int tcount = d.tcount > 1 ? d.tcount : 1;
    const int g_end_all = n_tiles * tcount;
    const int g_begin = blockIdx.y * tiles_per_split;
    const int g_end = min(g_end_all, g_begin + tiles_per_split);

    const int ch0 = cb * 32;
    const bool src0 = ch0 < d.c0;
    const char* xbase = reinterpret_cast<const char*>(src0 ? d.x0 : d.x1);
    const long long xts = src0 ? d.ts_x0 : d.ts_x1;
    const int xbytes = src0 ? d.bytes0 : d.bytes1;
    const int ldx = src0 ? d.ld0 : d.ld1, cho = src0 ? ch0 : ch0 - d.c0;
    constexpr unsigned OOB = 0xC0000000u;

    // ---- staging: dG tile 128 px x 64 n = 2048 float4: thread -> (px = tid / 16 + 32 j, n4 = tid % 16), j < 4;
    //               X patch NPIX px x 8 float4: thread -> (px = tid / 8 + 64 j, c4 = tid % 8), j < NXJ
    f32x4 rg[4], rx[NXJ];
    auto load_tile = [&](int gt) {
        const int tj = __builtin_amdgcn_readfirstlane(tcount > 1 ? gt / n_tiles : 0);      // timestep of the batch (block-uniform)
        const int t = gt - tj * n_tiles;
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xbase + (long long)tj * xts), 0, xbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(d.dy) + (long long)tj * d.ts_dy), 0, d.bytesy, 0x00020000);
        const int b0 = (t / tpi) * ti_n, trem = t - (t / tpi) * tpi;
        const int y0 = (trem / tpr) * 8, x0 = (trem - (trem / tpr) * tpr) * tw;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid / GL + 32 * j;                  // anchor of the tile
            const int ti = tw == 16 ? 0 : i >> 6, ay = tw == 16 ? i >> 4 : (i >> 3) & 7, ax = tw == 16 ? i & 15 : i & 7;
            const int m = ((b0 + ti) * H + y0 + ay) * W + x0 + ax;
            rg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsy, (unsigned)((m * d.ldy + nb * NCOL + (tid % GL) * 4) * 4), 0, 0));
        }
#pragma unroll
        for (int j = 0; j < NXJ; ++j) {
            const int p = (tid >> 3) + XPP * j;               // patch pixel: [image][row][column]
            const int ti = p / (12 * PWC), pr = p - ti * (12 * PWC);
            const int py = pr / PWC, px = pr - py * PWC;
            const int iy = y0 + py - 2, ix = x0 + px - 2;
            const bool ok = p < NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const unsigned off = ok ? (unsigned)(((((b0 + ti) * H + iy) * W + ix) * ldx + cho + (tid & 7) * 4) * 4) : OOB;
            rx[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0));
        }
    };
    // bias gradient = column sums of dG (fp32, on its way into LDS): the blocks of channel block 0 see every dG element of their columns once
    const bool do_bias = d.db != nullptr && cb == 0;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    float gscale = 1.0f;                                      // PCS = 2: dG's power-of-two scale (the largest |value| of the batch's timesteps)
    if constexpr (PCS == 2) {
        float m = 0.f;
        for (int j = 0; j < tcount; ++j) m = __builtin_fmaxf(m, d.dy_absmax[(size_t)j * d.dy_absmax_stride + 2 + lane]);
        m = wave_max(m);
        gscale = pivp_x3_scale_of_max(m);
    }
    auto store_tile = [&]() {
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bsum += rg[j];
        }
        const int n4 = tid % GL;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned char* dst = lds + (n4 >> 3) * W25_GH + (tid / GL + 32 * j) * 64 + (n4 & 7) * 8;
            if constexpr (PCS == 2) {
                float r0 = rg[j][0] * gscale, r1 = rg[j][1] * gscale, r2 = rg[j][2] * gscale, r3 = rg[j][3] * gscale;
                uint2 h, l;
                h.x = pivp_pack2h_rest(r0, r1); h.y = pivp_pack2h_rest(r2, r3);
                l.x = pivp_pack2h_rest(r0, r1); l.y = pivp_pack2h_rest(r2, r3);
                *reinterpret_cast<uint2*>(dst) = h;
                *reinterpret_cast<uint2*>(dst + GB) = l;
            } else if constexpr (PCS == 3) {
                float r0 = rg[j][0], r1 = rg[j][1], r2 = rg[j][2], r3 = rg[j][3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    uint2 v;
                    v.x = wpack2_rest(r0, r1); v.y = wpack2_rest(r2, r3);
                    *reinterpret_cast<uint2*>(dst + pl * GB) = v;
                }
            } else {
                uint2 v;
                v.x = wpack2(rg[j][0], rg[j][1]); v.y = wpack2(rg[j][2], rg[j][3]);
                *reinterpret_cast<uint2*>(dst) = v;
            }
        }
#pragma unroll
        for (int j = 0; j < NXJ; ++j) {
            unsigned char* dst = lds + PCS * GB + ((tid >> 3) + XPP * j) * 64 + (tid & 7) * 8;
            if constexpr (PCS == 2) {
                float r0 = rx[j][0], r1 = rx[j][1], r2 = rx[j][2], r3 = rx[j][3];
                uint2 h, l;
                h.x = pivp_pack2h_rest(r0, r1); h.y = pivp_pack2h_rest(r2, r3);
                l.x = pivp_pack2h_rest(r0, r1); l.y = pivp_pack2h_rest(r2, r3);
                *reinterpret_cast<uint2*>(dst) = h;
                *reinterpret_cast<uint2*>(dst + XB) = l;
            } else if constexpr (PCS == 3) {
                float r0 = rx[j][0], r1 = rx[j][1], r2 = rx[j][2], r3 = rx[j][3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    uint2 v;
                    v.x = wpack2_rest(r0, r1); v.y = wpack2_rest(r2, r3);
                    *reinterpret_cast<uint2*>(dst + pl * XB) = v;
                }
            } else {
                uint2 v;                                      // (pixels past the patch carry the zeros of their out-of-range loads)
                v.x = wpack2(rx[j][0], rx[j][1]); v.y = wpack2(rx[j][2], rx[j][3]);
                *reinterpret_cast<uint2*>(dst) = v;
            }
        }
    };

    // ---- fragments (transposing reads, see the kernel above): lane (g = lane / 16, q = (lane % 16) / 4, p = lane % 4) addresses pixel
    // k = 16 s + 8 (g / 2) + 4 j + q of k-step s, columns 16 (g % 2) + 4 p .. + 3 of the operand's 32
    const int g = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int nt = NW == 8 ? wave & 1 : 0, tg = NW == 8 ? wave >> 1 : wave;      // this wave's 32 columns; its taps are tg, tg + 4, ... (< 25)
    const unsigned a_base = lds0 + nt * W25_GH + (8 * (g >> 1) + q) * 64 + (16 * (g & 1) + 4 * p4) * 2;
    const unsigned b_lane = lds0 + PCS * GB + (TW == 16 ? 8 * (g >> 1) + q : (g >> 1) * PWC + q) * 64 + (16 * (g & 1) + 4 * p4) * 2;
    unsigned b_base[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int tap = min(tg + 4 * i, 24), ky = tap / 5, kx = tap - ky * 5;
        b_base[i] = b_lane + (ky * PWC + kx) * 64;
    }
    const bool seven = tg == 0;                               // taps 0, 4, ..., 24; the other three groups have six
    auto a_off = [](int s, int j) constexpr { return (16 * s + 4 * j) * 64; };
    auto b_off = [](int s, int j) constexpr {                 // patch pixel of k-step s, read j (relative to the tap's origin)
        return (TW == 16 ? s * PWC + 4 * j : ((s >> 2) * 12 + 2 * (s & 3)) * PWC + 4 * j) * 64;
    };

    f32x16 acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    auto kstep = [&](auto S) {
        constexpr int s = decltype(S)::value;
        if constexpr (PCS == 3) {
            auto j8 = [](const bf16x4& u, const bf16x4& v) { return __builtin_shufflevector(u, v, 0, 1, 2, 3, 4, 5, 6, 7); };
            bf16x4 ah0 = lds_read_tr<a_off(s, 0)>(a_base), ah1 = lds_read_tr<a_off(s, 1)>(a_base);
            bf16x4 am0 = lds_read_tr<a_off(s, 0) + GB>(a_base), am1 = lds_read_tr<a_off(s, 1) + GB>(a_base);
            bf16x4 al0 = lds_read_tr<a_off(s, 0) + 2 * GB>(a_base), al1 = lds_read_tr<a_off(s, 1) + 2 * GB>(a_base);
            bf16x8 fh, fm, fl;
            auto pass = [&](auto I0, auto N) {       // taps i0 .. i0 + n - 1 of this wave's seven
                constexpr int i0 = decltype(I0)::value, n = decltype(N)::value;
                bf16x4 h0[n], h1[n], m0[n], m1[n], l0[n], l1[n];
#pragma unroll
                for (int i = 0; i < n; ++i) {
                    h0[i] = lds_read_tr<b_off(s, 0)>(b_base[i0 + i]); h1[i] = lds_read_tr<b_off(s, 1)>(b_base[i0 + i]);
                    m0[i] = lds_read_tr<b_off(s, 0) + XB>(b_base[i0 + i]); m1[i] = lds_read_tr<b_off(s, 1) + XB>(b_base[i0 + i]);
                    l0[i] = lds_read_tr<b_off(s, 0) + 2 * XB>(b_base[i0 + i]); l1[i] = lds_read_tr<b_off(s, 1) + 2 * XB>(b_base[i0 + i]);
                }
                if constexpr (i0 == 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah0), "+v"(ah1), "+v"(am0), "+v"(am1), "+v"(al0), "+v"(al1));
                    fh = j8(ah0, ah1); fm = j8(am0, am1); fl = j8(al0, al1);
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
#pragma unroll
                for (int i = 0; i < n; ++i) {
                    asm volatile("" : "+v"(h0[i]), "+v"(h1[i]), "+v"(m0[i]), "+v"(m1[i]), "+v"(l0[i]), "+v"(l1[i]));      // (behind the wait above)
                    if (i0 + i < 6 || seven) {
                        const bf16x8 bh = j8(h0[i], h1[i]), bm = j8(m0[i], m1[i]), bl = j8(l0[i], l1[i]);
                        f32x16 c = acc[i0 + i];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, bh, c, 0, 0, 0);       // lo * hi
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, bl, c, 0, 0, 0);       // hi * lo
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fm, bm, c, 0, 0, 0);       // mid * mid
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fm, bh, c, 0, 0, 0);       // mid * hi
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, bm, c, 0, 0, 0);       // hi * mid
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, bh, c, 0, 0, 0);       // hi * hi
                        acc[i0 + i] = c;
                    }
                }
            };
            pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
            pass(std::integral_constant<int, 4>{}, std::integral_constant<int, 3>{});
            return;
        }
        bf16x4 a0 = lds_read_tr<a_off(s, 0)>(a_base), a1 = lds_read_tr<a_off(s, 1)>(a_base);
        bf16x4 b0[7], b1[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) { b0[i] = lds_read_tr<b_off(s, 0)>(b_base[i]); b1[i] = lds_read_tr<b_off(s, 1)>(b_base[i]); }
        if constexpr (PCS == 2) {
            // the second pieces: the same addresses one plane further (in the instruction's offset field)
            bf16x4 a0l = lds_read_tr<a_off(s, 0) + GB>(a_base), a1l = lds_read_tr<a_off(s, 1) + GB>(a_base);
            bf16x4 c0[7], c1[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) { c0[i] = lds_read_tr<b_off(s, 0) + XB>(b_base[i]); c1[i] = lds_read_tr<b_off(s, 1) + XB>(b_base[i]); }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0[0]), "+v"(b1[0]), "+v"(b0[1]), "+v"(b1[1]), "+v"(b0[2]), "+v"(b1[2]),
                         "+v"(b0[3]), "+v"(b1[3]), "+v"(b0[4]), "+v"(b1[4]), "+v"(b0[5]), "+v"(b1[5]), "+v"(b0[6]), "+v"(b1[6]));
            asm volatile("" : "+v"(a0l), "+v"(a1l), "+v"(c0[0]), "+v"(c1[0]), "+v"(c0[1]), "+v"(c1[1]), "+v"(c0[2]), "+v"(c1[2]),
                         "+v"(c0[3]), "+v"(c1[3]), "+v"(c0[4]), "+v"(c1[4]), "+v"(c0[5]), "+v"(c1[5]), "+v"(c0[6]), "+v"(c1[6]));
            auto h8 = [](const bf16x4& u, const bf16x4& v) { return __builtin_bit_cast(pivp_f16x8, __builtin_shufflevector(u, v, 0, 1, 2, 3, 4, 5, 6, 7)); };
            const pivp_f16x8 fa = h8(a0, a1), fal = h8(a0l, a1l);
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                if (i < 6 || seven) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal, h8(b0[i], b1[i]), acc[i], 0, 0, 0);       // lo * hi
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, h8(c0[i], c1[i]), acc[i], 0, 0, 0);        // hi * lo
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, h8(b0[i], b1[i]), acc[i], 0, 0, 0);        // hi * hi
                }
            }
            return;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0[0]), "+v"(b1[0]), "+v"(b0[1]), "+v"(b1[1]), "+v"(b0[2]), "+v"(b1[2]),
                     "+v"(b0[3]), "+v"(b1[3]), "+v"(b0[4]), "+v"(b1[4]), "+v"(b0[5]), "+v"(b1[5]), "+v"(b0[6]), "+v"(b1[6]));
        const bf16x8 fa = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_shufflevector(b0[i], b1[i], 0, 1, 2, 3, 4, 5, 6, 7), acc[i], 0, 0, 0);
        if (seven) acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_shufflevector(b0[6], b1[6], 0, 1, 2, 3, 4, 5, 6, 7), acc[6], 0, 0, 0);
    };
    if (g_begin < g_end) load_tile(g_begin);
    for (int gt = g_begin; gt < g_end; ++gt) {
        __syncthreads();                                       // every wave is done with the previous tile's images
        store_tile();
        __syncthreads();
        if (gt + 1 < g_end) load_tile(gt + 1);                 // in flight while this tile is multiplied
        kstep(std::integral_constant<int, 0>{}); kstep(std::integral_constant<int, 1>{});
        kstep(std::integral_constant<int, 2>{}); kstep(std::integral_constant<int, 3>{});
        kstep(std::integral_constant<int, 4>{}); kstep(std::integral_constant<int, 5>{});
        kstep(std::integral_constant<int, 6>{}); kstep(std::integral_constant<int, 7>{});
    }

    // ---- epilogue: accumulator row = n (8 (r / 4) + 4 (lane / 32) + r % 4 of the wave's 32), column = ci (lane % 32): a wave-instruction
    // adds two contiguous 128-B rows of the K-inner packed gradient [tap][ci / 32][n][32] -----------------------------------------------
    const int half = lane >> 5, l31 = lane & 31;
    if (g_begin < g_end) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int tap = tg + 4 * i;
            if (tap < 25) {
                float* base = d.dw + ((size_t)(tap * (d.wcin >> 5) + cb) * N + nb * NCOL + nt * 32) * 32 + l31;
                const float inv = 1.0f / gscale;                 // (1 without pieces; a power of two with them: exact)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    atomicAdd(base + ((r & 3) + 8 * (r >> 2) + 4 * half) * 32, acc[i][r] * inv);      // (a build without them: 0.5 / 0.2 ms of a train step, profiles/r04)
                }
            }
        }
    }
    if (do_bias) {   // thread (tid / GL, n4 = tid % GL) holds the sums of columns 4 n4 .. 4 n4 + 3 over its pixels: the lanes l ^ GL, l ^ 2 GL, ... pair up
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = bsum[e];
#pragma unroll
            for (int o = 32; o >= GL; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane < GL) atomicAdd(d.db + nb * NCOL + lane * 4 + e, v);
        }
    }
}

bool wgrad5x5_bf16_ok(const WgradDesc& d) {
    if (d.deconv || d.ksize != 5 || d.pad != 2 || d.stride != 1 || d.Hx != d.Hy || d.Wx != d.Wy) return false;
    if (d.N % 128 || d.cin % 32 || d.c0 % 32 || d.c1 % 32 || d.ld0 % 4 || d.ld1 % 4 || d.ldy % 4 || d.Hx % 8) return false;
    if (d.tcount > 1 && (d.ts_x0 % 16 || d.ts_x1 % 16 || d.ts_dy % 16)) return false;
    if (d.Wx % 16 == 0) return true;
    return d.Wx % 8 == 0 && d.B % 2 == 0;
}

// the 25-tap kernel: grid = (cin / 32) x (N / 64) output slices x pixel splits over the tiles of ALL timesteps of the batch
// four-wave blocks (the bf16 mode's batched launches): about one per CU, beside the main stream's kernels
static int launch_wgrad25_nw4(const WgradDesc& d, hipStream_t s) {
    constexpr int lds_bytes = W25_GH + W25_X_BYTES;
    const int tw = d.Wx % 16 == 0 ? 16 : 8, ti_n = tw == 16 ? 1 : 2;
    const int n_tiles = (d.B / ti_n) * (d.Hx / 8) * (d.Wx / tw) * (d.tcount > 1 ? d.tcount : 1);
    const int gx = (d.cin / 32) * (d.N / 32);
    int ns = (pivp_cu_count() + gx - 1) / gx;      // about one block per CU (config 3's step against the block target: 128: 11.84 ms, 256: 11.28, 384: 11.25, 512: 11.60)
    if (ns > n_tiles / 2) ns = n_tiles / 2;
    if (ns < 1) ns = 1;
    const int tps = (n_tiles + ns - 1) / ns;
    ns = (n_tiles + tps - 1) / tps;
    if (tw == 16) hipLaunchKernelGGL((wgrad25_bf16_kernel<16, 1, 4>), dim3(gx, ns), dim3(256), lds_bytes, s, d, tps);
    else hipLaunchKernelGGL((wgrad25_bf16_kernel<8, 1, 4>), dim3(gx, ns), dim3(256), lds_bytes, s, d, tps);
    return PIVP_LAUNCH_STATUS();
}
template <int PCS>
static int launch_wgrad25(const WgradDesc& d, hipStream_t s) {
    constexpr int lds_bytes = PCS * (W25_G_BYTES + W25_X_BYTES);
    static PerDeviceOnce once16, once8;
    if (pivp_ensure_dyn_lds(once16, reinterpret_cast<const void*>(&wgrad25_bf16_kernel<16, PCS>), lds_bytes) != PIVP_OK ||
        pivp_ensure_dyn_lds(once8, reinterpret_cast<const void*>(&wgrad25_bf16_kernel<8, PCS>), lds_bytes) != PIVP_OK)
        return PIVP_ERR_LAUNCH;
    const int tw = d.Wx % 16 == 0 ? 16 : 8, ti_n = tw == 16 ? 1 : 2;
    const int n_tiles = (d.B / ti_n) * (d.Hx / 8) * (d.Wx / tw) * (d.tcount > 1 ? d.tcount : 1);
    const int gx = (d.cin / 32) * (d.N / 64);
    // Pixel splits: one block per CU (8 waves, ~70 KB of LDS), at least 2 tiles per block; every split ends with 25 x 32 x 64 atomic adds
    // (205 KB: the batch of timesteps is what amortises them).
    // (fp16 pieces: its 8-wave blocks hold a CU's whole register file, and the sweep's small kernels need CUs without one: half the CUs)
    // bf16: three quarters (train step 11.86 -> 11.66 ms, profiles/r04/bf16_train_wgrad_batch_slots.txt)
    const int target = PCS >= 2 ? pivp_cu_count() / 2 : pivp_cu_count() * 3 / 4;
    int ns = (target + gx - 1) / gx;
    if (ns > n_tiles / 2) ns = n_tiles / 2;
    if (ns < 1) ns = 1;
    const int tps = (n_tiles + ns - 1) / ns;
    ns = (n_tiles + tps - 1) / tps;
    if (tw == 16) hipLaunchKernelGGL((wgrad25_bf16_kernel<16, PCS>), dim3(gx, ns), dim3(512), lds_bytes, s, d, tps);
    else hipLaunchKernelGGL((wgrad25_bf16_kernel<8, PCS>), dim3(gx, ns), dim3(512), lds_bytes, s, d, tps);
    return PIVP_LAUNCH_STATUS();
}

// d as for igemm_wgrad (ConvLSTM case: 5x5, stride 1, pad 2); dW and, when d.db is set, the bias gradient accumulated with atomics.
int wgrad5x5_bf16(const WgradDesc& d, hipStream_t s) {
    PIVP_CHECK_ARG(d.x0 && d.dy && d.dw && wgrad5x5_bf16_ok(d) && (d.c1 == 0 || d.x1) && d.wcin >= d.cin && d.wcin % 32 == 0);
    // A batch of timesteps goes to the 25-tap kernel (per timestep at B = 32, all seven layers: 168 us at 8 per launch, 225 us at 4); ONE
    // timestep to the kernel-row kernel below, whose 5 x cin/32 blocks per tile are then the better use of the chip (389 us against 580:
    // the sweep's t = 0 launches, sequences too short to batch).
    if (d.dy_absmax) {       // two fp16 pieces per operand: the 25-tap kernel, whatever the batch
        PIVP_CHECK_ARG(d.dy_absmax_stride >= 66 || d.tcount <= 1);
        return launch_wgrad25<2>(d, s);
    }
    if (d.pieces == 3) return launch_wgrad25<3>(d, s);      // three bf16 pieces per operand, likewise
    // A batch of timesteps (WgradDesc::form): the four-wave form (co-resident with the main stream's small kernels) on maps of up to 32 x 32 x 32 pixels per
    // timestep, the eight-wave form (half the patch traffic per multiply-add) on larger ones; the plan asks for the eight-wave form on every layer of frames
    // above 64 x 64 x 32 -- config 5, whose main-stream kernels fill the chip by themselves: 68.2 ms with the four-wave form everywhere against 66.3.
    // (One timestep on the four-wave form: config 3 11.36 against 11.29 ms: the kernel-row kernel below stays.)
    if (d.tcount > 1) return (d.form == 1 || (d.form == 0 && (long)d.B * d.Hx * d.Wx <= 32L * 32 * 32)) ? launch_wgrad25_nw4(d, s) : launch_wgrad25<1>(d, s);
    constexpr int lds_bytes = G_BYTES + X_BYTES;
    static PerDeviceOnce once16, once8;
    if (pivp_ensure_dyn_lds(once16, reinterpret_cast<const void*>(&wgrad5x5_bf16_kernel<16>), lds_bytes) != PIVP_OK ||
        pivp_ensure_dyn_lds(once8, reinterpret_cast<const void*>(&wgrad5x5_bf16_kernel<8>), lds_bytes) != PIVP_OK)
        return PIVP_ERR_LAUNCH;
    const int tw = d.Wx % 16 == 0 ? 16 : 8, ti_n = tw == 16 ? 1 : 2;
    const int n_tiles = (d.B / ti_n) * (d.Hx / 8) * (d.Wx / tw);
    const int gx = 5 * (d.cin / 32) * (d.N / 128);
    // Pixel splits: about one block per TWO CUs, at least 2 tiles per block.  Two blocks per CU could be resident, but the kernel is not
    // short of parallelism: it is short of tiles per block -- every block ends with 20,480 atomic adds into the same 80 KB of dW as the
    // other splits of its tile (timing-only build without them: 72 -> 43 us on lstm1 at 52 splits) -- and it runs on the side stream, where
    // a grid that fills every CU's registers starves the main stream's small kernels (a 5 us add_strided took 34 us beside it).
    // Train step in the bf16 mode against the block target: 512: 13.93 ms, 256: 13.00, 192: 12.73, 128: 12.54, 96: 13.08, 64: 14.57.
    const int target = pivp_cu_count() / 2;
    int ns = (target + gx - 1) / gx;
    if (ns > n_tiles / 2) ns = n_tiles / 2;
    if (ns < 1) ns = 1;
    const int tps = (n_tiles + ns - 1) / ns;
    ns = (n_tiles + tps - 1) / tps;
    if (tw == 16) hipLaunchKernelGGL(wgrad5x5_bf16_kernel<16>, dim3(gx, ns), dim3(256), lds_bytes, s, d, tps);
    else hipLaunchKernelGGL(wgrad5x5_bf16_kernel<8>, dim3(gx, ns), dim3(256), lds_bytes, s, d, tps);
    return PIVP_LAUNCH_STATUS();
}

}  // namespace pivp
