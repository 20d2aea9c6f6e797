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
#include "pivp_kernels.h"

namespace pivp {

constexpr int WG_PIX = 32;     // pixels (GEMM K) per chunk
constexpr int WG_CI = 64;      // input channels per block
constexpr int WG_N = 128;      // output columns per block
constexpr int WG_XP = WG_CI + 4;   // LDS pitches (floats); +4 keeps 16-B alignment of rows
constexpr int WG_YP = WG_N + 4;

__global__ __launch_bounds__(256, 1) void igemm_wgrad_kernel(const WgradDesc d) {
    __shared__ __attribute__((aligned(16))) float xs[2][WG_PIX * WG_XP];
    __shared__ __attribute__((aligned(16))) float ys[2][WG_PIX * WG_YP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;             // wave tile: 32 ci x 64 n
    const int l31 = lane & 31, half = lane >> 5;
    const int ncb = (d.cin + WG_CI - 1) / WG_CI, nnb = (d.N + WG_N - 1) / WG_N;
    int bid = blockIdx.x;
    const int nb = bid % nnb; bid /= nnb;
    const int cb = bid % ncb; bid /= ncb;
    const int tap = bid;                                  // 0 .. ksize*ksize-1
    const int ky = tap / d.ksize, kx = tap - ky * d.ksize;
    const int ci0 = cb * WG_CI, n0 = nb * WG_N;
    const int nsplit = gridDim.y, split = blockIdx.y;
    const int nchunks_total = (d.M + WG_PIX - 1) / WG_PIX;
    const int c_begin = (int)((long)nchunks_total * split / nsplit), c_end = (int)((long)nchunks_total * (split + 1) / nsplit);
    const int HWg = d.Hg * d.Wg;
    // offsets of the two operands relative to the anchor
    const int ady = d.deconv ? 0 : ky - d.pad, adx = d.deconv ? 0 : kx - d.pad;          // X side (after anchor*sa)
    const int bdy = d.deconv ? ky - 1 : 0, bdx = d.deconv ? kx - 1 : 0;                   // dY side (after anchor*sb)
    const int sa = d.deconv ? 1 : d.stride, sb = d.deconv ? 2 : 1;
    constexpr unsigned OOB = 0xC0000000u;
    const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x0), 0, d.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.c1 ? d.x1 : d.x0), 0, d.c1 ? d.bytes1 : d.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.dy), 0, d.bytesy, 0x00020000);

    // staging: X tile 32 pix x 64 ci = 512 float4 (2 per thread), dY tile 32 pix x 128 n = 1024 float4 (4 per thread)
    const int xr = tid >> 4, xc = (tid & 15) * 4;        // rows xr, xr+16
    const int yr = tid >> 5, yc = (tid & 31) * 4;        // rows yr, yr+8, yr+16, yr+24
    // this thread's 4 X columns live in x0 (channels [0,c0)) or x1 ([c0,c0+c1)); a 64-channel block may straddle
    // the two, so both descriptors are read and the one that does not hold the columns gets an out-of-range offset
    // (hardware returns 0) -- the sum is the value, no per-lane descriptor select.
    const int xci = ci0 + xc;
    const bool xcol_ok = xci < d.cin;
    const bool in0 = xci < d.c0;
    const bool ycol_ok = (n0 + yc) < d.N;
    f32x4 rx[2], rY[4];
    auto issue = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = chunk * WG_PIX + xr + 16 * j;
            unsigned off0 = OOB, off1 = OOB;
            if (m < d.M && xcol_ok) {
                const int b = m / HWg, rem = m - b * HWg, ay = rem / d.Wg, ax = rem - ay * d.Wg;
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
        for (int j = 0; j < 4; ++j) {
            const int m = chunk * WG_PIX + yr + 8 * j;
            unsigned off = OOB;
            if (m < d.M && ycol_ok) {
                const int b = m / HWg, rem = m - b * HWg, ay = rem / d.Wg, ax = rem - ay * d.Wg;
                const int oy = ay * sb + bdy, ox = ax * sb + bdx;
                if ((unsigned)oy < (unsigned)d.Hy && (unsigned)ox < (unsigned)d.Wy)
                    off = (unsigned)((((b * d.Hy + oy) * d.Wy + ox) * d.ldy + n0 + yc) * 4);
            }
            rY[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, off, 0, 0));
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(&xs[buf][(xr + 16 * j) * WG_XP + xc]) = rx[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(&ys[buf][(yr + 8 * j) * WG_YP + yc]) = rY[j];
    };
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (c_begin < c_end) {
        issue(c_begin);
        store(0);
        __syncthreads();
        for (int c = c_begin; c < c_end; ++c) {
            const int buf = (c - c_begin) & 1;
            if (c + 1 < c_end) issue(c + 1);
            const float* X = &xs[buf][wm * 32 + l31];
            const float* Y = &ys[buf][wn * 64 + l31];
#pragma unroll
            for (int k = 0; k < WG_PIX; k += 2) {
                const float a = X[(k + half) * WG_XP];
                const float b0 = Y[(k + half) * WG_YP], b1 = Y[(k + half) * WG_YP + 32];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
            }
            if (c + 1 < c_end) store(buf ^ 1);
            __syncthreads();
        }
    }
    // acc[t][r]: row i = ci (= (r&3) + 8*(r>>2) + 4*half), column j = n (= l31); packed gradient [tap][ci/32][n][ci%32]
    const int wtap = tap;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = n0 + wn * 64 + t * 32 + l31;
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

int igemm_wgrad(const WgradDesc& d, hipStream_t s) {
    PIVP_CHECK_ARG(d.x0 && d.dy && d.dw && d.c0 > 0 && d.c0 % 32 == 0 && d.c1 >= 0 && d.c1 % 32 == 0 && (d.c1 == 0 || d.x1));
    PIVP_CHECK_ARG(d.cin == d.c0 + d.c1 && d.wcin >= d.cin && d.wcin % 32 == 0 && d.N > 0 && d.N % 32 == 0);
    PIVP_CHECK_ARG(d.M == d.B * d.Hg * d.Wg && d.M > 0 && d.ksize >= 1 && d.ksize <= 7);
    PIVP_CHECK_ARG(d.bytes0 > 0 && d.bytesy > 0 && (d.c1 == 0 || d.bytes1 > 0));
    const int ncb = (d.cin + WG_CI - 1) / WG_CI, nnb = (d.N + WG_N - 1) / WG_N;
    const int tiles = d.ksize * d.ksize * ncb * nnb;
    const int chunks = (d.M + WG_PIX - 1) / WG_PIX;
    int nsplit = (1024 + tiles - 1) / tiles;              // aim at ~4 blocks per CU
    if (nsplit > chunks / 8) nsplit = chunks / 8;         // but keep >= 8 chunks (256 pixels) per block
    if (nsplit < 1) nsplit = 1;
    hipLaunchKernelGGL(igemm_wgrad_kernel, dim3(tiles, nsplit), dim3(256), 0, s, d);
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
