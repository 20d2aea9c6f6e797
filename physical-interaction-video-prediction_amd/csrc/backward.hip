// Backward (BPTT) kernels of the light ops and the optimizer step.  The reference gets all of this from Chainer's
// autograd inside optimizer.update (train_model.py:950); each kernel cites the forward code it differentiates
// ("TM" = src/models/train_model.py).  Heavy contractions (data / weight gradients of the convolutions) are in
// igemm_f32.hip / igemm_wgrad.hip.  All gradient buffers ACCUMULATE (Chainer: cleargrads() then backward()).
#include <string.h>

#include "pivp_kernels.h"

namespace pivp {

__device__ __forceinline__ float fast_tanh_b(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

// ------------------------------------------------------------------------------------------
// ConvLSTM gate math backward (TM:269-272):  c' = c*f + i*j,  h = tanh(c')*o  with the stored activations
// j = tanh(gj), i = s(gi), f = s(gf+1), o = s(go).  dh arrives in two pieces: from this step's LayerNorm backward
// (dh_a) and from the NEXT timestep's gate-conv data gradient, whose last C channels are d h_t (dh_b, may be null).
// Writes the pre-activation gate gradients dG [M][4C] (column order j,i,f,o like the weights) and turns dc into
// d c_{t-1} in place.
// ------------------------------------------------------------------------------------------
// Thread = 4 consecutive channels of one pixel, block = 1024 consecutive elements (NHWC) of one sample.  With `ln.dy` set, dh_a is not
// read but formed on the fly as the LayerNorm backward of the norm behind this cell (TM:203-208; what ln_bwd_apply_kernel would
// have written): dh_a = rstd * (dy*gamma - m1 - xhat*m2), xhat = (h - mean)*rstd, (m1, m2) = the sample's means of g and g*xhat
// from ln_bwd_stats_kernel's partials -- one launch and one 4-byte-per-element round trip less per ConvLSTM and timestep.
__global__ __launch_bounds__(256) void lstm_gates_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_old,
                                                             const float* __restrict__ c_new, const float* __restrict__ dh_a, int lda,
                                                             const float* __restrict__ dh_b, int ldb, float* __restrict__ dc,
                                                             int dc_valid, float* __restrict__ dG, int npix, int C, const LnFuse ln,
                                                             float* __restrict__ zero, long long zero_f4) {
    PIVP_SET_MAIN_PRIO();
    __shared__ float sums[2];
    const int b = blockIdx.y;
    const int n = npix * C;                                   // elements per sample
    if (zero) {   // destination of the K-split data gradient that follows (atomic adds): cleared here instead of by a memset launch of its own
        const long long nthr = (long long)gridDim.x * gridDim.y * 256;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (long long i = ((long long)b * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < zero_f4; i += nthr) reinterpret_cast<f32x4*>(zero)[i] = z;
    }
    if (ln.dy) {
        if (threadIdx.x < 64) {
            float a = 0.f, c2 = 0.f;
            for (int i = threadIdx.x; i < ln.S; i += 64) { a += ln.partials[((size_t)b * ln.S + i) * 2]; c2 += ln.partials[((size_t)b * ln.S + i) * 2 + 1]; }
            a = wave_sum(a); c2 = wave_sum(c2);
            if (threadIdx.x == 0) { sums[0] = a / (float)n; sums[1] = c2 / (float)n; }
        }
        __syncthreads();
    }
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;       // element of the sample
    if (e >= n) return;
    const int pix = e / C, ch = e - pix * C;
    const size_t m = (size_t)b * npix + pix;                  // global pixel
    const size_t idx = m * C + ch;
    const float* g = gates + m * 4 * C + ch;
    const f32x4 aj = *reinterpret_cast<const f32x4*>(g), ai = *reinterpret_cast<const f32x4*>(g + C);
    const f32x4 af = *reinterpret_cast<const f32x4*>(g + 2 * C), ao = *reinterpret_cast<const f32x4*>(g + 3 * C);
    const f32x4 cn = *reinterpret_cast<const f32x4*>(c_new + idx), co = *reinterpret_cast<const f32x4*>(c_old + idx);
    f32x4 dh = {0.f, 0.f, 0.f, 0.f};
    if (ln.dy) {
        const f32x4 dy = *reinterpret_cast<const f32x4*>(ln.dy + m * ln.lddy + ch);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(ln.gamma + e);
        const f32x4 hv = *reinterpret_cast<const f32x4*>(ln.h + idx);
        const float mean = ln.stat[b * 2], rstd = ln.stat[b * 2 + 1], m1 = sums[0], m2 = sums[1];
#pragma unroll
        for (int k = 0; k < 4; ++k) dh[k] = rstd * (dy[k] * gm[k] - m1 - (hv[k] - mean) * rstd * m2);
    } else if (dh_a) {
        dh = *reinterpret_cast<const f32x4*>(dh_a + m * lda + ch);
    }
    if (dh_b) dh += *reinterpret_cast<const f32x4*>(dh_b + m * ldb + ch);
    f32x4 dcv = {0.f, 0.f, 0.f, 0.f};
    if (dc_valid) dcv = *reinterpret_cast<const f32x4*>(dc + idx);
    f32x4 oj, oi, of, oo, dcn;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float tc = fast_tanh_b(cn[k]);
        const float dct = dh[k] * ao[k] * (1.f - tc * tc) + dcv[k];
        oj[k] = dct * ai[k] * (1.f - aj[k] * aj[k]);
        oi[k] = dct * aj[k] * ai[k] * (1.f - ai[k]);
        of[k] = dct * co[k] * af[k] * (1.f - af[k]);
        oo[k] = dh[k] * tc * ao[k] * (1.f - ao[k]);
        dcn[k] = dct * af[k];
    }
    float* o = dG + m * 4 * C + ch;
    *reinterpret_cast<f32x4*>(o) = oj; *reinterpret_cast<f32x4*>(o + C) = oi;
    *reinterpret_cast<f32x4*>(o + 2 * C) = of; *reinterpret_cast<f32x4*>(o + 3 * C) = oo;
    *reinterpret_cast<f32x4*>(dc + idx) = dcn;
}

int lstm_gates_bwd(const float* gates, const float* c_old, const float* c_new, const float* dh_a, int lda,
                   const float* dh_b, int ldb, float* dc, int dc_valid, float* dG, int M, int C, hipStream_t s, int B, const LnFuse* ln,
                   float* zero, long long zero_floats) {
    PIVP_CHECK_ARG(gates && c_old && c_new && dc && dG && M > 0 && C > 0 && C % 4 == 0 && (dh_a || dh_b || (ln && ln->dy)));
    PIVP_CHECK_ARG(B > 0 && M % B == 0 && (!dh_a || lda % 4 == 0) && (!dh_b || ldb % 4 == 0));
    PIVP_CHECK_ARG(!zero || (zero_floats > 0 && zero_floats % 4 == 0 && ((uintptr_t)zero & 15) == 0));
    LnFuse lf;
    memset(&lf, 0, sizeof(lf));
    if (ln && ln->dy) {
        PIVP_CHECK_ARG(ln->gamma && ln->stat && ln->partials && ln->h && ln->S > 0 && ln->lddy % 4 == 0);
        lf = *ln;
    }
    const int npix = M / B;
    const int xb = (npix * C / 4 + 255) / 256;
    hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3(xb, B), dim3(256), 0, s, gates, c_old, c_new, dh_a, lda, dh_b, ldb, dc,
                       dc_valid, dG, npix, C, lf, zero, zero ? zero_floats / 4 : 0);
    return PIVP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
// db[n] += sum over pixels of dY[pix][n]   (bias gradient of every conv / deconv / 1x1)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy, int ld, int N, int M, float* __restrict__ db) {
    // block = 128 columns (32 float4 lanes) x 8 row-lanes; rows strided over gridDim.y
    __shared__ f32x4 part[8][33];
    const int c4 = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int col = blockIdx.x * 128 + c4 * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        // 8 independent 16-B loads in flight per thread (one load per trip left each wave waiting on its own previous load)
        const int step = gridDim.y * 8;
        int r = blockIdx.y * 8 + ty;
        for (; r + 7 * step < M; r += 8 * step) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(dy + (size_t)(r + u * step) * ld + col);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; r < M; r += step) acc += *reinterpret_cast<const f32x4*>(dy + (size_t)r * ld + col);
    }
    part[ty][c4] = acc;
    __syncthreads();
    if (ty == 0 && col < N) {
        f32x4 t = part[0][c4];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += part[i][c4];
#pragma unroll
        for (int e = 0; e < 4; ++e) if (col + e < N) atomicAdd(db + col + e, t[e]);
    }
}

int bias_grad(const float* dy, int ld, int N, int M, float* db, hipStream_t s) {
    PIVP_CHECK_ARG(dy && db && N > 0 && M > 0 && ld >= N && N % 4 == 0 && ld % 4 == 0);
    const int xb = (N + 127) / 128;
    int yb = (M + 511) / 512; if (yb > 128) yb = 128; if (yb < 1) yb = 1;   // few row-groups: their sums meet in atomics on N addresses
    hipLaunchKernelGGL(bias_grad_kernel, dim3(xb, yb), dim3(256), 0, s, dy, ld, N, M, db);
    return PIVP_LAUNCH_STATUS();
}

// dy[pix][c] *= (y[pix][c] > 0)   (backward of the ReLU fused into a producer, TM:697-700)
__global__ __launch_bounds__(256) void relu_mask_kernel(float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy,
                                                        int C, long npix, const float* __restrict__ add, int ldadd) {
    PIVP_SET_MAIN_PRIO();
    const long total = npix * (C / 4);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / (C / 4); const int c = (int)(i - p * (C / 4)) * 4;
        f32x4 g = *reinterpret_cast<f32x4*>(dy + p * lddy + c);
        if (add) g += *reinterpret_cast<const f32x4*>(add + p * ldadd + c);
        const f32x4 v = *reinterpret_cast<const f32x4*>(y + p * ldy + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = v[e] > 0.f ? g[e] : 0.f;
        *reinterpret_cast<f32x4*>(dy + p * lddy + c) = g;
    }
}

int relu_mask(float* dy, int lddy, const float* y, int ldy, int C, long npix, hipStream_t s, const float* add, int ldadd) {
    PIVP_CHECK_ARG(dy && y && C > 0 && C % 4 == 0 && npix > 0 && lddy % 4 == 0 && ldy % 4 == 0 && (!add || ldadd % 4 == 0));
    const long total = npix * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(relu_mask_kernel, dim3(blocks), dim3(256), 0, s, dy, lddy, y, ldy, C, npix, add, ldadd);
    return PIVP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
// LayerNormalizationConv2D backward (TM:203-208): y = xhat*gamma + beta, xhat = (x - mean)*rstd per sample over
// n = C*H*W elements (+ optional ReLU after it).  With g = dy*gamma:
//   dx = rstd * (g - mean(g) - xhat * mean(g*xhat));  dgamma[e] += sum_b dy*xhat;  dbeta[e] += sum_b dy.
// dy (and the ReLU mask source y) may be channel slices of concat buffers: element e of sample b sits at
// (b*npix + e/C)*ld + e%C.  Three launches: per-slice partial sums, dx, parameter gradients.
// ------------------------------------------------------------------------------------------
constexpr int LNB_SLICE = 1024;     // elements of a sample per block of the sums kernels (= 256 threads x 4)

__global__ __launch_bounds__(256) void ln_bwd_stats_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy,
                                                           const float* __restrict__ x, const float* __restrict__ stat,
                                                           const float* __restrict__ gamma, float* __restrict__ partials,
                                                           int n, int C, int relu) {
    PIVP_SET_MAIN_PRIO();
    __shared__ float red[4];
    const int sl = blockIdx.x, b = blockIdx.y, S = gridDim.x;
    const float mean = stat[b * 2], rstd = stat[b * 2 + 1];
    const int base = sl * LNB_SLICE, cnt = min(LNB_SLICE, n - base);
    const size_t pix0 = (size_t)b * (n / C);
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x * 4; i < cnt; i += 1024) {
        const int e = base + i, pix = e / C, ch = e - pix * C;
        f32x4 g = *reinterpret_cast<const f32x4*>(dy + (pix0 + pix) * lddy + ch);
        if (relu) {
            const f32x4 yy = *reinterpret_cast<const f32x4*>(y + (pix0 + pix) * ldy + ch);
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = yy[k] > 0.f ? g[k] : 0.f;
        }
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)b * n + e);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + e);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gg = g[k] * gm[k];
            s1 += gg; s2 = fmaf(gg, (xv[k] - mean) * rstd, s2);
        }
    }
    s1 = block_sum<256>(s1, red);
    s2 = block_sum<256>(s2, red);
    if (threadIdx.x == 0) { partials[((size_t)b * S + sl) * 2] = s1; partials[((size_t)b * S + sl) * 2 + 1] = s2; }
}

__global__ __launch_bounds__(256) void ln_bwd_apply_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy,
                                                           const float* __restrict__ x, const float* __restrict__ stat,
                                                           const float* __restrict__ gamma, const float* __restrict__ partials,
                                                           float* __restrict__ dx, int n, int C, int relu) {
    PIVP_SET_MAIN_PRIO();
    __shared__ float sums[2];
    const int sl = blockIdx.x, b = blockIdx.y, S = gridDim.x;
    if (threadIdx.x < 64) {
        float a = 0.f, c2 = 0.f;
        for (int i = threadIdx.x; i < S; i += 64) { a += partials[((size_t)b * S + i) * 2]; c2 += partials[((size_t)b * S + i) * 2 + 1]; }
        a = wave_sum(a); c2 = wave_sum(c2);
        if (threadIdx.x == 0) { sums[0] = a / (float)n; sums[1] = c2 / (float)n; }
    }
    __syncthreads();
    const float m1 = sums[0], m2 = sums[1];
    const float mean = stat[b * 2], rstd = stat[b * 2 + 1];
    const int base = sl * LNB_SLICE, cnt = min(LNB_SLICE, n - base);
    const size_t pix0 = (size_t)b * (n / C);
    for (int i = threadIdx.x * 4; i < cnt; i += 1024) {
        const int e = base + i, pix = e / C, ch = e - pix * C;
        f32x4 g = *reinterpret_cast<const f32x4*>(dy + (pix0 + pix) * lddy + ch);
        if (relu) {
            const f32x4 yy = *reinterpret_cast<const f32x4*>(y + (pix0 + pix) * ldy + ch);
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = yy[k] > 0.f ? g[k] : 0.f;
        }
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)b * n + e);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + e);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = rstd * (g[k] * gm[k] - m1 - (xv[k] - mean) * rstd * m2);
        *reinterpret_cast<f32x4*>(dx + (size_t)b * n + e) = o;
    }
}

__global__ __launch_bounds__(256) void ln_bwd_params_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy,
                                                            const float* __restrict__ x, const float* __restrict__ stat,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int B, int n, int C, int relu, float* __restrict__ part) {
    // thread = 4 consecutive elements x one group of samples (blockIdx.y).  Groups combine with atomics, or -- with `part`, the
    // plan's path -- every group adds into its own [2][n] plane of the partial buffer with plain loads and stores (launches of one
    // norm are stream-ordered), and ln_bwd_params_reduce sums the planes once per sweep: 8 atomics per element per timestep less
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= n) return;
    const int pix = e / C, ch = e - pix * C;
    f32x4 ag = {0.f, 0.f, 0.f, 0.f}, ab = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4      // the samples' loads are independent: keep four samples' worth in flight (the rolled loop waited for each sample's pair)
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        const size_t pb = (size_t)b * (n / C) + pix;
        f32x4 g = *reinterpret_cast<const f32x4*>(dy + pb * lddy + ch);
        if (relu) {
            const f32x4 yy = *reinterpret_cast<const f32x4*>(y + pb * ldy + ch);
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = yy[k] > 0.f ? g[k] : 0.f;
        }
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)b * n + e);
        const float mean = stat[b * 2], rstd = stat[b * 2 + 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) { ag[k] = fmaf(g[k], (xv[k] - mean) * rstd, ag[k]); ab[k] += g[k]; }
    }
    if (part) {
        float* pg = part + (size_t)blockIdx.y * 2 * n + e;
        f32x4 og = *reinterpret_cast<f32x4*>(pg), ob = *reinterpret_cast<f32x4*>(pg + n);
        og += ag; ob += ab;
        *reinterpret_cast<f32x4*>(pg) = og; *reinterpret_cast<f32x4*>(pg + n) = ob;
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { atomicAdd(dgamma + e + k, ag[k]); atomicAdd(dbeta + e + k, ab[k]); }
}

// ln_bwd_stats_kernel and ln_bwd_params_kernel (its `part` form) in one pass over dy and x: a block owns one slice of LNB_SLICE elements
// and the samples b = blockIdx.y, + G, + 2G, ...; a thread keeps the parameter-gradient sums of its 4 elements over those samples and the
// block reduces each sample's (sum g, sum g xhat) over the slice.  71 launches per train step less than the two kernels (the sweep is
// made of 5-6 us launches here).  Loads are unconditional (indices clamped, contributions weighted by 0) so that four samples' loads stay
// in flight.
#ifndef LNB_SP_NJ
#define LNB_SP_NJ 4                  // samples in flight per trip (2: 64 registers instead of 88 -- a wave then fits on a SIMD beside two waves of the bf16 mode's
                                     // weight gradient; measured, profiles/r04/NOTES.md: no difference in any mode's train step)
#endif
__global__ __launch_bounds__(256) void ln_bwd_sums_params_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy,
                                                                 const float* __restrict__ x, const float* __restrict__ stat,
                                                                 const float* __restrict__ gamma, float* __restrict__ partials,
                                                                 int B, int n, int C, int relu, float* __restrict__ part) {
    PIVP_SET_MAIN_PRIO();
    constexpr int NJ = LNB_SP_NJ;
    __shared__ float red[4][2 * NJ];
    const int e0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const bool valid = e0 < n;
    const int e = valid ? e0 : 0;
    const int pix = e / C, ch = e - pix * C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = gridDim.x, G = gridDim.y, npix = n / C;
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + e);
    f32x4 ag = {0.f, 0.f, 0.f, 0.f}, ab = {0.f, 0.f, 0.f, 0.f};
    for (int b0 = blockIdx.y; b0 < B; b0 += NJ * G) {
        float sm[2 * NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int bj = b0 + j * G;
            const int b = bj < B ? bj : B - 1;
            const float wgt = (bj < B && valid) ? 1.f : 0.f;
            const size_t pb = (size_t)b * npix + pix;
            f32x4 g = *reinterpret_cast<const f32x4*>(dy + pb * lddy + ch);
            if (relu) {
                const f32x4 yy = *reinterpret_cast<const f32x4*>(y + pb * ldy + ch);
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = yy[k] > 0.f ? g[k] : 0.f;
            }
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)b * n + e);
            const float mean = stat[b * 2], rstd = stat[b * 2 + 1];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gk = g[k] * wgt, xh = (xv[k] - mean) * rstd, gg = gk * gm[k];
                ag[k] = fmaf(gk, xh, ag[k]); ab[k] += gk;
                s1 += gg; s2 = fmaf(gg, xh, s2);
            }
            sm[2 * j] = wave_sum(s1); sm[2 * j + 1] = wave_sum(s2);
        }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 2 * NJ; ++q) red[wave][q] = sm[q];
        }
        __syncthreads();
        if (threadIdx.x < 2 * NJ) {
            const int b = b0 + (threadIdx.x >> 1) * G;
            if (b < B) partials[((size_t)b * S + blockIdx.x) * 2 + (threadIdx.x & 1)] =
                (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        }
        __syncthreads();
    }
    if (!valid) return;
    float* pg = part + (size_t)blockIdx.y * 2 * n + e;
    f32x4 og = *reinterpret_cast<f32x4*>(pg), ob = *reinterpret_cast<f32x4*>(pg + n);
    og += ag; ob += ab;
    *reinterpret_cast<f32x4*>(pg) = og; *reinterpret_cast<f32x4*>(pg + n) = ob;
}

// dgamma, dbeta += the sum of the sample-group planes of ln_bwd_params_kernel's partial buffer ([groups][2][n])
__global__ __launch_bounds__(256) void ln_bwd_params_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, int n, int groups) {
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= n) return;
    f32x4 ag = {0.f, 0.f, 0.f, 0.f}, ab = {0.f, 0.f, 0.f, 0.f};
    for (int y = 0; y < groups; ++y) {
        ag += *reinterpret_cast<const f32x4*>(part + (size_t)y * 2 * n + e);
        ab += *reinterpret_cast<const f32x4*>(part + (size_t)y * 2 * n + n + e);
    }
    f32x4 og = *reinterpret_cast<f32x4*>(dgamma + e), ob = *reinterpret_cast<f32x4*>(dbeta + e);
    og += ag; ob += ab;
    *reinterpret_cast<f32x4*>(dgamma + e) = og; *reinterpret_cast<f32x4*>(dbeta + e) = ob;
}

static int ln_bwd_param_groups(int B, int n) {
    const int xb = (n / 4 + 255) / 256;
    int yb = 512 / xb; if (yb < 1) yb = 1; if (yb > B) yb = B; if (yb > 8) yb = 8;
    return yb;
}
long long ln_bwd_param_part_floats(int n) { return 8LL * 2 * n; }      // up to 8 sample groups
int ln_bwd_params_reduce(const float* part, float* dgamma, float* dbeta, int B, int n, hipStream_t s) {
    PIVP_CHECK_ARG(part && dgamma && dbeta && B > 0 && n > 0 && n % 4 == 0);
    hipLaunchKernelGGL(ln_bwd_params_reduce_kernel, dim3((n / 4 + 255) / 256), dim3(256), 0, s, part, dgamma, dbeta, n, ln_bwd_param_groups(B, n));
    return PIVP_LAUNCH_STATUS();
}

int ln_bwd_slices(int n) { return (n + LNB_SLICE - 1) / LNB_SLICE; }

int ln_backward(const float* dy, int lddy, const float* y, int ldy, const float* x, const float* stat, const float* gamma,
                float* partials, float* dx, float* dgamma, float* dbeta, int B, int n, int C, int relu, hipStream_t s, float* param_part) {
    // dx == nullptr: the consumer forms dx itself from the partials (lstm_gates_bwd with LnFuse); only the sums and the parameter gradients run here
    PIVP_CHECK_ARG(dy && x && stat && gamma && partials && dgamma && dbeta && B > 0 && n > 0 && C > 0 && C % 4 == 0 && n % C == 0);
    PIVP_CHECK_ARG(lddy >= C && lddy % 4 == 0 && (!relu || (y && ldy >= C && ldy % 4 == 0)));
    const int S = ln_bwd_slices(n);
    if (param_part) {   // the plan's path: sums and parameter-gradient planes in one launch
        hipLaunchKernelGGL(ln_bwd_sums_params_kernel, dim3(S, ln_bwd_param_groups(B, n)), dim3(256), 0, s, dy, lddy, y, ldy, x, stat, gamma,
                           partials, B, n, C, relu, param_part);
        if (dx) hipLaunchKernelGGL(ln_bwd_apply_kernel, dim3(S, B), dim3(256), 0, s, dy, lddy, y, ldy, x, stat, gamma, partials, dx, n, C, relu);
        return PIVP_LAUNCH_STATUS();
    }
    hipLaunchKernelGGL(ln_bwd_stats_kernel, dim3(S, B), dim3(256), 0, s, dy, lddy, y, ldy, x, stat, gamma, partials, n, C, relu);
    if (dx) hipLaunchKernelGGL(ln_bwd_apply_kernel, dim3(S, B), dim3(256), 0, s, dy, lddy, y, ldy, x, stat, gamma, partials, dx, n, C, relu);
    {
        const int xb = (n / 4 + 255) / 256, yb = ln_bwd_param_groups(B, n);
        hipLaunchKernelGGL(ln_bwd_params_kernel, dim3(xb, yb), dim3(256), 0, s, dy, lddy, y, ldy, x, stat, dgamma, dbeta, B, n, C, relu, param_part);
    }
    return PIVP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
// Chainer 2 Adam (optimizers.Adam at TM:860; AdamRule.update_core): m += (1-b1)(g-m); v += (1-b2)(g*g-v);
// p -= alpha*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)   -- eps is added to the UNcorrected sqrt(v).
// One launch over the flat parameter buffer; `lr_t` is the bias-corrected step size computed on the host.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr_t, float omb1, float omb2, float eps,
                                                   float gscale) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 3 < n) {
            f32x4 pp = *reinterpret_cast<f32x4*>(p + i), gg = *reinterpret_cast<const f32x4*>(g + i);
            f32x4 mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gk = gg[k] * gscale;
                mm[k] += omb1 * (gk - mm[k]);
                vv[k] += omb2 * (gk * gk - vv[k]);
                pp[k] -= lr_t * mm[k] / (sqrtf(vv[k]) + eps);
            }
            *reinterpret_cast<f32x4*>(p + i) = pp; *reinterpret_cast<f32x4*>(m + i) = mm; *reinterpret_cast<f32x4*>(v + i) = vv;
        } else {
            for (long j = i; j < n; ++j) {
                const float gk = g[j] * gscale;
                m[j] += omb1 * (gk - m[j]);
                v[j] += omb2 * (gk * gk - v[j]);
                p[j] -= lr_t * m[j] / (sqrtf(v[j]) + eps);
            }
        }
    }
}

int adam_step(float* p, const float* g, float* m, float* v, long n, double lr_t, double beta1, double beta2, double eps,
              double gscale, hipStream_t s) {
    PIVP_CHECK_ARG(p && g && m && v && n > 0);
    const long blocks = (n / 4 + 255) / 256;
    // (1 - beta) is formed in double like Chainer's Python float, then rounded once to fp32
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, p, g, m, v, n, (float)lr_t,
                       (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)gscale);
    return PIVP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
// bf16 payload of the data-parallel gradient all-reduce (BASELINE.json config 3; SURVEY.md 8e: 18.4 MB instead of 36.9 MB per
// step).  pack: fp32 gradient slice -> bf16 send buffer (round to nearest even, NaN stays NaN: the hardware convert); unpack: the
// summed bf16 slice back into the fp32 flat gradient buffer, which stays the optimizer's (fp32 "master") input.  HBM-bound
// streaming kernels: 16 B per lane on the wide side.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void grad_pack_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long n) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 3 < n) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
            *reinterpret_cast<bf16x4_t*>(dst + i) = __builtin_convertvector(v, bf16x4_t);
        } else {
            for (long j = i; j < n; ++j) dst[j] = __builtin_bit_cast(unsigned short, (__bf16)src[j]);
        }
    }
}
__global__ __launch_bounds__(256) void grad_unpack_bf16_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst, long n) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 3 < n) {
            const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(src + i);
            *reinterpret_cast<f32x4*>(dst + i) = __builtin_convertvector(v, f32x4);
        } else {
            for (long j = i; j < n; ++j) dst[j] = (float)__builtin_bit_cast(__bf16, src[j]);
        }
    }
}
// Data-parallel reduce-scatter, local half (parallel.py: GradAllReduce(algo='rs_ag')): `nsh` shards of `len` elements, one per peer
// (bf16 as they travelled, or fp32), summed in fp32 in the fixed order 0 .. nsh-1 and rounded ONCE to the output type.  An all-reduce
// of a bf16 buffer rounds the running sum at every hop instead (7 roundings at 8 ranks).
template <bool IN_BF16, bool OUT_BF16>
__global__ __launch_bounds__(256) void grad_sum_shards_kernel(const void* __restrict__ src_, void* __restrict__ dst_, int nsh, long len) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < len; i += (long)gridDim.x * 1024) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < nsh; ++k) {
            f32x4 v;
            if (IN_BF16) v = __builtin_convertvector(*reinterpret_cast<const bf16x4_t*>((const unsigned short*)src_ + (long)k * len + i), f32x4);
            else v = *reinterpret_cast<const f32x4*>((const float*)src_ + (long)k * len + i);
            acc += v;
        }
        if (OUT_BF16) *reinterpret_cast<bf16x4_t*>((unsigned short*)dst_ + i) = __builtin_convertvector(acc, bf16x4_t);
        else *reinterpret_cast<f32x4*>((float*)dst_ + i) = acc;
    }
}
int grad_sum_shards(const void* src, int src_bf16, int nsh, long len, void* dst, int dst_bf16, hipStream_t s) {
    PIVP_CHECK_ARG(src && dst && nsh > 0 && len > 0 && len % 4 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0);
    PIVP_CHECK_ARG(!src_bf16 || (len * 2) % 16 == 0);       // every shard starts 16-B aligned
    const long blocks = (len / 4 + 255) / 256;
    const dim3 grid((unsigned)(blocks < 2048 ? blocks : 2048));
    if (src_bf16 && dst_bf16) hipLaunchKernelGGL((grad_sum_shards_kernel<true, true>), grid, dim3(256), 0, s, src, dst, nsh, len);
    else if (src_bf16) hipLaunchKernelGGL((grad_sum_shards_kernel<true, false>), grid, dim3(256), 0, s, src, dst, nsh, len);
    else if (dst_bf16) hipLaunchKernelGGL((grad_sum_shards_kernel<false, true>), grid, dim3(256), 0, s, src, dst, nsh, len);
    else hipLaunchKernelGGL((grad_sum_shards_kernel<false, false>), grid, dim3(256), 0, s, src, dst, nsh, len);
    return PIVP_LAUNCH_STATUS();
}
int grad_pack_bf16(const float* src, void* dst, long n, hipStream_t s) {
    PIVP_CHECK_ARG(src && dst && n > 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0);
    const long blocks = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(grad_pack_bf16_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, src, (unsigned short*)dst, n);
    return PIVP_LAUNCH_STATUS();
}
int grad_unpack_bf16(const void* src, float* dst, long n, hipStream_t s) {
    PIVP_CHECK_ARG(src && dst && n > 0 && ((uintptr_t)dst & 15) == 0 && ((uintptr_t)src & 7) == 0);
    const long blocks = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(grad_unpack_bf16_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, (const unsigned short*)src, dst, n);
    return PIVP_LAUNCH_STATUS();
}

}  // namespace pivp

PIVP_DEFINE_MAIN_PRIO_SETTER(backward)
