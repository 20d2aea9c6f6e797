// HBM-bound helper kernels of the per-timestep program: enc0 conv, flattened-CHW LayerNorm,
// enc3 (smear + 1x1) with the state predictor, loss / PSNR reductions, layout taps.
// Reference call sites are cited per kernel ("TM" = src/models/train_model.py).
#include "pivp_kernels.h"

namespace pivp {

// ------------------------------------------------------------------------------------------
// enc0: L.Convolution2D(32, (5,5), stride=2, pad=2) on the 3-channel frame (TM:500, run at TM:595).
// img planar [B][3][H][W]; w [75][32] with k = (ky*5+kx)*3 + ci; out NHWC [B][H/2][W/2][32].
// 64 output pixels x (4 groups of 8 channels) per 256-thread block; weights live in LDS.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_enc0_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int B, int H, int W) {
    __shared__ __attribute__((aligned(16))) float wl[75 * 32];
    for (int i = threadIdx.x; i < 75 * 32; i += 256) wl[i] = w[i];
    __syncthreads();
    const int H2 = H >> 1, W2 = W >> 1;
    const int total = B * H2 * W2;
    const int pix = blockIdx.x * 64 + (threadIdx.x >> 2);
    const int cg = threadIdx.x & 3;
    if (pix >= total) return;
    const int b = pix / (H2 * W2);
    const int rem = pix - b * H2 * W2;
    const int oy = rem / W2, ox = rem - oy * W2;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = bias[cg * 8 + o];
    const float* ib = img + (size_t)b * 3 * H * W;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
        const int iy = 2 * oy - 2 + ky;
        if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            const int ix = 2 * ox - 2 + kx;
            if ((unsigned)ix >= (unsigned)W) continue;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float v = ib[(size_t)ci * H * W + iy * W + ix];
                const float* wr = wl + ((ky * 5 + kx) * 3 + ci) * 32 + cg * 8;
#pragma unroll
                for (int o = 0; o < 8; ++o) acc[o] = fmaf(v, wr[o], acc[o]);
            }
        }
    }
    float* op = out + (size_t)pix * 32 + cg * 8;
    *reinterpret_cast<f32x4*>(op) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *reinterpret_cast<f32x4*>(op + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
}

// Row-tile version (W/2 in {8, 16, 32, 64}): a block owns 64 output pixels = whole output rows; its input patch
// ((2*rows+3) x (W+3) x 3, zero padded) and the weights are staged in LDS with every load of a thread in flight at
// once.  The element-per-thread kernel above issued 75 image loads behind bounds branches and re-read the weights with
// one load in flight: ~11 us for 6 MB of traffic.
constexpr int E0_PR = 4;   // patch rows per thread and channel
__global__ __launch_bounds__(256) void conv_enc0_rows_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             int B, int H, int W, float* __restrict__ ln_part) {
    extern __shared__ __attribute__((aligned(16))) float sm0[];
    float* wl = sm0;                 // [75][32]
    float* patch = sm0 + 75 * 32;    // [3][R][PWc]
    const int tid = threadIdx.x;
    const int H2 = H >> 1, W2 = W >> 1;
    const int rows = 64 / W2, R = 2 * rows + 3, PWc = W + 3;
    const int tiles_per_img = H2 / rows;
    const int b = blockIdx.x / tiles_per_img, oy0 = (blockIdx.x - b * tiles_per_img) * rows;
    const float* ib = img + (size_t)b * 3 * H * W;
    {
        f32x4 tw[3];
        float tp[3][E0_PR];
        const int x = tid % PWc, r0 = tid / PWc, rstep = 256 / PWc;   // two divisions per thread, none per element
        const bool lane_on = tid < rstep * PWc;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int f = min(tid + 256 * j, 599) * 4;                       // 600 float4 of weights (clamped: no branch)
            tw[j] = *reinterpret_cast<const f32x4*>(w + f);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int u = 0; u < E0_PR; ++u) {
                const int r = r0 + u * rstep, iy = 2 * oy0 - 2 + r, ix = x - 2;
                const bool ok = lane_on && r < R && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                tp[c][u] = ok ? ib[(size_t)c * H * W + iy * W + ix] : 0.f;
            }
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (tid + 256 * j < 600) *reinterpret_cast<f32x4*>(wl + (tid + 256 * j) * 4) = tw[j];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int u = 0; u < E0_PR; ++u) {
                const int r = r0 + u * rstep;
                if (lane_on && r < R) patch[(c * R + r) * PWc + x] = tp[c][u];
            }
    }
    __syncthreads();
    const int pl = tid >> 2, cg = tid & 3;
    const int ry = pl / W2, ox = pl - ry * W2;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = bias[cg * 8 + o];
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float v = patch[(ci * R + 2 * ry + ky) * PWc + 2 * ox + kx];
                const float* wr = wl + ((ky * 5 + kx) * 3 + ci) * 32 + cg * 8;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr), w1 = *reinterpret_cast<const f32x4*>(wr + 4);
                acc[0] = fmaf(v, w0[0], acc[0]); acc[1] = fmaf(v, w0[1], acc[1]); acc[2] = fmaf(v, w0[2], acc[2]); acc[3] = fmaf(v, w0[3], acc[3]);
                acc[4] = fmaf(v, w1[0], acc[4]); acc[5] = fmaf(v, w1[1], acc[5]); acc[6] = fmaf(v, w1[2], acc[6]); acc[7] = fmaf(v, w1[3], acc[7]);
            }
    float* op = out + (((size_t)b * H2 + oy0 + ry) * W2 + ox) * 32 + cg * 8;
    *reinterpret_cast<f32x4*>(op) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *reinterpret_cast<f32x4*>(op + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
    if (ln_part) {   // LayerNorm partial (count, mean, M2) of the block's 64 x 32 outputs (norm_enc0 follows, TM:595)
        __shared__ float red[4];
        float s1 = 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) s1 += acc[o];
        const float mean = block_sum<256>(s1, red) * (1.0f / 2048.0f);
        float q = 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) { const float dd = acc[o] - mean; q = fmaf(dd, dd, q); }
        q = block_sum<256>(q, red);
        if (tid == 0) {
            float* pp = ln_part + (size_t)blockIdx.x * 4;   // blocks of a sample are consecutive: [b][tile]
            pp[0] = 2048.f; pp[1] = mean; pp[2] = q; pp[3] = 0.f;
        }
    }
}

int conv_enc0(const float* img, const float* w, const float* bias, float* out, int B, int H, int W, hipStream_t s, float* ln_part,
              int ln_cap, int* ln_nparts) {
    if (ln_nparts) *ln_nparts = 0;
    PIVP_CHECK_ARG(img && w && bias && out && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0);
    const int H2 = H / 2, W2 = W / 2;
    const int total = B * H2 * W2;
    if ((W2 == 8 || W2 == 16 || W2 == 32 || W2 == 64) && H2 % (64 / W2) == 0 && ((uintptr_t)w & 15) == 0) {
        const int rows = 64 / W2, R = 2 * rows + 3, npatch = 3 * R * (W + 3);
        if (W + 3 <= 256 && R <= E0_PR * (256 / (W + 3))) {
            const size_t lds = sizeof(float) * (75 * 32 + npatch);
            const int np = H2 / rows;   // row tiles = LayerNorm partials per sample
            float* lp = (ln_part && np <= ln_cap) ? ln_part : nullptr;
            if (lp && ln_nparts) *ln_nparts = np;
            hipLaunchKernelGGL(conv_enc0_rows_kernel, dim3(B * np), dim3(256), lds, s, img, w, bias, out, B, H, W, lp);
            return PIVP_LAUNCH_STATUS();
        }
    }
    hipLaunchKernelGGL(conv_enc0_kernel, dim3((total + 63) / 64), dim3(256), 0, s, img, w, bias, out, B, H, W);
    return PIVP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
// LayerNormalizationConv2D (TM:203-208): LayerNorm over the flattened C*H*W vector of each sample,
// gamma/beta of size C*H*W (here stored in NHWC-flat order to match the activations).
// Two launches: per-slice (count, mean, M2) partials with a local two-pass (values stay in
// registers), then an apply pass that merges the slices with Chan's formula.  Slices are 4096
// floats so a B=32, n=32768 LayerNorm fills the chip with 256 blocks.
// ------------------------------------------------------------------------------------------
constexpr int LN_SLICE = 4096;

int ln_stats_slices(int n) { return (n + LN_SLICE - 1) / LN_SLICE; }

__global__ __launch_bounds__(256) void ln_stats_kernel(const float* __restrict__ x, float* __restrict__ partials, int n) {
    __shared__ float red[4];
    const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
    const float* xb = x + (size_t)b * n;
    const int base = s * LN_SLICE;
    const int cnt = min(LN_SLICE, n - base);
    f32x4 v[4];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = (j * 256 + threadIdx.x) * 4;
        if (i < cnt) {
            v[j] = *reinterpret_cast<const f32x4*>(xb + base + i);
            sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        } else {
            v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float mean = block_sum<256>(sum, red) / (float)cnt;
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = (j * 256 + threadIdx.x) * 4;
        if (i < cnt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float dd = v[j][e] - mean; m2 = fmaf(dd, dd, m2); }
        }
    }
    m2 = block_sum<256>(m2, red);
    if (threadIdx.x == 0) {
        float* p = partials + ((size_t)b * S + s) * 4;
        p[0] = (float)cnt; p[1] = mean; p[2] = m2; p[3] = 0.f;
    }
}

int ln_stats(const float* x, float* partials, int B, int n, hipStream_t s) {
    PIVP_CHECK_ARG(x && partials && B > 0 && n > 0 && n % 4 == 0);
    hipLaunchKernelGGL(ln_stats_kernel, dim3(ln_stats_slices(n), B), dim3(256), 0, s, x, partials, n);
    return PIVP_LAUNCH_STATUS();
}

__global__ __launch_bounds__(256) void ln_apply_kernel(const float* __restrict__ x, const float* __restrict__ partials,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ out, int n, int C, int ldo, float eps, int relu,
                                                       float* __restrict__ stat_out, int S) {
    __shared__ float stat[2];
    const int s = blockIdx.x, b = blockIdx.y;   // S partials per sample: ln_stats slices, or the tiles of a producer
    // The slice's x, gamma and beta do not depend on the statistics: all twelve 16-B loads of a thread are requested first (clamped,
    // unpredicated), so the merge of the partials below runs under their round trip instead of in front of four more.
    const float* xb = x + (size_t)b * n;
    const int base = s * LN_SLICE;
    const int cnt = min(LN_SLICE, n - base);    // multiple of 4, >= 4
    f32x4 v[4], g[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = base + min((j * 256 + (int)threadIdx.x) * 4, cnt - 4);
        v[j] = *reinterpret_cast<const f32x4*>(xb + idx);
        g[j] = *reinterpret_cast<const f32x4*>(gamma + idx);
        be[j] = *reinterpret_cast<const f32x4*>(beta + idx);
    }
    if (threadIdx.x < 64) {
        float mean, rstd;
        if (S > 0) ln_merge_partials(partials, b, S, eps, mean, rstd);
        else { mean = partials[b * 2]; rstd = partials[b * 2 + 1]; }   // S == 0: `partials` is a saved [B][2] (mean, rstd)
        if (threadIdx.x == 0) {
            stat[0] = mean; stat[1] = rstd;
            if (stat_out && s == 0) { stat_out[b * 2] = mean; stat_out[b * 2 + 1] = rstd; }   // kept for the backward pass
        }
    }
    __syncthreads();
    const float mean = stat[0], rstd = stat[1];
    const size_t pix0 = (size_t)b * (n / C);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = (j * 256 + threadIdx.x) * 4;
        if (i < cnt) {
            const int idx = base + i;
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = (v[j][e] - mean) * rstd * g[j][e] + be[j][e];
                y[e] = relu ? fmaxf(t, 0.f) : t;
            }
            const int pix = idx / C, ch = idx - pix * C;
            *reinterpret_cast<f32x4*>(out + (pix0 + pix) * ldo + ch) = y;
        }
    }
}

int ln_apply(const float* x, const float* partials, const float* gamma, const float* beta, float* out,
             int B, int n, int C, int ldo, float eps, int relu, hipStream_t s, float* stat_out, int nparts) {
    PIVP_CHECK_ARG(x && partials && gamma && beta && out && B > 0 && n > 0 && C > 0 && C % 4 == 0 && n % C == 0);
    PIVP_CHECK_ARG(ldo >= C && ldo % 4 == 0 && nparts >= -1);
    // nparts: > 0 producer-written partials, 0 the ln_stats slices, -1 `partials` is a saved [B][2] (mean, rstd)
    hipLaunchKernelGGL(ln_apply_kernel, dim3(ln_stats_slices(n), B), dim3(256), 0, s,
                       x, partials, gamma, beta, out, n, C, ldo, eps, relu, stat_out,
                       nparts > 0 ? nparts : nparts == 0 ? ln_stats_slices(n) : 0);
    return PIVP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
// Group 3 of the op program (TM:598): smear(state_action) -> concat -> 1x1 conv (74 -> 64) -> ReLU.
// Algebraically the tiled 10-vector only adds a per-sample bias W[:,64:74].sa, so nothing is
// tiled or concatenated.  The same launch evaluates current_state = Linear(state_action) (TM:730).
// e2/e3 NHWC [B][HW8][64]; w3 [64 (+10)][64] K-major; wcs reference layout (5,10).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void enc3_state_kernel(const float* __restrict__ e2, const float* __restrict__ action,
                                                         const float* __restrict__ state, const float* __restrict__ w3,
                                                         const float* __restrict__ b3, const float* __restrict__ wcs,
                                                         const float* __restrict__ bcs, float* __restrict__ e3,
                                                         float* __restrict__ state_out, int HW8, int use_state) {
    __shared__ float xt[64 * 65];
    __shared__ __attribute__((aligned(16))) float wl[64 * 64];
    __shared__ float sb[64];
    __shared__ float sa[10];
    const int b = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    if (tid < 5) sa[tid] = action[b * 5 + tid];
    else if (tid < 10) sa[tid] = state[b * 5 + tid - 5];
    const int npx = min(64, HW8 - tile * 64);
    const float* src = e2 + ((size_t)b * HW8 + tile * 64) * 64;
    {   // all eight 16-B loads of a thread are issued before the first LDS store (one round trip, not thirty-two)
        f32x4 tw[4], tx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = (tid + 256 * j) * 4;   // float index in the 64 x 64 weight block / pixel tile
            tw[j] = *reinterpret_cast<const f32x4*>(w3 + f);
            tx[j] = (f >> 6) < npx ? *reinterpret_cast<const f32x4*>(src + f) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = (tid + 256 * j) * 4, p = f >> 6, k = f & 63;
            *reinterpret_cast<f32x4*>(wl + f) = tw[j];
            xt[p * 65 + k] = tx[j][0]; xt[p * 65 + k + 1] = tx[j][1]; xt[p * 65 + k + 2] = tx[j][2]; xt[p * 65 + k + 3] = tx[j][3];
        }
    }
    __syncthreads();
    if (tid < 64) {
        float v = b3[tid];
        if (use_state) {
#pragma unroll
            for (int j = 0; j < 10; ++j) v = fmaf(sa[j], w3[(64 + j) * 64 + tid], v);
        }
        sb[tid] = v;
    }
    if (tile == 0 && tid >= 64 && tid < 69) {
        const int o = tid - 64;
        float v = bcs[o];
#pragma unroll
        for (int j = 0; j < 10; ++j) v = fmaf(wcs[o * 10 + j], sa[j], v);
        state_out[b * 5 + o] = v;
    }
    __syncthreads();
    const int p = tid >> 2, cg = tid & 3;
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = sb[cg * 16 + o];
    for (int k = 0; k < 64; ++k) {
        const float xv = xt[p * 65 + k];
        const f32x4* wr = reinterpret_cast<const f32x4*>(wl + k * 64 + cg * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 wv = wr[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[q * 4 + e] = fmaf(xv, wv[e], acc[q * 4 + e]);
        }
    }
    if (p < npx) {
        float* op = e3 + ((size_t)b * HW8 + tile * 64 + p) * 64 + cg * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(op + q * 4) = f32x4{fmaxf(acc[q * 4], 0.f), fmaxf(acc[q * 4 + 1], 0.f),
                                                          fmaxf(acc[q * 4 + 2], 0.f), fmaxf(acc[q * 4 + 3], 0.f)};
    }
}

int enc3_state(const float* e2, const float* action, const float* state, const float* w3, const float* b3,
               const float* wcs, const float* bcs, float* e3, float* state_out,
               int B, int HW8, int use_state, hipStream_t s) {
    PIVP_CHECK_ARG(e2 && action && state && w3 && b3 && wcs && bcs && e3 && state_out && B > 0 && HW8 > 0);
    hipLaunchKernelGGL(enc3_state_kernel, dim3((HW8 + 63) / 64, B), dim3(256), 0, s,
                       e2, action, state, w3, b3, wcs, bcs, e3, state_out, HW8, use_state);
    return PIVP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
// Loss / PSNR (TM:737-759).  Deterministic two-stage reductions (no float atomics).
// ------------------------------------------------------------------------------------------
constexpr int LOSS_CHUNK = 8192;
int loss_partials_count(int n) { return (n + LOSS_CHUNK - 1) / LOSS_CHUNK; }

__global__ __launch_bounds__(256) void sqerr_partials_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             float* __restrict__ partials, int n) {
    __shared__ float red[4];
    const int base = blockIdx.x * LOSS_CHUNK;
    float acc = 0.f;
    for (int i = base + threadIdx.x * 4; i < min(n, base + LOSS_CHUNK); i += 1024) {
        if (i + 3 < n) {
            const f32x4 va = *reinterpret_cast<const f32x4*>(a + i), vb = *reinterpret_cast<const f32x4*>(b + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float dd = va[e] - vb[e]; acc = fmaf(dd, dd, acc); }
        } else {
            for (int e = i; e < n; ++e) { const float dd = a[e] - b[e]; acc = fmaf(dd, dd, acc); }
        }
    }
    acc = block_sum<256>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

int frame_sqerr_partials(const float* a, const float* b, float* partials, int n, hipStream_t s) {
    PIVP_CHECK_ARG(a && b && partials && n > 0 && n % 4 == 0);
    hipLaunchKernelGGL(sqerr_partials_kernel, dim3(loss_partials_count(n)), dim3(256), 0, s, a, b, partials, n);
    return PIVP_LAUNCH_STATUS();
}

// results: [0] loss, [1] psnr_all, [2..2+nf) recon_cost, [2+nf..2+2nf) psnr, [2+2nf..2+3nf) state_cost
// One WAVE per frame (round 6: one wave walked the frames one after the other -- two dependent round trips, a log and two wave sums per frame,
// 15-18 us at the end of every rollout); thread 0 then adds the per-frame terms in frame order (the same sum as before), reading them back
// from `results` behind the block barrier (any number of frames).
constexpr int LF_WAVES = 16;
__global__ __launch_bounds__(64 * LF_WAVES) void loss_finalize_kernel(const float* __restrict__ fp, int nparts, int nframes,
                                                                     int frame_numel, const float* __restrict__ st,
                                                                     const float* __restrict__ sg, int state_numel, float denom,
                                                                     float* __restrict__ results) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int f = wave; f < nframes; f += LF_WAVES) {
        float acc = 0.f;
        for (int i = lane; i < nparts; i += 64) acc += fp[f * nparts + i];
        float sacc = 0.f;
        for (int i = lane; i < state_numel; i += 64) {
            const float dd = st[f * state_numel + i] - sg[f * state_numel + i];
            sacc = fmaf(dd, dd, sacc);
        }
        acc = wave_sum(acc);
        sacc = wave_sum(sacc);
        const float mse = acc / (float)frame_numel;
        const float psnr = 10.0f * logf(1.0f / mse) / 2.302585092994046f;
        const float scost = sacc / (float)state_numel * 1e-4f;
        if (lane == 0) {
            results[2 + f] = mse;
            results[2 + nframes + f] = psnr;
            results[2 + 2 * nframes + f] = scost;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float loss = 0.f, psnr_all = 0.f;
        const volatile float* rv = results;          // written by the other waves of this block in front of the barrier
        for (int f = 0; f < nframes; ++f) { loss += rv[2 + f] + rv[2 + 2 * nframes + f]; psnr_all += rv[2 + nframes + f]; }
        results[0] = loss / denom; results[1] = psnr_all;
    }
}

int loss_finalize(const float* frame_partials, int nparts, int nframes, int frame_numel,
                  const float* states_true, const float* states_gen, int state_numel,
                  float denom, float* results, hipStream_t s) {
    PIVP_CHECK_ARG(frame_partials && states_true && states_gen && results && nparts > 0 && nframes >= 0 && denom != 0.f);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64 * LF_WAVES), 0, s, frame_partials, nparts, nframes, frame_numel,
                       states_true, states_gen, state_numel, denom, results);
    return PIVP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
// F.resize_images of the predict path (predict_model.py:119-122): bilinear, sample positions linspace(0, in-1, out)
// (align-corners), neighbours clamped to [0, in-2] as Chainer 2 does; planar [N][C][Hin][Win] -> [N][C][Hout][Wout], x scale.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int planes,
                                                              int Hin, int Win, int Hout, int Wout, float scale) {
    const long total = (long)planes * Hout * Wout;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int pl = (int)(i / (Hout * Wout)), rem = (int)(i - (long)pl * Hout * Wout), oy = rem / Wout, ox = rem - oy * Wout;
        const double u = Wout > 1 ? (double)ox * (double)(Win - 1) / (double)(Wout - 1) : 0.0;
        const double v = Hout > 1 ? (double)oy * (double)(Hin - 1) / (double)(Hout - 1) : 0.0;
        int u0 = (int)floor(u), v0 = (int)floor(v);
        u0 = min(max(u0, 0), Win - 2); v0 = min(max(v0, 0), Hin - 2);
        const float wu = (float)(u - u0), wv = (float)(v - v0);
        const float* p = in + (size_t)pl * Hin * Win;
        const float a = p[v0 * Win + u0], b = p[v0 * Win + u0 + 1], c = p[(v0 + 1) * Win + u0], d = p[(v0 + 1) * Win + u0 + 1];
        out[i] = scale * ((1.f - wu) * (1.f - wv) * a + wu * (1.f - wv) * b + (1.f - wu) * wv * c + wu * wv * d);
    }
}
int resize_bilinear(const float* in, float* out, int planes, int Hin, int Win, int Hout, int Wout, float scale, hipStream_t s) {
    PIVP_CHECK_ARG(in && out && planes > 0 && Hin >= 2 && Win >= 2 && Hout >= 1 && Wout >= 1);
    const long total = (long)planes * Hout * Wout;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, s,
                       in, out, planes, Hin, Win, Hout, Wout, scale);
    return PIVP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
// NHWC (pixel stride ld) -> planar NCHW, for the conv_res taps (TM:734) and tests.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int C, int HW, int ld) {
    __shared__ float t[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        t[r][tx] = (p < HW && c < C) ? in[((size_t)b * HW + p) * ld + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (c < C && p < HW) out[((size_t)b * C + c) * HW + p] = t[tx][r];
    }
}

int nhwc_to_nchw(const float* in, float* out, int B, int C, int HW, int ld, hipStream_t s) {
    PIVP_CHECK_ARG(in && out && B > 0 && C > 0 && HW > 0 && ld >= C);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, s, in, out, C, HW, ld);
    return PIVP_LAUNCH_STATUS();
}

}  // namespace pivp
