// Skinny Linear on flatten(hidden5) (the motion head's generator, TM:321-323 / TM:457-458): the K-slice partial-sum body of
// skinny_linear_partials_kernel (heads.hip).
#pragma once
#include "pivp_kernels.h"

namespace pivp {

constexpr int LIN_KS = 64;    // K per slice
constexpr int LIN_BG = 16;    // batch rows per block

// grid (K/64 slices, B/16 groups): 256 blocks at B = 32, K = 8192, so the 8 MB weight matrix streams across the
// whole chip (twice: once per batch group).  A block's 64 weight loads per thread are all issued before the first FMA.  ACC = float for the CDNA kernels, double for the STP regressor (its output steers a bilinear warp
// that amplifies a 1e-6 error in theta to ~3e-5 pixels).
__device__ __forceinline__ float fma_acc(float x, float w, float a) { return fmaf(x, w, a); }
__device__ __forceinline__ double fma_acc(float x, float w, double a) { return fma((double)x, (double)w, a); }

// block = K slice ks x batch group bgrp of a grid of KS slices
template <typename ACC>
__device__ __forceinline__ void skinny_linear_partials_body(const float* __restrict__ x, const float* __restrict__ wt,
                                                            float* __restrict__ partials, int B, int K, int ks, int bgrp, int KS) {
    __shared__ __attribute__((aligned(16))) float xs[LIN_BG * LIN_KS];
    const int b0 = bgrp * LIN_BG, o = threadIdx.x;
    const int k0 = ks * LIN_KS;
    const int nb = min(LIN_BG, B - b0);
    // the weight loads go out first so that their round trip overlaps the x tile's (they do not depend on it)
    float wv[LIN_KS];
#pragma unroll
    for (int k = 0; k < LIN_KS; ++k) {   // index clamped (x is zero there), not predicated: keeps the 64 loads branch-free
        const float w = wt[(size_t)min(k0 + k, K - 1) * 256 + o];
        wv[k] = k0 + k < K ? w : 0.f;
    }
    {   // the x tile: clamped addresses, every load requested before the first LDS store (no predicate around a load)
        constexpr int XIT = LIN_BG * LIN_KS / 256;
        float xv[XIT];
#pragma unroll
        for (int u = 0; u < XIT; ++u) {
            const int i = threadIdx.x + 256 * u, bb = i / LIN_KS, k = i - bb * LIN_KS;
            xv[u] = x[(size_t)(b0 + min(bb, nb - 1)) * K + min(k0 + k, K - 1)];
        }
#pragma unroll
        for (int u = 0; u < XIT; ++u) {
            const int i = threadIdx.x + 256 * u, bb = i / LIN_KS, k = i - bb * LIN_KS;
            xs[i] = (bb < nb && k0 + k < K) ? xv[u] : 0.f;
        }
    }
    __syncthreads();
    ACC acc[LIN_BG];
#pragma unroll
    for (int bb = 0; bb < LIN_BG; ++bb) acc[bb] = (ACC)0;
#pragma unroll
    for (int k = 0; k < LIN_KS; k += 4) {
#pragma unroll
        for (int bb = 0; bb < LIN_BG; ++bb) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + bb * LIN_KS + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[bb] = fma_acc(xv[e], wv[k + e], acc[bb]);
        }
    }
    // [B][KS][256]: a sample's partials are one contiguous run (a [KS][B][256] image made the finisher's loads a 32-KB
    // stride, i.e. one L2 channel)
    for (int bb = 0; bb < nb; ++bb) partials[((size_t)(b0 + bb) * KS + ks) * 256 + o] = (float)acc[bb];
}

// sum of the K-slice partials of output o of sample b, fixed order (bitwise reproducible), 4 independent chains
__device__ __forceinline__ double sum_partials(const float* __restrict__ partials, int B, int KS, int b, int o) {
    // chain c takes slices c, c+4, c+8, ... in increasing order; 32 loads are in flight per round trip
    double a[4] = {0, 0, 0, 0};
    for (int ks0 = 0; ks0 < KS; ks0 += 32) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = partials[((size_t)b * KS + min(ks0 + u, KS - 1)) * 256 + o];   // clamped, not
#pragma unroll                                                       // predicated: a uniform predicate becomes a branch + vmcnt(0) per load
        for (int u = 0; u < 32; ++u) a[u & 3] += ks0 + u < KS ? (double)v[u] : 0.0;
    }
    return (a[0] + a[1]) + (a[2] + a[3]);
}

// The finishers of the motion head as block bodies (256 threads, sample b; v: 256 floats of LDS).  Bodies of cdna_kernels_finish_kernel /
// stp_params_finish_kernel (heads.hip) and of the "rider" blocks a transposed-conv launch carries behind its own tiles (deconv_tile.hip, round 6).
// CDNA (TM:326-329): + bias, relu(k - RELU_SHIFT) + RELU_SHIFT, divide by the 5x5 sum.
__device__ __forceinline__ void cdna_finish_block(const float* __restrict__ partials, const float* __restrict__ bias, float* __restrict__ kerns,
                                                  int B, int KS, int nout, float* __restrict__ vpre, int b, float* v) {
    const int o = threadIdx.x;
    float acc = 0.f;
    if (o < nout) {
        const double a = (double)bias[o] + sum_partials(partials, B, KS, b, o);
        if (vpre) vpre[(size_t)b * 256 + o] = (float)a;
        acc = fmaxf((float)a - 1e-12f, 0.f) + 1e-12f;
    }
    if (o < 256) v[o] = acc;
    __syncthreads();
    if (o < nout) {
        const int g = (o / 25) * 25;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 25; ++i) sum += v[g + i];
        kerns[(size_t)b * nout + o] = acc / sum;
    }
}
// STP (TM:458-468): relu(Linear(100)) -> shared Linear(6) + identity.  w2 reference layout (6,100).
__device__ __forceinline__ void stp_finish_block(const float* __restrict__ partials, const float* __restrict__ b1, const float* __restrict__ w2,
                                                 const float* __restrict__ b2, float* __restrict__ theta, int B, int KS, float* __restrict__ s1_out,
                                                 int b, float* s1) {
    const int o = threadIdx.x;
    if (o < 100) {
        const double acc = (double)b1[o] + sum_partials(partials, B, KS, b, o);
        s1[o] = fmaxf((float)acc, 0.f);
        if (s1_out) s1_out[(size_t)b * 256 + o] = s1[o];
    }
    __syncthreads();
    if (o < 6) {
        double acc = b2[o];
        for (int i = 0; i < 100; ++i) acc = fma((double)w2[o * 100 + i], (double)s1[i], acc);
        theta[b * 6 + o] = (float)(acc + ((o == 0 || o == 4) ? 1.0 : 0.0));
    }
}

}  // namespace pivp
