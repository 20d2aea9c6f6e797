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


}  // namespace pivp
