// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the ConvLSTM + CDNA rollout.
// Wavefront = 64 lanes everywhere; no other architecture is targeted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PIVP_OK 0
#define PIVP_ERR_BADARG (-1)
#define PIVP_ERR_LAUNCH (-2)
#define PIVP_ERR_STATE (-3)

#define PIVP_CHECK_ARG(cond) do { if (!(cond)) return PIVP_ERR_BADARG; } while (0)
#define PIVP_LAUNCH_STATUS() (hipGetLastError() == hipSuccess ? PIVP_OK : PIVP_ERR_LAUNCH)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace pivp {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Block-wide sum for blockDim.x == NT (multiple of 64); `red` is LDS scratch of >= NT/64 floats.
// Every thread gets the total.  Two barriers.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

// Chan et al. pairwise combination of (count, mean, M2).
__device__ __forceinline__ void chan_combine(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
    if (nb == 0.f) return;
    const float nt = n + nb;
    const float d = mb - mean;
    const float r = nb / nt;
    mean += d * r;
    m2 += m2b + d * d * n * r;
    n = nt;
}

}  // namespace pivp
