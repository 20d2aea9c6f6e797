// What a per-sample hand-off inside ONE launch costs (the ConvLSTM cell's backward chain -- LayerNorm sums -> gate backward -- as one
// launch per cell and timestep needs the 8 blocks of a sample to meet once, through an L2 counter, instead of the whole grid at a kernel boundary).
// 256 blocks x 256 threads = 32 groups of 8 (one block per CU); every round a block writes a partial (256 floats), fences, adds 1 to its group's
// counter and spins until the counter shows that all 8 arrived, then reads the 8 partials.  Also: the same rounds as a dependent chain of launches
// (the kernel boundary it would replace), and the barrier beside a second kernel that fills every CU's second slot (the side stream's place).
// hipcc --offload-arch=gfx950 -O3 scripts/micro/group_barrier.hip -o scripts/micro/group_barrier
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// XCD: groups are the blocks of ONE XCD (linear id mod 8 equal: members g, g + ngroups, ...) and the hand-off uses workgroup-scope release / acquire
// plus loads that bypass the CU's L1 -- coherent through the XCD's own L2, no write-back to memory; otherwise agent scope (any CUs).
template <bool XCD>
__global__ __launch_bounds__(256) void barrier_rounds(float* part, unsigned* counters, float* out, int rounds, int gsize, long long* cyc) {
    const int ngroups = gridDim.x / gsize;
    const int g = XCD ? blockIdx.x % ngroups : blockIdx.x / gsize;
    float acc = 0.f;
    const long long t0 = wall_clock64();
    for (int it = 1; it <= rounds; ++it) {
        part[((size_t)(it & 1) * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = acc + (float)(threadIdx.x + it);
        if (XCD) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); else __threadfence();     // the partial is visible (to the XCD / device-wide) before the arrival is
        __syncthreads();
        if (threadIdx.x == 0) {
            int guard = 0;      // (an exit every wave reaches: a block that is not resident must not hang the others for ever)
            if (XCD) {
                // (the counter stays at agent scope: with workgroup-scope adds and polls the other CUs' arrivals are never seen -- every wait ran into its guard)
                __hip_atomic_fetch_add(&counters[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(&counters[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it * gsize) && ++guard < (1 << 22)) __builtin_amdgcn_s_sleep(1);
            } else {
                __hip_atomic_fetch_add(&counters[g], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(&counters[g], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it * gsize) && ++guard < (1 << 22)) __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        float s = 0.f;
        for (int k = 0; k < gsize; ++k) {                  // the group's partials of this round (written by other CUs: read past this CU's L1)
            const int member = XCD ? g + k * ngroups : g * gsize + k;
            s += __hip_atomic_load(&part[((size_t)(it & 1) * gridDim.x + member) * 256 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        acc = s / (float)gsize;
    }
    const long long t1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(256) void one_round(float* part, float* out, int it, int gsize) {
    const int g = blockIdx.x / gsize;
    float s = 0.f;
    for (int k = 0; k < gsize; ++k) s += part[((size_t)((it - 1) & 1) * gridDim.x + g * gsize + k) * 256 + threadIdx.x];
    part[((size_t)(it & 1) * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = s * 0.125f + (float)(threadIdx.x + it);      // (timing only)
    if (it < 0) out[0] = s;
}
// a kernel that keeps every CU's second block slot busy with MFMAs for a while (the side stream's weight gradient, as a stand-in)
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void busy_mfma(float* out, int iters) {
    f32x16 a = {0};
    float x = threadIdx.x * 0.001f, y = 1.0f;
    for (int i = 0; i < iters; ++i) a = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a, 0, 0, 0);
    out[blockIdx.x * 256 + threadIdx.x] = a[0];
}

int main() {
    const int NB = 256, R = 2000;
    float *part, *out, *out2; unsigned* cnt; long long* cyc;
    CK(hipMalloc(&part, (size_t)2 * NB * 256 * 4)); CK(hipMalloc(&out, NB * 256 * 4)); CK(hipMalloc(&out2, 1024 * 256 * 4));
    CK(hipMalloc(&cnt, 64 * 4)); CK(hipMalloc(&cyc, NB * 8));
    CK(hipMemset(part, 0, (size_t)2 * NB * 256 * 4));
    hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 4; ++variant) {
        const int gsize = variant == 3 ? 256 : variant == 2 ? 16 : 8;
        const bool xcd = variant == 1;
        for (int beside = 0; beside < 2; ++beside) {
            CK(hipMemsetAsync(cnt, 0, 64 * 4, s)); CK(hipStreamSynchronize(s));
            if (beside) hipLaunchKernelGGL(busy_mfma, dim3(256), dim3(256), 0, s2, out2, 400000);     // ~10 ms of MFMAs on every CU
            CK(hipEventRecord(e0, s));
            if (xcd) hipLaunchKernelGGL(barrier_rounds<true>, dim3(NB), dim3(256), 0, s, part, cnt, out, R, gsize, cyc);
            else hipLaunchKernelGGL(barrier_rounds<false>, dim3(NB), dim3(256), 0, s, part, cnt, out, R, gsize, cyc);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<long long> h(NB); CK(hipMemcpy(h.data(), cyc, NB * 8, hipMemcpyDeviceToHost));
            long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
            std::vector<float> ho(NB * 256); CK(hipMemcpy(ho.data(), out, NB * 256 * 4, hipMemcpyDeviceToHost));
            int wrong = 0;
            for (int b = 0; b < NB; ++b) for (int t = 0; t < 256; ++t) { const double e = (double)R * t + 0.5 * R * (R + 1); if (fabs(ho[b * 256 + t] - e) > 1e-3 * e) ++wrong; }
            printf("groups of %3d blocks%s, %s: %.2f us per round (launch / rounds), slowest block %.2f; %d of %d results wrong\n", gsize,
                   xcd ? " of one XCD (workgroup-scope fence, L2-coherent loads)" : "", beside ? "beside an MFMA kernel on every CU" : "alone", ms * 1e3 / R, mx * 0.01 / R,
                   wrong, NB * 256);
        }
    }
    // the kernel boundary the hand-off would replace: the same rounds as a chain of dependent launches
    CK(hipEventRecord(e0, s));
    for (int it = 1; it <= R; ++it) hipLaunchKernelGGL(one_round, dim3(NB), dim3(256), 0, s, part, out, it, 8);
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("the same rounds as %d dependent launches: %.2f us per round\n", R, ms * 1e3 / R);
    return 0;
}
