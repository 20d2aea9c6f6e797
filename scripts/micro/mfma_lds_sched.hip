// Which LDS-read placement lets ONE wave per SIMD keep v_mfma_f32_32x32x2_f32 fed?  Per k-group: 5 ds_read_b128
// (1 A + 4 B fragments) feed 16 MFMAs, exactly the ConvLSTM kernel's ratio.
//   MODE 0: reads, then their 16 MFMAs (no prefetch)
//   MODE 1: double-buffered, reads of group g+1 pinned BEFORE the MFMAs of group g (sched_group_barrier)
//   MODE 2: double-buffered, reads of group g+1 interleaved one per MFMA gap at the start of group g
//   MODE 3: as 2, but ds_read_b64 x2 instead of each b128
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* clk, int iters, const float* seed) {
    __shared__ __attribute__((aligned(16))) float lds[9216 * 2];
    for (int i = threadIdx.x; i < 9216 * 2; i += 256) lds[i] = seed[i & 8191];
    __syncthreads();
    f32x16 acc[4];
    for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const float* A = lds + (wave * 32 + l31) * 36 + 4 * half;
    const float* B = lds + 128 * 36 + l31 * 36 + 4 * half;
    f32x4 fa[2], fb[2][4];
    auto rd = [&](int set, int g) {
        const int o = (g & 3) * 8 + ((g >> 2) & 1) * 9216;
        fa[set] = *reinterpret_cast<const f32x4*>(A + o);
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[set][t] = *reinterpret_cast<const f32x4*>(B + t * 32 * 36 + o);
    };
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    rd(0, 0);
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cur = u, nxt = u ^ 1;
            if (MODE == 0) {
                rd(cur, it + u);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s], fb[cur][t][s], acc[t], 0, 0, 0);
            } else {
                rd(nxt, it + u + 1);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s], fb[cur][t][s], acc[t], 0, 0, 0);
                if (MODE == 1) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);   // 5 DS reads
                    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);  // 16 MFMA
                } else {
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 11, 0);
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s + fa[0][0] + fb[0][0][0];
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int blocks, int iters, float* out, unsigned long long* clk, float* seed) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(256), 0, 0, out, clk, iters, seed);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 2);
        hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
        double flops = (double)blocks * 4 * iters * 16 * 32 * 32 * 2 * 2;
        if (rep == 2) printf("%-44s blocks %4d  %.3f ms  %.1f TFLOP/s  clock %.0f MHz  cycles/MFMA %.1f\n", name, blocks, ms,
                             flops / ms / 1e9, (double)h[0] / (double)h[1] * 100.0, (double)h[0] / (iters * 16.0));
    }
}

int main() {
    float *out, *seed; unsigned long long* clk;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&clk, 1024 * 16); hipMalloc(&seed, 8192 * 4);
    std::vector<float> hs(8192);
    for (int i = 0; i < 8192; ++i) hs[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.0f - 0.5f;
    hipMemcpy(seed, hs.data(), 8192 * 4, hipMemcpyHostToDevice);
    run<0>("0 reads then MFMAs (no prefetch)", 256, 20000, out, clk, seed);
    run<1>("1 prefetch g+1, reads pinned before MFMAs", 256, 20000, out, clk, seed);
    run<2>("2 prefetch g+1, 1 read per MFMA gap", 256, 20000, out, clk, seed);
    run<2>("2 same, 2 blocks/CU", 512, 20000, out, clk, seed);
    return 0;
}
