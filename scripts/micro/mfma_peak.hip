// Ceiling probes for v_mfma_f32_32x32x2_f32 on this box (cdna_hip_programming.md rule 10: measure a known-good
// reference on the same hardware).  Modes: 0 = register-only, 1 = operands re-read from LDS (ds_read_b128) each step.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* clk, int iters, const float* seed) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = seed[i];
    __syncthreads();
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const int lane = threadIdx.x & 63;
    f32x4 a = *reinterpret_cast<const f32x4*>(lds + lane * 36);
    f32x4 b[NACC];
    for (int n = 0; n < NACC; ++n) b[n] = *reinterpret_cast<const f32x4*>(lds + 2304 + (n * 64 + lane) * 4 % 4096);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            a = *reinterpret_cast<const f32x4*>(lds + ((lane + it) & 63) * 36);
            for (int n = 0; n < NACC; ++n) b[n] = *reinterpret_cast<const f32x4*>(lds + 2304 + ((n * 64 + lane + it) & 255) * 4);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[n][s], acc[n], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE, int NACC>
void run(const char* name, int blocks, int iters, float* out, unsigned long long* clk, float* seed) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE, NACC>), dim3(blocks), dim3(256), 0, 0, out, clk, iters, seed);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 2);
        hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
        double flops = (double)blocks * 4 * iters * 4 * NACC * 32 * 32 * 2 * 2;
        double mhz = (double)h[0] / (double)h[1] * 100.0;
        if (rep == 2) printf("%-28s blocks %4d  %.3f ms  %.1f TFLOP/s  in-kernel clock %.0f MHz  cycles/MFMA %.1f\n", name, blocks, ms,
                             flops / ms / 1e9, mhz, (double)h[0] / (iters * 4.0 * NACC));
    }
}

int main() {
    float *out, *seed; unsigned long long* clk;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&clk, 1024 * 16); hipMalloc(&seed, 8192 * 4);
    std::vector<float> hs(8192);
    for (int i = 0; i < 8192; ++i) hs[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.0f - 0.5f;
    hipMemcpy(seed, hs.data(), 8192 * 4, hipMemcpyHostToDevice);
    run<0, 4>("regs only, 4 acc, 1 blk/CU", 256, 20000, out, clk, seed);
    run<0, 4>("regs only, 4 acc, 2 blk/CU", 512, 20000, out, clk, seed);
    run<1, 4>("LDS b128 reads, 4 acc, 1/CU", 256, 20000, out, clk, seed);
    run<1, 4>("LDS b128 reads, 4 acc, 2/CU", 512, 20000, out, clk, seed);
    run<0, 1>("regs only, 1 acc, 1 blk/CU", 256, 80000, out, clk, seed);
    return 0;
}
