// Wave-wide xor-butterfly sum without the LDS crossbar (round 6): __shfl_xor is ds_bpermute_b32, ~120 cycles of latency per level and six dependent levels
// per sum; the LayerNorm statistics merge in front of nine kernels per timestep runs three such sums.  Same pairs, same order (32, 16, 8, 4, 2, 1), so
// bit-identical: v_permlane32_swap / v_permlane16_swap (gfx950) for the two levels that cross 16-lane rows, DPP for the rest.
// Checks bit-equality against the shuffle form on random data and times a chain of dependent sums of one wave per CU.
//     hipcc --offload-arch=gfx950 -O3 -o scripts/micro/wave_sum scripts/micro/wave_sum.hip && scripts/micro/wave_sum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

template <int CTRL, int BANKM = 0xf>
__device__ __forceinline__ float dpp(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, BANKM, false));
}
__device__ __forceinline__ float sum_shfl(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <int VARIANT>
__device__ __forceinline__ float swap_add(float v, bool wide) {      // v + v[lane ^ 32] (wide) or v + v[lane ^ 16]
    if (VARIANT == 0) {                  // the builtin, both operands the same value
        if (wide) { auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
                    return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]); }
        auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
        return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
    }
    float a = v, b;
    if (wide) asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "=&v"(b));
    else asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "=&v"(b));
    return a + b;
}
template <int VARIANT>
__device__ __forceinline__ float sum_dpp(float v) {
    v = swap_add<VARIANT>(v, true);
    v = swap_add<VARIANT>(v, false);
    v += dpp<0x128>(v, v);               // row_ror:8 = lane ^ 8
    float t = dpp<0x104, 0x5>(v, v);     // row_shl:4 into banks 0, 2 ...
    t = dpp<0x114, 0xa>(t, v);           // ... row_shr:4 into banks 1, 3: lane ^ 4
    v += t;
    v += dpp<0x4e>(v, v);                // quad_perm [2,3,0,1]
    v += dpp<0xb1>(v, v);                // quad_perm [1,0,3,2]
    return v;
}
template <int WHICH>
__global__ void check(const float* x, float* o) {
    const float v = x[blockIdx.x * 64 + threadIdx.x];
    o[blockIdx.x * 64 + threadIdx.x] = WHICH == 0 ? sum_shfl(v) : WHICH == 1 ? sum_dpp<0>(v) : sum_dpp<1>(v);
}
template <int WHICH>
__global__ void chain(float* o, int n) {
    float v = (float)threadIdx.x * 1e-3f;
    for (int i = 0; i < n; ++i) { v = (WHICH == 0 ? sum_shfl(v) : WHICH == 1 ? sum_dpp<0>(v) : sum_dpp<1>(v)) * 1e-2f + (float)threadIdx.x; }
    o[blockIdx.x * 64 + threadIdx.x] = v;
}
int main() {
    const int NB = 4096;
    std::vector<float> h(NB * 64), r[3];
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *x, *o;
    hipMalloc(&x, NB * 64 * 4); hipMalloc(&o, NB * 64 * 4);
    hipMemcpy(x, h.data(), NB * 64 * 4, hipMemcpyHostToDevice);
    for (int w = 0; w < 3; ++w) {
        if (w == 0) check<0><<<NB, 64>>>(x, o); else if (w == 1) check<1><<<NB, 64>>>(x, o); else check<2><<<NB, 64>>>(x, o);
        r[w].resize(NB * 64);
        hipMemcpy(r[w].data(), o, NB * 64 * 4, hipMemcpyDeviceToHost);
    }
    for (int w = 1; w < 3; ++w) printf("variant %d (%s) vs shuffle form: %s\n", w, w == 1 ? "builtin swap" : "asm swap", memcmp(r[0].data(), r[w].data(), NB * 64 * 4) ? "DIFFERENT" : "bit-identical");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 20000;
    for (int w = 0; w < 3; ++w) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (w == 0) chain<0><<<256, 64>>>(o, n); else if (w == 1) chain<1><<<256, 64>>>(o, n); else chain<2><<<256, 64>>>(o, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("chain of %d dependent sums, form %d: %.1f ns per sum\n", n, w, ms * 1e6 / n);
    }
    return 0;
}
