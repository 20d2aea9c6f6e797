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

// ---- per-device host-side caches -----------------------------------------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize and the CU count belong to a DEVICE, and one process may drive several (Model(device=
// 'cuda:1') after 'cuda:0'): both are remembered per hipGetDevice() index.  These tables and the PIVP_* tuning knobs read once with
// getenv() are the library's only process-global state; they are write-once per device / per process and idempotent, so concurrent
// first calls from several host threads are benign.  Everything else lives in the plan or in the caller's buffers.
constexpr int PIVP_MAX_DEV = 64;
inline int pivp_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PIVP_MAX_DEV) dev = 0;
    return dev;
}
struct PerDeviceOnce { bool done[PIVP_MAX_DEV] = {}; };
// raise a kernel's dynamic-LDS cap once per device; the return code is checked (a launch with the default 64 KB cap would fail later)
inline int pivp_ensure_dyn_lds(PerDeviceOnce& once, const void* kernel, int bytes) {
    const int dev = pivp_current_device();
    if (once.done[dev]) return PIVP_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return PIVP_ERR_LAUNCH;
    once.done[dev] = true;
    return PIVP_OK;
}
inline int pivp_cu_count() {
    static int cus[PIVP_MAX_DEV] = {};
    const int dev = pivp_current_device();
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace pivp {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- two fp16 pieces per fp32 operand (the fp16x3 forms of the ConvLSTM gate conv and of the enc5 / enc6 transposed convs) -----------------------
typedef _Float16 pivp_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pivp_f16x2 __attribute__((ext_vector_type(2)));
typedef float pivp_f32x2 __attribute__((ext_vector_type(2)));
// two fp32 -> packed fp16 (round to nearest even, saturating at the largest finite fp16); a, b become the remainders v - fp16(v), exact in fp32
__device__ __forceinline__ unsigned pivp_pack2h_rest(float& a, float& b) {
    const float ca = __builtin_fminf(__builtin_fmaxf(a, -65504.f), 65504.f), cb = __builtin_fminf(__builtin_fmaxf(b, -65504.f), 65504.f);
    pivp_f32x2 v = {ca, cb};
    const pivp_f16x2 h = __builtin_convertvector(v, pivp_f16x2);
    a -= (float)h[0]; b -= (float)h[1];
    return __builtin_bit_cast(unsigned, h);
}
// the power of two that puts a tensor's largest |value| (the maximum of the 64 partial maxima at tail[2..65]: absmax_partials) into [2^14, 2^15)
__device__ __forceinline__ float pivp_x3_scale_of_max(float m) {
    if (!(m > 0.f) || !(m < 3.0e38f)) return 1.0f;
    int e;
    (void)__builtin_frexpf(m, &e);                  // m = f * 2^e, f in [0.5, 1)
    int k = 15 - e;
    k = k < -60 ? -60 : k > 60 ? 60 : k;
    return __builtin_ldexpf(1.0f, k);
}

// n / d for 0 <= n < 2^31 and a run-time d >= 1 without the division sequence: sh = ceil(log2 d), mul = floor(2^32 (2^sh - d) / d) + 1,
// n / d = (mulhi(mul, n) + n) >> sh  (Granlund & Montgomery's round-up form; d a power of two: mul = 1, the quotient is n >> sh)
inline void pivp_fastdiv(unsigned d, unsigned* mul, unsigned* sh) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    *sh = l;
    *mul = (unsigned)((((1ull << l) - d) << 32) / d + 1);
}
__device__ __forceinline__ int pivp_fdiv(int n, unsigned mul, unsigned sh) { return (int)((__umulhi(mul, (unsigned)n) + (unsigned)n) >> sh); }

// Wave-wide xor butterflies (levels 32, 16, 8, 4, 2, 1; every lane gets the result) without the LDS crossbar (round 6): __shfl_xor is ds_bpermute_b32, six
// dependent LDS round trips per reduction (188 ns; scripts/micro/wave_sum.hip).  The two levels that cross 16-lane rows use gfx950's
// v_permlane32_swap / v_permlane16_swap, the rest DPP: the same pairs in the same order -- bit-identical to the shuffle form -- in 79 ns.
// (The swaps through inline asm: __builtin_amdgcn_permlane32_swap(v, v) with one value for both operands is folded to {v, v} by this compiler.)
__device__ __forceinline__ void wave_swap_halves(float v, bool wide, float& a, float& b) {      // a, b: v and v[lane ^ 32] (wide) / v[lane ^ 16], in either order
    a = v;
    if (wide) asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "=&v"(b));
    else asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "=&v"(b));
}
template <int CTRL, int BANKM = 0xf>
__device__ __forceinline__ float wave_dpp(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, BANKM, false));
}
__device__ __forceinline__ float wave_xor4(float v) {           // v[lane ^ 4]: row_shl:4 into banks 0 and 2, row_shr:4 into banks 1 and 3
    return wave_dpp<0x114, 0xa>(wave_dpp<0x104, 0x5>(v, v), v);
}
__device__ __forceinline__ float wave_sum(float v) {
    float a, b;
    wave_swap_halves(v, true, a, b); v = a + b;
    wave_swap_halves(v, false, a, b); v = a + b;
    v += wave_dpp<0x128>(v, v);          // row_ror:8 = lane ^ 8
    v += wave_xor4(v);
    v += wave_dpp<0x4e>(v, v);           // quad_perm [2,3,0,1]
    v += wave_dpp<0xb1>(v, v);           // quad_perm [1,0,3,2]
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    float a, b;
    wave_swap_halves(v, true, a, b); v = fmaxf(a, b);
    wave_swap_halves(v, false, a, b); v = fmaxf(a, b);
    v = fmaxf(v, wave_dpp<0x128>(v, v));
    v = fmaxf(v, wave_xor4(v));
    v = fmaxf(v, wave_dpp<0x4e>(v, v));
    v = fmaxf(v, wave_dpp<0xb1>(v, v));
    return v;
}

__device__ __forceinline__ float pivp_x3_scale_wave(const float* tail) {       // one partial per lane, xor-tree maximum: every lane returns the scale
    float m = tail[2 + (threadIdx.x & 63)];
    return pivp_x3_scale_of_max(wave_max(m));
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

// Merge the S (count, mean, M2) LayerNorm partials of sample b; call with the 64 lanes of ONE wave.
// Every lane returns the same (mean, 1/sqrt(var + eps)); the summation order is fixed (bitwise reproducible).
// Two weighted sums instead of a tree of pairwise Chan merges (round 6): mean = sum n_i mean_i / sum n_i, then M2 = sum (M2_i + n_i (mean_i - mean)^2) --
// three xor-tree wave sums and ONE division, where the pairwise form paid a division and two dependent shuffles per tree level.  Every consumer of a
// producer's partials runs this in front of its first useful instruction (nine kernels per timestep: the four ln_apply, enc1 / enc2 / enc5 / enc6 with the
// norm folded into their staging, frame_head): 2.0 us of each of them by the stamps of deconv3x3s2_tile_kernel (profiles/r06/NOTES.md 3).
// ln_partial_first: the lane's first partial, requested early (a consumer issues it, then its own first loads, then merges: the partials' round trip and
// the merge's arithmetic run under the consumer's loads instead of in front of them).
__device__ __forceinline__ f32x4 ln_partial_first(const float* __restrict__ partials, int b, int S) {
    const int lane = threadIdx.x & 63;
    return lane < S ? reinterpret_cast<const f32x4*>(partials)[(size_t)b * S + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void ln_merge_partials(f32x4 first, const float* __restrict__ partials, int b, int S, float eps, float& mean_out,
                                                  float& rstd_out) {
    const int lane = threadIdx.x & 63;
    const f32x4* p = reinterpret_cast<const f32x4*>(partials) + (size_t)b * S;
    // the first partial of a lane stays in registers for the second pass (S <= 64 wherever the producer is a tile kernel)
    float cn = first[0], cs = first[0] > 0.f ? first[0] * first[1] : 0.f;
    for (int i = lane + 64; i < S; i += 64) { const f32x4 v = p[i]; cn += v[0]; cs += v[0] > 0.f ? v[0] * v[1] : 0.f; }
    cn = wave_sum(cn); cs = wave_sum(cs);
    const float mean = cs / cn;
    float d = first[1] - mean;
    float q = first[0] > 0.f ? fmaf(first[0] * d, d, first[2]) : 0.f;      // (a slot with count 0 carries nothing: its other fields are not read into the sums)
    for (int i = lane + 64; i < S; i += 64) { const f32x4 v = p[i]; d = v[1] - mean; q += v[0] > 0.f ? fmaf(v[0] * d, d, v[2]) : 0.f; }
    q = wave_sum(q);
    mean_out = mean;
    rstd_out = 1.0f / sqrtf(q / cn + eps);
}
__device__ __forceinline__ void ln_merge_partials(const float* __restrict__ partials, int b, int S, float eps, float& mean_out, float& rstd_out) {
    ln_merge_partials(ln_partial_first(partials, b, S), partials, b, S, eps, mean_out, rstd_out);
}

}  // namespace pivp

// Wave priority of the kernels on the backward sweep's critical path (the caller's stream).  Beside the side stream's weight-gradient
// waves, which keep the matrix pipe of their SIMD busy, a co-resident wave at the default priority 0 gets a VALU issue slot every few
// hundred cycles: a chain of 500 dependent FMAs takes 111 us instead of 3.6 (scripts/contention_probe.py), 45 us with s_setprio 3.
// With priority 3 the sweep's small kernels run 2-4x faster beside the side stream (igemm_small 171 -> 43 us) and the single-GPU train step
// gains 1.5 % (28.21 -> 27.78 ms fp32, 23.99 -> 23.34 three pieces, profiles/r04/NOTES.md); in a data-parallel job the collective's long-lived
// waves, which win the oldest-first arbitration at equal priority, would lose it.  So it is a RUN-TIME switch (round 5): one device word per
// translation unit (the library is built without relocatable device code), written by the unit's setter below on the plan's stream when the
// wanted value changes -- pivp_plan_set_main_priority: on for plans without a gradient listener, off under data parallelism.  The word is per
// device and process, not per plan: two plans that want different values on one device take turns (priority never changes results).
namespace pivp {
__attribute__((unused)) static __device__ int g_main_prio;
}
#define PIVP_SET_MAIN_PRIO() do { if (pivp::g_main_prio) __builtin_amdgcn_s_setprio(3); } while (0)
#define PIVP_DEFINE_MAIN_PRIO_SETTER(unit)                                                                                              \
    namespace pivp {                                                                                                                     \
    int main_prio_set_##unit(int on, hipStream_t s) {                                                                                    \
        static const int kVal[2] = {0, 1};    /* the copy is asynchronous: its source must outlive the call */                           \
        return hipMemcpyToSymbolAsync(HIP_SYMBOL(g_main_prio), &kVal[on ? 1 : 0], sizeof(int), 0, hipMemcpyHostToDevice, s) == hipSuccess \
                   ? PIVP_OK : PIVP_ERR_LAUNCH;                                                                                          \
    }                                                                                                                                    \
    }
