// Shared pieces of the bf16 / split-precision 5x5 kernels (csrc/convlstm_ring.h: weights through an LDS ring; csrc/convlstm_l2direct.h: weights
// from L2 straight into the MFMA operand registers): patch geometry, fragment reads as inline asm, packing helpers, phase stamps.
// Included by convlstm_bf16.hip (the ConvLSTM cell forms) and conv5x5_bf16.hip (the plain 5x5 convolution = the ConvLSTM data gradient).
#pragma once
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "pivp_kernels.h"

namespace pivp {

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int PP = 144;                // patch pixel pitch (bytes): 64 bf16 + 16
constexpr int TH = 8, PH = TH + 4;     // anchor rows per tile, patch rows
constexpr int NPJ = 5;                 // patch pixels per staging thread (512 threads = 64 pixels x 8 pieces per pass)
// (NPJ * 64 = 320 pixels are staged per pass; two 8x8 images need 2 * 12 * 12 = 288, one 8x16 tile 12 * 20 = 240)
// Patch ROW pitch.  A 16-lane phase of the A fragment's ds_read_b128 covers pixels of TWO (8 x 16 tile) or FOUR (8 x 8 tiles)
// patch rows; with rows simply 20 / 12 pixels apart (2880 / 1728 B) half of its lanes landed on the banks of the other half
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.47, profiles/r03: every fragment read took two passes, and four multiplying waves
// then keep the LDS busy for as long as their MFMAs run).  The 144-B pixel pitch puts 16 consecutive pixels on 16 distinct 16-B
// bank slots; a row pitch that is a multiple of 256 B continues that sequence into the next row (8 x 16 tile: pixels 16..31 of a
// 32-row M tile), one that is 128 (mod 256) gives the complementary slots to alternate rows (8 x 8 tiles: 8 pixels per row).
constexpr int RP16 = 3072;             // 20 px * 144 B = 2880 -> 12 * 256
constexpr int RP8 = 1920;              // 12 px * 144 B = 1728 -> 7 * 256 + 128
constexpr int PATCH_BYTES = 2 * PH * RP8;   // 46,080 (= 320 * 144 as before); one 8 x 16 tile: 12 * 3072 = 36,864
static_assert(PH * RP16 <= PATCH_BYTES && RP16 % 256 == 0 && RP8 % 256 == 128 && RP16 >= 21 * PP && RP8 >= 13 * PP, "patch rows");
template <int PL> constexpr int patch_plane_bytes() { return PATCH_BYTES; }      // bytes of one patch plane
// swizzle of a weight row's eight 16-B pieces (ring slot = [row][64 bf16] = 128-B rows, so two consecutive rows span the 64 banks):
// the 16 lanes of a ds_read_b128 phase read one piece each from 16 different rows and are conflict-free iff the 8 even and the 8
// odd rows among them all use different pieces.  Rows of a phase: 32 consecutive MFMA columns are ring rows r0 + {0-3, 12-15, 20-27}
// or r0 + {4-11, 16-19, 28-31} when a wave's columns are consecutive rows or 16-row runs 32 rows apart (plain conv; 32-channel
// ConvLSTM blocks): (row >> 1) & 7 separates them.  16-channel ConvLSTM blocks take 8-row runs of the four gates (16 rows apart):
// ((row >> 1) & 3) | (gate >> 1) << 2.  (The first version used row & 7: two passes per read, same counter.)
template <int NCH, bool LSTM>
__device__ __forceinline__ int ring_swizzle(int row) {
    if constexpr (LSTM && NCH == 16) return ((row >> 1) & 3) | (((row >> 5) & 1) << 2);
    else return (row >> 1) & 7;
}
// Weight ring: DEP taps of LDS-DMA prefetch in NSL = DEP + 1 slots.  Round 3 ran 3 taps ahead and measured the tap loop at 38 GB/s of weight
// stream per CU, 800 cycles per tap for 512 of MFMA; a DMA takes ~1.1 us from issue to landing, so three 16-KB taps in flight ARE 38-43 GB/s
// (Little's law), not the CU's fill rate (the guide's ring GEMM takes in 68 GB/s with 84 KB in flight).  Round 4: as many slots as the
// 160 KB of LDS hold beside the patch -- 7 of 16 KB for 32-channel blocks (6 taps = 96 KB in flight), 8 of 8 KB for 16-channel ones;
// the split mode's two patch planes leave the old 4 (16-channel blocks) or 2 (32-channel blocks, LATE schedule).
// MEASURED (one box, full rebuilds, profiles/r04/NOTES.md): the deep ring is SLOWER -- seven layers at B = 32 188.7 us against 179.9 with 3 taps
// ahead, at B = 256 994.8 against 970.3, bf16 rollout 2.97 against 2.88 ms, train step 11.93 against 11.72 -- so the in-flight depth was not
// what held the stream at 38 GB/s per CU; the depth stays 3 (the deep ring is in the history).
template <int NCH, int PL>
constexpr int ring_depth() {
    if (PL == 2 && NCH == 32) return 1;                                          // LATE schedule
    const int fit = (160 * 1024 - PL * patch_plane_bytes<PL>()) / (PL * 4 * NCH * 128) - 1;     // slots that fit, minus one = taps ahead
    return fit > 3 ? 3 : fit;
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

__device__ __forceinline__ float b_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float b_tanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

__device__ __forceinline__ unsigned pack2(float a, float b) {   // two fp32 -> packed bf16, round to nearest even
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// LDS reads and their waits as inline asm (see the kernel: the compiler must not see them as LDS accesses)
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read_b128(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// the in/out operands tie the fragments to the wait so that no MFMA that consumes them moves above it
__device__ __forceinline__ void wait_lgkm(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& e) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(e));
}
__device__ __forceinline__ void wait_lgkm(bf16x8& a, bf16x8& b, bf16x8& c) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c));
}
__device__ __forceinline__ void wait_lgkm(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& e, bf16x8& f, bf16x8& g) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(e), "+v"(f), "+v"(g));
}
__device__ __forceinline__ void wait_lgkm(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& e, bf16x8& f, bf16x8& g, bf16x8& h, bf16x8& i) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i));
}
__device__ __forceinline__ void wait_lgkm(bf16x8& a, bf16x8& b) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
}
typedef pivp_f16x8 f16x8;                  // (two fp16 pieces per operand: pivp_pack2h_rest, pivp_x3_scale_of_max in pivp_common.h)
template <class F, size_t... I> __device__ __forceinline__ void static_for_impl(F&& f, std::index_sequence<I...>) { (f(std::integral_constant<int, (int)I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_index_sequence<N>{}); }
}  // namespace

inline bool bf16_geometry_ok(const IgemmDesc& d) {
    if (d.ksize != 5 || d.pad != 2 || d.in_step != 1 || d.Hin % TH) return false;
    if (d.c0 % 8 || d.c1 % 8 || d.ld0 % 4 || d.ld1 % 4) return false;
    if (d.Win % 16 == 0) return true;
    return d.Win % 8 == 0 && d.B % 2 == 0;
}

#ifdef PIVP_BF16_STAMPS   // in-kernel phase stamps of every block's wave 0, constant-rate 100 MHz counter (scripts/bf16_stamps.py)
__device__ long long pivp_bf16_stamps[2048 * 8];
// entries 6, 7: the shader-cycle counter (s_memtime) at stamps 2 and 3: cycles / wall time = the clock the chip holds inside the tap loop
#define BF_STAMP(i) do { if (tid == 0 && blockIdx.x < 2048 && blockIdx.y == 0) { pivp_bf16_stamps[blockIdx.x * 8 + (i)] = (long long)wall_clock64(); \
    if ((i) == 2 || (i) == 3) pivp_bf16_stamps[blockIdx.x * 8 + 4 + (i)] = (long long)clock64(); } } while (0)
#else
#define BF_STAMP(i)
#endif

}  // namespace pivp
