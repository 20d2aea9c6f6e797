// ConvLSTM gate convolution with bf16 operands and fp32 accumulation (BASELINE.json config 3, "bf16"): the opt-in
// reduced-precision form of igemm_lstm (igemm_f32.hip).  Reference op: BasicConvLSTMCell.__call__, TM:234-276.
//
// With v_mfma_f32_32x32x16_bf16 a 32x32 tile of a K = 32 chunk costs 64 MFMA cycles instead of the 1024 of the fp32
// instruction, so the gather scheme of igemm_f32.hip (a fresh A tile per tap) would be bound by its staging, not by the
// matrix pipe.  This kernel therefore keeps the block's INPUT PATCH resident: a block owns 128 anchors of one image
// (8 x 16 pixels; on 8-wide maps 8 x 8 pixels of two images) and 4 gates x NCH channels.  For every group of 64 input
// channels of concat(x, h) it stages the patch with its 2-pixel halo ONCE (fp32 NHWC in HBM -> bf16 in LDS, pixel pitch
// 144 B so that the 16-lane phases of a ds_read_b128 hit distinct banks; out-of-image pixels and channels past Cin are
// the hardware zeros of an out-of-range buffer load) and runs the 25 taps against it: a tap only shifts the LDS address
// of the A fragment.  The weights ([group][tap][4C][64] bf16, packed once per rollout by pack_lstm_bf16) stream through a
// 4-slot LDS ring filled by global_load_lds_dwordx4 three taps ahead of their use: no VGPRs, no ds_writes, and the
// loads stay in flight across the per-tap barrier (raw s_barrier + counted vmcnt).  The DMAs are issued by four LOADER
// waves (waves 4..7 of the 8-wave block) that do nothing else in the tap loop; waves 0..3 multiply (2 x 2 over the
// 128 x 4 NCH block tile) and issue no VMEM instruction there.  A ring slot is lane-linear (the DMA
// writes base + lane * 16), so bank conflicts of the B fragment reads are removed by an XOR swizzle of the 16-B pieces
// of a 128-B weight row, applied to the per-lane SOURCE address of the DMA and to the read address alike.
// Accumulators, gate math, cell state, h and the LayerNorm partial stay fp32; the epilogue is that of igemm_f32.hip.
// The same kernel with a plain epilogue (LSTM = false) is a general 5x5 stride-1 convolution: the ConvLSTM data gradient.
#include <stdlib.h>
#include <type_traits>

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
// bytes of one patch plane: the three-piece mode (PL = 3) serves 16-wide tiles only, whose 12 rows need 36,864 B: three planes then leave
// room for two 24-KB ring slots (16-channel blocks), which the two-image 8 x 8 layout's 46,080 would not
template <int PL> constexpr int patch_plane_bytes() { return PL == 3 ? PH * RP16 : PATCH_BYTES; }
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
// what held the stream at 38 GB/s per CU; the default stays 3.  The general ring (any depth, counted vmcnt tail) stays, tested at both depths.
#ifndef PIVP_BF16_DEPTH
#define PIVP_BF16_DEPTH 3              // taps ahead at most; 0 = the deepest ring that fits
#endif
template <int NCH, int PL>
constexpr int ring_depth() {
    if ((PL == 2 && NCH == 32) || PL == 3) return 1;                             // LATE schedule (PL = 3: not used, its ring is per k-step)
    const int fit = (160 * 1024 - PL * patch_plane_bytes<PL>()) / (PL * 4 * NCH * 128) - 1;     // slots that fit, minus one = taps ahead
    const int cap = fit > 7 ? 7 : fit;
    return (PIVP_BF16_DEPTH > 0 && PIVP_BF16_DEPTH < cap) ? PIVP_BF16_DEPTH : cap;
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
// fp16 pieces: the weights are packed times a power of two chosen per tensor so that the largest lands in [2^14, 2^15) -- the second piece of any
// weight down to 2^-18 of the largest is then a normal fp16 number.  The pack's tail (256 2-byte elements behind the fragments) holds the scale
// (float 0) and the 64 partial maxima it was taken from (floats 2..65).
// 64 blocks (one partial per lane of the consumers' wave) x 512 threads with EIGHT 16-byte loads in flight per thread: the data gradients of the fp16x3 mode
// call it once per cell and timestep on a dG of up to 17 MB, beside the side stream's weight gradients (the first form, 256 threads and one load per trip,
// took 32 us per launch there: 2.2 ms of a train step)
__global__ __launch_bounds__(512) void absmax_partials_kernel(const float* __restrict__ w, long n, float* __restrict__ tail) {
    float m = 0.f;
    const f32x4* w4 = reinterpret_cast<const f32x4*>(w);               // (n is a multiple of 32: K-inner packed weights, NHWC tensors of >= 8 channels)
    const long n4 = n >> 2, stride = 64L * 512;
    long i = (long)blockIdx.x * 512 + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = w4[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v[u][0]), __builtin_fabsf(v[u][1]))), __builtin_fmaxf(__builtin_fabsf(v[u][2]), __builtin_fabsf(v[u][3])));
    }
    for (; i < n4; i += stride) {
        const f32x4 v = w4[i];
        m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1]))), __builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float red[8];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = red[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) r = __builtin_fmaxf(r, red[k]);
        tail[2 + blockIdx.x] = r;
    }
}
__device__ __forceinline__ float x3_scale_of(const float* tail) {      // every caller computes the same power of two from the 64 partial maxima
    float m = 0.f;
    for (int i = 0; i < 64; ++i) m = __builtin_fmaxf(m, tail[2 + i]);
    return pivp_x3_scale_of_max(m);
}
__device__ __forceinline__ void wait_lgkm(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& e, bf16x8& f, bf16x8& g, bf16x8& h, bf16x8& i, bf16x8& j) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i), "+v"(j));
}
}  // namespace

// fp32 K-inner packed weights [25][wcin/32][N][32] -> bf16 [ceil(wcin/64)][25][PL][Np][64], channels past wcin and rows past N zero.
// PL = 1: plane 0 = bf16(w).  PL = 2 (split mode): plane 0 = hi = bf16(w), plane 1 = lo = bf16(w - hi).  PL = 3 (three pieces): plane 1 =
// mid = bf16(w - hi), plane 2 = lo = bf16((w - hi) - mid); both differences are exact in fp32, so hi + mid + lo = w.
// fp16 = 1 (PL = 2): the two planes are FP16 pieces of s w, s = the tensor's power-of-two scale from the partial maxima absmax_partials_kernel left
// in the pack's tail (the ring kernel's form of the fp16x3 mode: the layers on 8-wide maps)
__global__ void pack_lstm_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wb, int wcin, int N, int Np, int PL, long total, int fp16) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    float wscale = 1.0f;
    if (fp16) {
        __shared__ float sc;
        if (threadIdx.x == 0) {
            sc = x3_scale_of(reinterpret_cast<const float*>(wb + total));
            if (blockIdx.x == 0) reinterpret_cast<float*>(wb + total)[0] = sc;
        }
        __syncthreads();
        wscale = sc;
    }
    if (i >= total) return;
    const int c64 = (int)(i & 63);
    long r = i >> 6;
    const int n = (int)(r % Np); r /= Np;
    const int pl = (int)(r % PL); r /= PL;
    const int tap = (int)(r % 25);
    const int cg = (int)(r / 25);
    const int ch = cg * 64 + c64;
    float v = 0.f;
    if (ch < wcin && n < N) v = w[(((long)tap * (wcin >> 5) + (ch >> 5)) * N + n) * 32 + (ch & 31)];
    if (fp16) {
        v = __builtin_fminf(__builtin_fmaxf(v * wscale, -65504.f), 65504.f);
        _Float16 hh = (_Float16)v;
        if (pl == 1) { v -= (float)hh; hh = (_Float16)v; }
        wb[i] = __builtin_bit_cast(unsigned short, hh);
        return;
    }
    __bf16 h = (__bf16)v;
    if (pl >= 1) { v -= (float)h; h = (__bf16)v; }
    if (pl == 2) { v -= (float)h; h = (__bf16)v; }
    wb[i] = __builtin_bit_cast(unsigned short, h);
}

// PL = 3 pack, FRAGMENT-MAJOR: [group][tap][k-step][plane][8-channel group c8][lane][8 bf16] -- one B fragment of
// v_mfma_f32_32x32x16_bf16 whose 32 columns are the four gates of eight channels (lane (half, l31): gate l31 / 8, channel c8 * 8 + l31 % 8;
// k = the k-step's channels half * 8 .. + 8) is 1 KB in lane order: one global_load_lds_dwordx4 of a wave moves exactly one fragment into
// a lane-linear (conflict-free) kilobyte of the ring, one global_load_dwordx4 of a wave loads it straight into the MFMA's operand registers.
#ifndef PIVP_X3_SHARE_B
#define PIVP_X3_SHARE_B 0   // 1: the eight-wave two-fp16-piece kernels load every B fragment ONCE per block (one wave each, a tap ahead) and hand it to the waves
#endif                      // that share it through two LDS slots of a tap's fragments, one block barrier per tap, instead of every wave loading its own copy
                            // from L2.  Built because the timing-only variant WITHOUT the B loads runs 26-31 % faster (the MFMA-bound time).  Correct (tests,
                            // 60 bit-identical rollouts) and SLOWER: 389.3 against 360.3 us per seven layers with the barrier per tap, 388.3 against 348.8
                            // with one per k-step (the first form): neither the duplicated L2 traffic nor the barrier count is what the loads cost
#ifndef PIVP_X6_READS_FIRST
#define PIVP_X6_READS_FIRST 0      // 1: the fp16 kernels issue all A reads of the next k-step in front of a k-step's MFMAs (measured: 349.0 against 347.7 us: no change)
#endif
#ifndef PIVP_X3_RD8
#define PIVP_X3_RD8 0       // 1: the 4 x 2 two-fp16-piece kernel keeps EIGHT k-steps (two taps) of B fragments in flight instead of four (the counters show 30 % of
                            // its wave cycles in s_waitcnt, almost none of it on LDS): measured 159.4 against 147.2 us for lstm3 / 4 / 6: slower, off
#endif
#ifndef PIVP_X3_DOUBLE
#define PIVP_X3_DOUBLE 0    // 1: the eight-wave two-fp16-piece kernels take TWO k-steps (32 channels) per wait.  Built on the guess that their short k-step
#endif                      // (3 MT MFMAs) pays a per-wait cost twice as often as the three-piece form; measured: 349.0 against 346.2 us per seven layers: off
template <class F, size_t... I> __device__ __forceinline__ void static_for_impl(F&& f, std::index_sequence<I...>) { (f(std::integral_constant<int, (int)I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_index_sequence<N>{}); }
#ifndef PIVP_X6_MIDLOAD
#define PIVP_X6_MIDLOAD 0   // 1: the eight-wave L2-direct kernels issue their fragment loads in the middle of a k-step's MFMAs instead of behind them
                            // (measured: three pieces 466.4 against 465.9 us per six layers, two fp16 pieces 308.8 against 298.6: off)
#endif
#ifndef PIVP_X6_ABL
#define PIVP_X6_ABL 0       // timing-only ablations of the three-piece kernel (results are then wrong): 1 no per-k-step barriers, 2 no B fragment
#endif                      // reads, 4 no reads of the A mid / lo planes, 8 no weight DMAs, 16 one MFMA per product instead of six
constexpr int X6_CHUNK = 3 * 2 * 1024;      // ring slot of the 16-channel-block kernel: one k-step = 3 planes x 2 wave columns x 1 KB
// plain = 1: the same pack for a plain 5x5 convolution (the data gradient): fragment c8 = the 32 consecutive output columns c8 * 32 .. + 32 of the
// Np padded ones (rows past N zero)
// pieces = 2: TWO FP16 pieces of s w (hi = fp16, lo = fp16(s w - hi)), s = the tensor's power-of-two scale (absmax_partials_kernel ran before)
// One thread = the 8 consecutive input channels of one (fragment, lane) -- 32 contiguous bytes of the fp32 pack -- and ALL the pieces' planes of
// them: two 16-B loads, `pieces` 16-B stores, one index decode per 8 * pieces output elements (the first version decoded per element: 20 us per
// layer and rollout, 3 % of an fp16x3 rollout).  nthreads = lstm_bf16_weight_elems / 8.
__global__ void pack_lstm_x6_kernel(const float* __restrict__ w, unsigned short* __restrict__ wb, int wcin, int N, int Np, int plain, int pieces, long nthreads) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = nthreads * 8 * pieces;
    float wscale = 1.0f;
    if (pieces == 2) {
        __shared__ float sc;
        if (threadIdx.x == 0) {
            sc = x3_scale_of(reinterpret_cast<const float*>(wb + total));
            if (blockIdx.x == 0) reinterpret_cast<float*>(wb + total)[0] = sc;      // what the convolution's epilogue divides by
        }
        __syncthreads();
        wscale = sc;
    }
    if (j >= nthreads) return;
    const int lane = (int)(j & 63);
    long r = j >> 6;
    const int n8 = Np / 32, C = N / 4;
    const int c8 = (int)(r % n8); r /= n8;
    const int ks = (int)(r & 3); r >>= 2;
    const int tap = (int)(r % 25);
    const int cg = (int)(r / 25);
    const int half = lane >> 5, l31 = lane & 31;
    const int n = plain ? c8 * 32 + l31 : (l31 >> 3) * C + c8 * 8 + (l31 & 7);
    const int ch = cg * 64 + ks * 16 + half * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ch < wcin && n < N) {
        const f32x4* src = reinterpret_cast<const f32x4*>(w + (((long)tap * (wcin >> 5) + (ch >> 5)) * N + n) * 32 + (ch & 31));
        const f32x4 a = src[0], b = src[1];
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
    const long frag0 = (((long)(cg * 25 + tap) * 4 + ks) * pieces) * n8 + c8;       // plane pl: + pl * n8
    unsigned short* dst = wb + frag0 * 512 + lane * 8;
    if (pieces == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= wscale;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            uint4 o;
            o.x = pivp_pack2h_rest(v[0], v[1]); o.y = pivp_pack2h_rest(v[2], v[3]); o.z = pivp_pack2h_rest(v[4], v[5]); o.w = pivp_pack2h_rest(v[6], v[7]);
            *reinterpret_cast<uint4*>(dst + (long)pl * n8 * 512) = o;
        }
        return;
    }
    for (int pl = 0; pl < pieces; ++pl) {
        uint4 o;
        unsigned* ow = reinterpret_cast<unsigned*>(&o);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned p2 = pack2(v[2 * q], v[2 * q + 1]);
            ow[q] = p2;
            v[2 * q] -= __builtin_bit_cast(float, p2 << 16); v[2 * q + 1] -= __builtin_bit_cast(float, p2 & 0xffff0000u);
        }
        *reinterpret_cast<uint4*>(dst + (long)pl * n8 * 512) = o;
    }
}

#ifdef PIVP_BF16_STAMPS   // in-kernel phase stamps of every block's wave 0, constant-rate 100 MHz counter (scripts/bf16_stamps.py)
__device__ long long pivp_bf16_stamps[2048 * 8];
// entries 6, 7: the shader-cycle counter (s_memtime) at stamps 2 and 3: cycles / wall time = the clock the chip holds inside the tap loop
#define BF_STAMP(i) do { if (tid == 0 && blockIdx.x < 2048 && blockIdx.y == 0) { pivp_bf16_stamps[blockIdx.x * 8 + (i)] = (long long)wall_clock64(); \
    if ((i) == 2 || (i) == 3) pivp_bf16_stamps[blockIdx.x * 8 + 4 + (i)] = (long long)clock64(); } } while (0)
#else
#define BF_STAMP(i)
#endif

// LSTM = true: the ConvLSTM cell (block columns = 4 gates x NCH channels, gate epilogue).  LSTM = false: a plain 5x5 stride-1 "same"
// convolution out[m][n] (+)= sum x[m + tap][k] w[tap][k][n] with block columns = 4 NCH consecutive n (the ConvLSTM DATA gradient: x = dG,
// w = the flipped transposed weights); gridDim.y splits the channel groups, partial sums then meet in `out` by atomic adds.
// PL = 2: split mode.  Every fp32 operand travels as TWO bf16 numbers, hi = bf16(v) and lo = bf16(v - hi) (two patch planes, two weight
// planes per ring slot), and a product a * b is formed as a_lo * b_hi + a_hi * b_lo + a_hi * b_hi on three MFMAs (each exact in fp32):
// 16 bits of product mantissa instead of 8, 3e-5 instead of 2e-2 per-pixel on the config 1 rollout (scripts/split_bf16_study.py).
// PL = 3: THREE pieces per operand (hi + mid + lo = v exactly) and the six products whose weight is >= 2^-16 of the leading one:
// hi*hi into the main accumulator, lo*hi + hi*lo + mid*mid + mid*hi + hi*mid into a second one that joins it in front of the epilogue (the
// main accumulator then rounds once per 16 exact products, the corrections' own rounding is 2^-8 of that): fp32-grade gate pre-activations on
// the bf16 matrix cores, six MFMAs per product = a 417 TFLOP/s ceiling (the fp32 MFMA's is 157).  16-wide tiles, 16-channel blocks, two ring
// slots' worth of LDS, organised as EIGHT 6-KB slots of one k-step each (KST schedule below): with whole taps in two slots the loop ran at
// the latency of one LDS-DMA per tap (1.2 us for 0.64 of MFMA); a barrier per k-step frees a slot four times per tap, so the DMAs run seven
// k-steps = 1.1 us of MFMA time ahead in the same 48 KB.
// F16 (PL = 2, LSTM): the two planes are FP16 pieces (the weights times the power of two in the pack's tail, the sums scaled back): the fp16x3
// mode's cell on maps this kernel's 8 x 8 tiles serve and the L2-direct kernel's 16-wide ones do not.
template <int NCH, bool LSTM, int PL = 1, bool F16 = false>
__global__ __launch_bounds__(512, 1) void convlstm_bf16_kernel(const IgemmDesc d, const unsigned short* __restrict__ wb, int tw, int ncols) {
    static_assert(!F16 || (PL == 2 && NCH == 16), "fp16 pieces: the split form, 16-channel / 64-column blocks");
    // (F16 without LSTM: the data gradient on 8-wide maps; the activations -- gradients -- are staged times d.wscale_part's power of two, as in convlstm_x6g_kernel)
    constexpr int BN = 4 * NCH;                 // block columns: [gate][channel]
    constexpr int PLANE = BN * 128;             // one weight plane of a ring slot: BN rows x 64 bf16
    constexpr int SLOT = PL * PLANE;            // bytes of one ring slot
    // Split mode with 32-channel blocks: two patch planes (92 KB) leave room for TWO 32 KB ring slots only, so the schedule changes: the
    // loaders bring tap it + 1 in during tap it (one tap of lookahead), and the block barrier sits at the END of a tap (LATE).
    constexpr bool LATE = PL == 2 && NCH == 32;
    constexpr bool KST = PL == 3;                   // ring of eight one-k-step slots, a block barrier per k-step
    constexpr int PB = patch_plane_bytes<PL>();     // bytes of one patch plane
    static_assert(PL != 3 || NCH == 16, "three pieces: 16-channel blocks only");
    constexpr int DEP = ring_depth<NCH, PL>();      // taps of weight prefetch
    constexpr int NSL = DEP + 1;                    // ring slots
    static_assert(PL * PB + (KST ? 8 * X6_CHUNK : NSL * PL * BN * 128) <= 160 * 1024 && DEP >= 1, "the ring must fit beside the patch");
    constexpr int G = BN / 32;                  // global_load_lds per loader thread and tap
    constexpr int TPW = NCH / 16;               // MFMA column tiles per wave
    constexpr int CPW = NCH / 2;                // channels per wave (all 4 gates of a channel stay in one wave)
    constexpr int GPT = 32 / CPW;               // gates per MFMA tile
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // patch | ring
    unsigned char* const patch = lds;                  // patch plane(s) | ring
    const int tid = threadIdx.x, lane = tid & 63;
    // waves 0..3 multiply (2 x 2 over the 128 x BN block tile), waves 4..7 only feed the weight ring: a global_load_lds costs
    // 60-180 cycles of its wave's issue time, which in a multiplying wave is time the matrix pipe idles (4 per tap: a quarter
    // of the tap); issued by a second wave of the same SIMD they overlap the MFMAs.
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave8 >= 4;
    const int wave = wave8 & 3;
    const int wm = wave & 1, wn = wave >> 1;
    const int half = lane >> 5, l31 = lane & 31;
    const int C = d.C, N = d.N;                       // (plain conv: N = rows of the padded weight pack, a multiple of BN)
    const int n_nblk = LSTM ? C / NCH : N / BN;
    const int H = d.Hin, W = d.Win;
    // tile geometry: tw = 16: one image, 8 x 16 anchors; tw = 8: two images, 8 x 8 anchors each
    const int ti_n = tw == 16 ? 1 : 2;
    const int PW = tw + 4;
    const int RP = tw == 16 ? RP16 : RP8;             // patch row pitch (bytes)
    const int npix = ti_n * PH * PW;
    const int tpr = W / tw, tpi = (H / TH) * tpr;          // tiles per row / per image
    const int n_tiles = (d.B / ti_n) * tpi;
    int lid = blockIdx.x;
    if ((gridDim.x & 7) == 0) lid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-aware, column-block major
    const int nblk = lid / n_tiles, tile = lid - nblk * n_tiles;
    const int b0 = (tile / tpi) * ti_n, trem = tile - (tile / tpi) * tpi;
    const int y0 = (trem / tpr) * TH, x0 = (trem - (trem / tpr) * tpr) * tw;
    BF_STAMP(0);
    const int c0 = d.c0, ld0 = d.ld0, ld1 = d.ld1;
    const int cin = c0 + d.c1;
    const int ncg_all = (cin + 63) >> 6;
    const int cgbase = (int)blockIdx.y * ncg_all / (int)gridDim.y;                 // this block's channel groups: [cgbase, cgbase + ncg)
    const int ncg = ((int)blockIdx.y + 1) * ncg_all / (int)gridDim.y - cgbase;
    const int nchunks = 25 * ncg;

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x0), 0, d.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.c1 ? d.x1 : d.x0), 0, d.c1 ? d.bytes1 : d.bytes0, 0x00020000);
    constexpr unsigned OOB = 0xC0000000u;

    // ---- patch staging (all 8 waves): thread = (pixel (tid >> 3) + 64 j, 8-channel piece tid & 7), j < 5 -------------------
    const int cpiece = tid & 7;
    int a_pix[NPJ];                                    // global pixel index, or -1 outside the image / past the patch
    int a_lds[NPJ];                                    // the pixel's byte offset in the patch (pixels past the patch: row 0's padding)
#pragma unroll
    for (int j = 0; j < NPJ; ++j) {
        const int p = (tid >> 3) + 64 * j;
        const int ti = p / (PH * PW), pr = p - ti * (PH * PW);
        const int py = pr / PW, px = pr - py * PW;
        const int iy = y0 - 2 + py, ix = x0 - 2 + px;
        const bool ok = p < npix && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        a_pix[j] = ok ? ((b0 + ti) * H + iy) * W + ix : -1;
        a_lds[j] = p < npix ? (ti * PH + py) * RP + px * PP : PW * PP;
    }
    f32x4 plo[NPJ], phi[NPJ];                          // a patch in flight (live only between the two halves of a staging)
    auto patch_load = [&](int cg) {
        const int ch = (cgbase + cg) * 64 + cpiece * 8;   // first of this thread's 8 channels of concat(x, h)
        const bool s0 = ch < c0, s1 = !s0 && ch < cin;
        const int ld = s0 ? ld0 : ld1, co = s0 ? ch : ch - c0;
#pragma unroll
        for (int j = 0; j < NPJ; ++j) {
            const unsigned off = (a_pix[j] >= 0 && (s0 || s1)) ? (unsigned)((a_pix[j] * ld + co) * 4) : OOB;
            if (s0) {
                plo[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0));
                phi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 16, 0));
            } else {
                plo[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0));
                phi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 16, 0));
            }
        }
    };
    float a_scale = 1.0f;
    if constexpr (F16 && !LSTM) a_scale = pivp_x3_scale_wave(d.wscale_part);
    auto patch_store = [&]() {
#pragma unroll
        for (int j = 0; j < NPJ; ++j) {
            // unconditional (pixels past npix write their zeros into the padding behind row 0, which nobody reads): a predicated write
            // leaves the loads "pending" on the skipped path for hipcc's wait-count pass, which then drains vmcnt inside the tap loop
            if constexpr (F16) {
                float r[8] = {plo[j][0], plo[j][1], plo[j][2], plo[j][3], phi[j][0], phi[j][1], phi[j][2], phi[j][3]};
                if constexpr (!LSTM) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] *= a_scale;
                }
                uint4 hh, ll;
                hh.x = pivp_pack2h_rest(r[0], r[1]); hh.y = pivp_pack2h_rest(r[2], r[3]); hh.z = pivp_pack2h_rest(r[4], r[5]); hh.w = pivp_pack2h_rest(r[6], r[7]);
                ll.x = pivp_pack2h_rest(r[0], r[1]); ll.y = pivp_pack2h_rest(r[2], r[3]); ll.z = pivp_pack2h_rest(r[4], r[5]); ll.w = pivp_pack2h_rest(r[6], r[7]);
                *reinterpret_cast<uint4*>(patch + a_lds[j] + cpiece * 16) = hh;
                *reinterpret_cast<uint4*>(patch + PB + a_lds[j] + cpiece * 16) = ll;
                continue;
            }
            uint4 v;
            v.x = pack2(plo[j][0], plo[j][1]); v.y = pack2(plo[j][2], plo[j][3]);
            v.z = pack2(phi[j][0], phi[j][1]); v.w = pack2(phi[j][2], phi[j][3]);
            *reinterpret_cast<uint4*>(patch + a_lds[j] + cpiece * 16) = v;
            if constexpr (PL >= 2) {                   // second plane: bf16(v - hi); hi as a float is its 16 bits shifted up
                float r[8] = {plo[j][0], plo[j][1], plo[j][2], plo[j][3], phi[j][0], phi[j][1], phi[j][2], phi[j][3]};
                auto rest = [&](unsigned p2, int i) {      // r[i], r[i + 1] become the remainders (exact in fp32); returns them as bf16
                    r[i] -= __builtin_bit_cast(float, p2 << 16); r[i + 1] -= __builtin_bit_cast(float, p2 & 0xffff0000u);
                    return pack2(r[i], r[i + 1]);
                };
                uint4 l;
                l.x = rest(v.x, 0); l.y = rest(v.y, 2); l.z = rest(v.z, 4); l.w = rest(v.w, 6);
                *reinterpret_cast<uint4*>(patch + PB + a_lds[j] + cpiece * 16) = l;
                if constexpr (PL == 3) {               // third plane: bf16((v - hi) - mid)
                    uint4 q;
                    q.x = rest(l.x, 0); q.y = rest(l.y, 2); q.z = rest(l.z, 4); q.w = rest(l.w, 6);
                    *reinterpret_cast<uint4*>(patch + 2 * PB + a_lds[j] + cpiece * 16) = q;
                }
            }
        }
    };

    // Every block walks the 25 taps of a channel group in its own rotation (tap0, tap0 + 1, ... mod 25): blocks that run in
    // step would otherwise all pull the same 16 KB of weights out of the same few L2 channels at the same time.
    const int tap0 = (lid * 7) % 25;

    // =========================================================================================================================
    // loader waves: DMA j of a tap writes ring bytes [(j * 256 + lt) * 16, +16): row (j * 32 + lt / 8), piece lt % 8, which holds
    // SOURCE piece (lt % 8) ^ (row % 8) of that row (lt = thread index within the four loader waves)
    // =========================================================================================================================
    if (loader) {
        const int lt = tid - 256;
        unsigned char* const ring = lds + PL * PB;
        const unsigned char* wsrc[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int row = j * 32 + (lt >> 3), g = row / NCH, cl = row - g * NCH;
            const int piece = (lt & 7) ^ ring_swizzle<NCH, LSTM>(row);
            const int grow = LSTM ? g * C + nblk * NCH + cl : nblk * BN + row;
            wsrc[j] = reinterpret_cast<const unsigned char*>(wb) + (size_t)grow * 128 + piece * 16;
        }
        const size_t wstep = (size_t)PL * N * 128;     // bytes between consecutive (group, tap) weight tiles ([PL][N][64] bf16 each)
        int i_slot = 0, i_tap = tap0, i_cg = 0;        // the ring slot of the next tap to issue, and which tap that is
        auto issue_weights = [&]() {
            const int slot = i_slot;
            i_slot = i_slot + 1 == NSL ? 0 : i_slot + 1;
            const size_t goff = (size_t)((cgbase + i_cg) * 25 + i_tap) * wstep;
            i_tap = i_tap == 24 ? 0 : i_tap + 1;
            i_cg += i_tap == tap0 ? 1 : 0;
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    unsigned char* dst = ring + slot * SLOT + pl * PLANE + (j * 256 + wave * 64) * 16;   // wave-uniform; the DMA adds lane * 16
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[j] + goff + (size_t)pl * N * 128),
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                }
        };
        if constexpr (KST) {
            // chunk q = (tap index it = q / 4 in this block's rotation, k-step q % 4) lives in slot q % 8.  Loader wave 4 + p moves plane p (two
            // fragments, wn = 0 / 1, per chunk); the fourth loader wave only stages the patch and keeps the barriers.
            const bool mover = wave < 3;
            const size_t pls = (size_t)(N / 32) * 1024;                 // bytes between the planes of a k-step in the pack (N: its rows, 4 C for the cell)
            const unsigned char* const wbase = reinterpret_cast<const unsigned char*>(wb) + (size_t)wave * pls + (size_t)(nblk * 2) * 1024 + (size_t)lane * 16;
            const int NQ = nchunks * 4;
            int i_q = 0, i_tap = tap0, i_cg = 0;
            auto issue_chunk = [&]() {
                const int slot = i_q & 7, ks = i_q & 3;
                const size_t goff = (size_t)(((cgbase + i_cg) * 25 + i_tap) * 4 + ks) * 3 * pls;
                ++i_q;
                if (ks == 3) { i_tap = i_tap == 24 ? 0 : i_tap + 1; i_cg += i_tap == tap0 ? 1 : 0; }
                if (mover && !(PIVP_X6_ABL & 8)) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        unsigned char* dst = ring + slot * X6_CHUNK + wave * 2048 + j * 1024;      // wave-uniform; the DMA adds lane * 16
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + goff + j * 1024),
                                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                    }
                }
            };
            issue_chunk(); issue_chunk(); issue_chunk();           // chunks 0..2 in front of the patch's loads
            patch_load(0);
            patch_store();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            issue_chunk(); issue_chunk(); issue_chunk(); issue_chunk(); issue_chunk();     // chunks 3..7 (all eight slots): in flight across the barrier
            __builtin_amdgcn_s_barrier();                          // the patch and chunk 0 are published
            int tap = tap0, cg = 0;
            for (int q = 0; q < NQ; ++q) {
                // barrier q publishes chunk q + 1 (issued so far: chunks up to q + 7; the `newer` ones behind q + 1 may stay in flight, 2 DMAs each)
                int newer = NQ - q - 2;
                newer = newer < 0 ? 0 : newer > 6 ? 6 : newer;
                if (newer >= 6) wait_vmcnt<12>();
                else if (newer == 5) wait_vmcnt<10>();
                else if (newer == 4) wait_vmcnt<8>();
                else if (newer == 3) wait_vmcnt<6>();
                else if (newer == 2) wait_vmcnt<4>();
                else if (newer == 1) wait_vmcnt<2>();
                else wait_vmcnt<0>();
                if (!(PIVP_X6_ABL & 1)) __builtin_amdgcn_s_barrier();
                if (q + 8 < NQ) issue_chunk();                     // every multiplying wave holds chunk q in registers: its slot takes chunk q + 8
                if ((q & 3) == 3) {
                    tap = tap == 24 ? 0 : tap + 1;
                    if (tap == tap0 && ++cg < ncg) {               // next 64 input channels: all 8 waves restage the patch
                        __syncthreads();
                        patch_load(cg);
                        patch_store();
                        __syncthreads();
                    }
                }
            }
            return;
        }
        if constexpr (LATE) {
            issue_weights();                                   // tap 0
            patch_load(0);
            patch_store();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // tap 0 and the patch are published
            int tap = tap0, cg = 0;
            for (int it = 0; it < nchunks; ++it) {
                if (it + 1 < nchunks) issue_weights();         // tap it + 1 into the slot tap it - 1 left at the last barrier
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                  // end of tap it: tap it + 1 is published
                tap = tap == 24 ? 0 : tap + 1;
                if (tap == tap0 && ++cg < ncg) {               // next 64 input channels: all 8 waves restage the patch
                    patch_load(cg);
                    patch_store();
                    __syncthreads();
                }
            }
            return;
        }
        constexpr int DEP0 = DEP < 3 ? DEP : 3;      // taps requested in front of the patch (the patch's loads return behind them)
#pragma unroll
        for (int i = 0; i < DEP0; ++i)
            if (i < nchunks) issue_weights();
        patch_load(0);
        patch_store();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // taps 0..2 and this thread's part of the patch are in LDS
#pragma unroll
        for (int i = DEP0; i < DEP; ++i)             // the rest of the ring: in flight across the barrier
            if (i < nchunks) issue_weights();
        __builtin_amdgcn_s_barrier();
        int tap = tap0, cg = 0;
        for (int it = 0; it < nchunks; ++it) {
            // the multiplying waves' mid-tap barrier publishes tap it + 1: this wave's share of it must have landed.  Issued so far:
            // taps up to it + DEP - 1; the `newer` ones behind tap it + 1 may stay in flight (G * PL DMAs per thread and tap).
            {
                int newer = nchunks - it - 2;
                newer = newer < 0 ? 0 : newer > DEP - 2 ? DEP - 2 : newer;
                constexpr int Q = G * PL;
                if (newer <= 0) wait_vmcnt<0>();
                else if (newer == 1) wait_vmcnt<Q>();
                else if (newer == 2) wait_vmcnt<2 * Q>();
                else if (newer == 3) wait_vmcnt<3 * Q>();
                else if (newer == 4) wait_vmcnt<4 * Q>();
                else wait_vmcnt<5 * Q>();
                static_assert(DEP - 2 <= 5 && 5 * Q <= 63, "vmcnt immediates");
            }
            __builtin_amdgcn_s_barrier();
            // every multiplying wave is past tap it - 1: its ring slot takes tap it + DEP
            if (it + DEP < nchunks) issue_weights();
            tap = tap == 24 ? 0 : tap + 1;
            if (tap == tap0 && ++cg < ncg) {           // next 64 input channels: all 8 waves restage the patch
                __syncthreads();
                patch_load(cg);
                patch_store();
                __syncthreads();
            }
        }
        return;                                        // (the last iteration drained this wave's DMAs)
    }

    // =========================================================================================================================
    // multiplying waves
    // =========================================================================================================================
    f32x16 acc[2][TPW];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][t][r] = 0.f;

    // fragment addresses (bytes).  A: row l31 of M tile mt = anchor 64 wm + 32 mt + l31; k piece `half` of the k-step
    int a_off[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int i = 64 * wm + 32 * mt + l31;
        const int ti = tw == 16 ? 0 : i >> 6, ay = tw == 16 ? i >> 4 : (i >> 3) & 7, ax = tw == 16 ? i & 15 : i & 7;
        a_off[mt] = (ti * PH + ay) * RP + ax * PP + half * 16;
    }
    // B: MFMA column l31 of tile t = gate t * GPT + l31 / CPW, channel wn * CPW + l31 % CPW; ring row = gate * NCH + channel
    int b_row[TPW], b_sw[4];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
        b_row[t] = (LSTM ? (t * GPT + l31 / CPW) * NCH + wn * CPW + (l31 % CPW) : (wn * TPW + t) * 32 + l31) * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)      // (the tiles of a wave are 32 or 64 ring rows apart: one swizzle value serves them all)
        b_sw[ks] = ((2 * ks + half) ^ ring_swizzle<NCH, LSTM>(b_row[0] >> 7)) * 16;

    // LDS reads go through inline asm: hipcc knows that an LDS-DMA writes LDS and puts s_waitcnt vmcnt(0) in front of every
    // ds_read it can see.  The waits below are explicit instead.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    bf16x8 fa[2][2], fb[2][TPW];                       // [register set][tile]
    bf16x8 fal[2][2], fbl[2][TPW];                     // ... and their second planes (split modes)
    bf16x8 fa3[2][2], fb3[2][TPW];                     // ... and their third planes (PL = 3)
    f32x16 accl[2][TPW];                               // PL = 3: the five correction products' accumulator; F16: the two cross terms'
    if constexpr (PL == 3 || F16) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) accl[mt][t][r] = 0.f;
    }
    auto wait_frags = [&](auto SET) {
        constexpr int st = decltype(SET)::value;
        if constexpr (PL == 3) wait_lgkm(fa[st][0], fa[st][1], fb[st][0], fal[st][0], fal[st][1], fbl[st][0], fa3[st][0], fa3[st][1], fb3[st][0]);
        else if constexpr (PL == 2 && TPW == 2) wait_lgkm(fa[st][0], fa[st][1], fb[st][0], fb[st][1], fal[st][0], fal[st][1], fbl[st][0], fbl[st][1]);
        else if constexpr (PL == 2) wait_lgkm(fa[st][0], fa[st][1], fb[st][0], fal[st][0], fal[st][1], fbl[st][0]);
        else if constexpr (TPW == 2) wait_lgkm(fa[st][0], fa[st][1], fb[st][0], fb[st][1]);
        else wait_lgkm(fa[st][0], fa[st][1], fb[st][0]);
    };
    auto read_frags = [&](auto SET, auto KS, int tp, int slot) {   // fragments of k-step KS of tap tp (weights in ring slot `slot`)
        constexpr int st = decltype(SET)::value, ks = decltype(KS)::value;
        const int ty = tp / 5, tx = tp - ty * 5;
        const unsigned ab = lds0 + ty * RP + tx * PP;
        const unsigned bb = lds0 + PL * PB + slot * SLOT + b_sw[ks];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) fa[st][mt] = lds_read_b128<ks * 32>(ab + a_off[mt]);
        if constexpr (KST) {      // `slot` = 0 / 4: the tap's first slot; k-step ks is slot + ks, a plane 2 KB, this wave's fragment wn, lane-linear
            const unsigned kb = lds0 + PL * PB + slot * X6_CHUNK + wn * 1024 + lane * 16;
            if (!(PIVP_X6_ABL & 2)) {
                fb[st][0] = lds_read_b128<ks * X6_CHUNK>(kb);
                fbl[st][0] = lds_read_b128<ks * X6_CHUNK + 2048>(kb);
                fb3[st][0] = lds_read_b128<ks * X6_CHUNK + 4096>(kb);
            }
            if (!(PIVP_X6_ABL & 4)) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    fal[st][mt] = lds_read_b128<ks * 32>(ab + PB + a_off[mt]);
                    fa3[st][mt] = lds_read_b128<ks * 32>(ab + 2 * PB + a_off[mt]);
                }
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) fb[st][t] = lds_read_b128<0>(bb + b_row[t]);
        if constexpr (PL >= 2) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) fal[st][mt] = lds_read_b128<ks * 32>(ab + PB + a_off[mt]);
#pragma unroll
            for (int t = 0; t < TPW; ++t) fbl[st][t] = lds_read_b128<0>(bb + PLANE + b_row[t]);
        }
        if constexpr (PL == 3) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) fa3[st][mt] = lds_read_b128<ks * 32>(ab + 2 * PB + a_off[mt]);
#pragma unroll
            for (int t = 0; t < TPW; ++t) fb3[st][t] = lds_read_b128<0>(bb + 2 * PLANE + b_row[t]);
        }
    };
    auto mfmas = [&](auto SET) {
        constexpr int st = decltype(SET)::value;
        if constexpr (PL == 3) {       // term-major, so that consecutive MFMAs write different accumulators; corrections smallest first
#pragma unroll
            for (int term = (PIVP_X6_ABL & 16) ? 5 : 0; term < 6; ++term)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        if (term == 0) accl[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa3[st][mt], fb[st][t], accl[mt][t], 0, 0, 0);        // lo * hi
                        else if (term == 1) accl[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][mt], fb3[st][t], accl[mt][t], 0, 0, 0);   // hi * lo
                        else if (term == 2) accl[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[st][mt], fbl[st][t], accl[mt][t], 0, 0, 0);  // mid * mid
                        else if (term == 3) accl[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[st][mt], fb[st][t], accl[mt][t], 0, 0, 0);   // mid * hi
                        else if (term == 4) accl[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][mt], fbl[st][t], accl[mt][t], 0, 0, 0);   // hi * mid
                        else acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][mt], fb[st][t], acc[mt][t], 0, 0, 0);                      // hi * hi
                    }
            return;
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                if constexpr (F16) {
                    auto h = [](const bf16x8& v) { return __builtin_bit_cast(pivp_f16x8, v); };
                    // (the cross terms on their own accumulator: on ONE the three roundings per k-step over K = 4800 measured 1.56 x the fp32
                    // kernel's rms error, with two it is at the L2-direct form's 0.95 x)
                    accl[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fal[st][mt]), h(fb[st][t]), accl[mt][t], 0, 0, 0);
                    accl[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fa[st][mt]), h(fbl[st][t]), accl[mt][t], 0, 0, 0);
                    acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fa[st][mt]), h(fb[st][t]), acc[mt][t], 0, 0, 0);
                    continue;
                }
                if constexpr (PL == 2) {               // the two cross terms first, the leading term last
                    acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[st][mt], fb[st][t], acc[mt][t], 0, 0, 0);
                    acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][mt], fbl[st][t], acc[mt][t], 0, 0, 0);
                }
                acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][mt], fb[st][t], acc[mt][t], 0, 0, 0);
            }
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;

    // ---- prologue: this thread's part of the patch, the epilogue's operands ---------------------------------------------------
    // bias and c_{t-1} are requested here: read in the epilogue they cost one exposed HBM round trip per accumulator row
    // (16-32 in a row), more than the whole tap loop.  Lane (grp, channel) owns accumulator rows r with r % GPT == grp.
    patch_load(0);
    const int chl = wn * CPW + (l31 % CPW);
    const int ch = nblk * NCH + chl;
    const int grp = l31 / CPW;
    constexpr int OWN = 16 / GPT;
    float bj = 0.f, bi = 0.f, bf = 0.f, bo = 0.f;
    float cpre[2][OWN];
    if constexpr (LSTM) {
        bj = d.bias[ch]; bi = d.bias[C + ch]; bf = d.bias[2 * C + ch] + 1.0f; bo = d.bias[3 * C + ch];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int k = 0; k < OWN; ++k) {
                const int r = k * GPT + grp;
                const int i = 64 * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int ti = tw == 16 ? 0 : i >> 6, ay = tw == 16 ? i >> 4 : (i >> 3) & 7, ax = tw == 16 ? i & 15 : i & 7;
                const int m = ((b0 + ti) * H + y0 + ay) * W + x0 + ax;
                cpre[mt][k] = d.cstate_in[(size_t)m * C + ch];
            }
    }
    patch_store();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    BF_STAMP(1);
    __builtin_amdgcn_s_barrier();                      // patch and taps 0..2 are in LDS
    BF_STAMP(2);
    if constexpr (KST) {
        // a block barrier per k-step: barrier q (behind the wait for chunk q's fragments, so its slot is free) publishes chunk q + 1, whose
        // fragments are requested right behind it and fly during chunk q's twelve MFMAs
        int tap = tap0, cg = 0;
        auto kbar = [&]() { if (!(PIVP_X6_ABL & 1)) __builtin_amdgcn_s_barrier(); };
        auto one_mfma = [&](auto CUR, auto I) {
            constexpr int st = decltype(CUR)::value, i = decltype(I)::value, term = i >> 1, mt = i & 1;
            if constexpr ((PIVP_X6_ABL & 16) && term != 5) return;
            if constexpr (term == 0) accl[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa3[st][mt], fb[st][0], accl[mt][0], 0, 0, 0);        // lo * hi
            else if constexpr (term == 1) accl[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][mt], fb3[st][0], accl[mt][0], 0, 0, 0);   // hi * lo
            else if constexpr (term == 2) accl[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[st][mt], fbl[st][0], accl[mt][0], 0, 0, 0);  // mid * mid
            else if constexpr (term == 3) accl[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[st][mt], fb[st][0], accl[mt][0], 0, 0, 0);   // mid * hi
            else if constexpr (term == 4) accl[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][mt], fbl[st][0], accl[mt][0], 0, 0, 0);   // hi * mid
            else acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][mt], fb[st][0], acc[mt][0], 0, 0, 0);                                // hi * hi
        };
        auto one_read = [&](auto NXT, auto KS, auto I, unsigned ab, unsigned kb) {
            constexpr int st = decltype(NXT)::value, ks = decltype(KS)::value, i = decltype(I)::value;
            if constexpr (i < 3) {                         // B: plane i of the k-step's slot (published by the barrier just passed)
                if constexpr (PIVP_X6_ABL & 2) return;
                bf16x8 v = lds_read_b128<ks * X6_CHUNK + i * 2048>(kb);
                if constexpr (i == 0) fb[st][0] = v; else if constexpr (i == 1) fbl[st][0] = v; else fb3[st][0] = v;
            } else {                                       // A: plane (i - 3) / 2, M tile (i - 3) % 2
                constexpr int pl = (i - 3) >> 1, mt = (i - 3) & 1;
                if constexpr ((PIVP_X6_ABL & 4) && pl > 0) return;
                bf16x8 v = lds_read_b128<ks * 32>(ab + pl * PB + a_off[mt]);
                if constexpr (pl == 0) fa[st][mt] = v; else if constexpr (pl == 1) fal[st][mt] = v; else fa3[st][mt] = v;
            }
        };
        // A k-step of register set CUR (its fragments have landed): four MFMAs with the six A reads of the next k-step behind them (the patch does
        // not depend on the barrier), the block barrier -- behind queued MFMAs, so the matrix pipe keeps working while the waves meet --, four MFMAs
        // with the three B reads the barrier has just published, four more MFMAs.  (Measured, profiles/r04/NOTES.md 9: this order, reads one per
        // MFMA gap, and barrier + all reads in front of the MFMAs run within 2 % of each other: ~610 cycles per k-step at the 2.03 GHz the chip holds
        // here, for 12 MFMAs of 37 = 445; what is left is one wave per SIMD's exposed waits, which the 32-channel form below hides with a second wave.)
        auto kstep = [&](auto CUR, auto NXT, auto KS, int tp, int slot, auto RD) {
            constexpr bool rd = decltype(RD)::value;
            const int ty = tp / 5, tx = tp - ty * 5;
            const unsigned ab = lds0 + ty * RP + tx * PP;
            const unsigned kb = lds0 + PL * PB + slot * X6_CHUNK + wn * 1024 + lane * 16;
#define PIVP_X6_M(I) one_mfma(CUR, std::integral_constant<int, I>{});
#define PIVP_X6_R(I) if constexpr (rd) one_read(NXT, KS, std::integral_constant<int, I>{}, ab, kb);
#define PIVP_X6_S __builtin_amdgcn_sched_barrier(0);
            PIVP_X6_M(0) PIVP_X6_R(3) PIVP_X6_R(4) PIVP_X6_S
            PIVP_X6_M(1) PIVP_X6_R(5) PIVP_X6_R(6) PIVP_X6_S
            PIVP_X6_M(2) PIVP_X6_R(7) PIVP_X6_S
            PIVP_X6_M(3) PIVP_X6_R(8) PIVP_X6_S
            kbar();
            PIVP_X6_S
            PIVP_X6_M(4) PIVP_X6_R(0) PIVP_X6_S
            PIVP_X6_M(5) PIVP_X6_R(1) PIVP_X6_S
            PIVP_X6_M(6) PIVP_X6_R(2) PIVP_X6_S
            PIVP_X6_M(7) PIVP_X6_M(8) PIVP_X6_M(9) PIVP_X6_M(10) PIVP_X6_M(11)
            PIVP_X6_S
#undef PIVP_X6_M
#undef PIVP_X6_R
#undef PIVP_X6_S
        };
        read_frags(S0{}, K0{}, tap, 0);
        for (int it = 0; it < nchunks; ++it) {
            const int sb = (it & 1) * 4;
            wait_frags(S0{}); kstep(S0{}, S1{}, K1{}, tap, sb, std::true_type{});
            wait_frags(S1{}); kstep(S1{}, S0{}, K2{}, tap, sb, std::true_type{});
            wait_frags(S0{}); kstep(S0{}, S1{}, K3{}, tap, sb, std::true_type{});
            tap = tap == 24 ? 0 : tap + 1;
            const bool regroup = tap == tap0;
            wait_frags(S1{});
            if (!regroup) kstep(S1{}, S0{}, K0{}, tap, sb ^ 4, std::true_type{});
            else kstep(S1{}, S0{}, K0{}, tap, sb ^ 4, std::false_type{});
            if (regroup && ++cg < ncg) {
                __syncthreads();
                patch_load(cg);
                patch_store();
                __syncthreads();
                read_frags(S0{}, K0{}, tap, sb ^ 4);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (LATE) {
        // end-of-tap barrier schedule: the first fragments of a tap are requested right behind the barrier that published it
        int tap = tap0, cg = 0;
        for (int it = 0; it < nchunks; ++it) {
            const int slot = it & 1;
            read_frags(S0{}, K0{}, tap, slot);
            wait_frags(S0{}); read_frags(S1{}, K1{}, tap, slot); mfmas(S0{});
            __builtin_amdgcn_sched_barrier(0);
            wait_frags(S1{}); read_frags(S0{}, K2{}, tap, slot); mfmas(S1{});
            __builtin_amdgcn_sched_barrier(0);
            wait_frags(S0{}); read_frags(S1{}, K3{}, tap, slot); mfmas(S0{});
            __builtin_amdgcn_sched_barrier(0);
            wait_frags(S1{}); mfmas(S1{});
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            tap = tap == 24 ? 0 : tap + 1;
            if (tap == tap0 && ++cg < ncg) {
                patch_load(cg);
                patch_store();
                __syncthreads();
            }
        }
    } else {
    read_frags(S0{}, K0{}, tap0, 0);

    // One tap = 4 k-steps of 16 channels; fragments of the next k-step are requested before the MFMAs of the current one.  The
    // barrier that publishes the NEXT tap's weights sits in the middle of the tap (its skew hides behind queued MFMAs), so the
    // first fragments of the next tap can be requested right after the last k-step.  No VMEM instruction in this loop.
    int tap = tap0, cg = 0, slot = 0;
    for (int it = 0; it < nchunks; ++it) {
        const int nslot = slot + 1 == NSL ? 0 : slot + 1;
        wait_frags(S0{}); read_frags(S1{}, K1{}, tap, slot); mfmas(S0{});
        __builtin_amdgcn_sched_barrier(0);
        wait_frags(S1{}); read_frags(S0{}, K2{}, tap, slot); mfmas(S1{});
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        wait_frags(S0{}); read_frags(S1{}, K3{}, tap, slot); mfmas(S0{});
        __builtin_amdgcn_sched_barrier(0);
        tap = tap == 24 ? 0 : tap + 1;
        const bool regroup = tap == tap0;
        wait_frags(S1{});
        if (!regroup) read_frags(S0{}, K0{}, tap, nslot);
        mfmas(S1{});
        if (regroup && ++cg < ncg) {                   // next 64 input channels: all 8 waves restage the patch
            __syncthreads();                           // every wave is done with the old patch
            patch_load(cg);
            patch_store();
            __syncthreads();
            read_frags(S0{}, K0{}, tap, nslot);
        }
        slot = nslot;
        __builtin_amdgcn_sched_barrier(0);
    }
    }   // !LATE
    if constexpr (PL == 3) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][t][r] += accl[mt][t][r];
    }
    if constexpr (F16) {           // the weights were packed times a power of two (the pack's tail, behind its [groups][25][2][N][64] elements)
        float inv = 1.0f / *reinterpret_cast<const float*>(wb + (size_t)((d.wcin + 63) >> 6) * 25 * 2 * N * 64);      // (d.wcin: the pack's channels, also at t = 0)
        if constexpr (!LSTM) inv *= 1.0f / a_scale;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][t][r] = (acc[mt][t][r] + accl[mt][t][r]) * inv;
    }

    if constexpr (!LSTM) {
        // ---- plain epilogue: accumulator row = anchor, column = output channel; 32 lanes write 128 contiguous bytes ----------
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = 64 * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int ti = tw == 16 ? 0 : i >> 6, ay = tw == 16 ? i >> 4 : (i >> 3) & 7, ax = tw == 16 ? i & 15 : i & 7;
                const size_t m = (size_t)(((b0 + ti) * H + y0 + ay) * W + x0 + ax);
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const int col = nblk * BN + (wn * TPW + t) * 32 + l31;
                    if (col < ncols) {                  // the pack's rows past the real column count are zero padding
                        float* o = d.out + m * d.ldo + col;
                        if (gridDim.y > 1) atomicAdd(o, acc[mt][t][r]);
                        else if (d.accum) *o += acc[mt][t][r];
                        else *o = acc[mt][t][r];
                    }
                }
            }
        return;
    }
    BF_STAMP(3);
    // ---- epilogue: gates, state update, optional gate activations and LayerNorm partial ----------------------------------
    // Accumulator row r of a lane is one anchor; its column is (gate t * GPT + grp, channel): the 4 gates of an (anchor, channel)
    // sit in the GPT lanes lane ^ (x * CPW) and the TPW tiles.  Lane grp takes rows r = k * GPT + grp: it keeps its own gate of
    // that row and receives the others from its partners in GPT - 1 xor-shuffles per tile, each partner sending the row its
    // receiver owns.  Every lane then updates one (anchor, channel) per k: no idle lanes, 2 (NCH 32) or 3 (NCH 16) shuffles per
    // cell instead of 8 or 16.  Register arrays are only indexed statically; per-lane choices are select chains.
    auto pick = [&](const float (&v)[GPT], int idx) -> float {
        if constexpr (GPT == 2) {
            return idx ? v[1] : v[0];
        } else {
            const float lo = (idx & 1) ? v[1] : v[0], hi = (idx & 1) ? v[3] : v[2];
            return (idx & 2) ? hi : lo;
        }
    };
    float sv[2][OWN];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int k = 0; k < OWN; ++k) {
            float g4[4];
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                float rows[GPT], val[GPT];
#pragma unroll
                for (int g = 0; g < GPT; ++g) rows[g] = acc[mt][t][k * GPT + g];
                val[0] = pick(rows, grp);
#pragma unroll
                for (int x = 1; x < GPT; ++x) val[x] = __shfl_xor(pick(rows, grp ^ x), x * CPW, 64);
#pragma unroll
                for (int g = 0; g < GPT; ++g) g4[t * GPT + g] = pick(val, g ^ grp);
            }
            const int r = k * GPT + grp;
            const int i = 64 * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half;       // anchor within the block
            const int ti = tw == 16 ? 0 : i >> 6, ay = tw == 16 ? i >> 4 : (i >> 3) & 7, ax = tw == 16 ? i & 15 : i & 7;
            const int m = ((b0 + ti) * H + y0 + ay) * W + x0 + ax;
            const size_t o = (size_t)m * C + ch;
            const float aj = b_tanh(g4[0] + bj), ai = b_sigmoid(g4[1] + bi);
            const float af = b_sigmoid(g4[2] + bf), ao = b_sigmoid(g4[3] + bo);
            const float cn = cpre[mt][k] * af + ai * aj;
            d.cstate_out[o] = cn;
            const float hn = b_tanh(cn) * ao;
            d.hout[o] = hn;
            sv[mt][k] = hn;
            if (d.gates_out) {
                float* gp = d.gates_out + (size_t)m * 4 * C + ch;
                gp[0] = aj; gp[C] = ai; gp[2 * C] = af; gp[3 * C] = ao;
            }
        }
    BF_STAMP(4);
#ifdef PIVP_BF16_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the block's stores have left
    BF_STAMP(5);
#endif
    if (d.ln_part) {
        // (count, mean, M2) of the h values of each image of the tile; with two images wave pair wm owns image wm.
        float* red = reinterpret_cast<float*>(lds);
        float s1 = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int k = 0; k < OWN; ++k) s1 += sv[mt][k];
        s1 = wave_sum(s1);
        const float c1 = 64.f * 2 * OWN;
        __syncthreads();
        if (lane == 0) { red[wave] = s1; red[4 + wave] = c1; }
        __syncthreads();
        float cnt, mean;
        if (ti_n == 1) {
            cnt = (red[4] + red[5]) + (red[6] + red[7]);
            mean = ((red[0] + red[1]) + (red[2] + red[3])) / cnt;
        } else {
            cnt = red[4 + wm] + red[6 + wm];
            mean = (red[wm] + red[2 + wm]) / cnt;
        }
        float q = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int k = 0; k < OWN; ++k) { const float dd = sv[mt][k] - mean; q = fmaf(dd, dd, q); }
        q = wave_sum(q);
        if (lane == 0) red[8 + wave] = q;
        __syncthreads();
        if (ti_n == 1) {
            if (tid == 0) {
                float* p = d.ln_part + ((size_t)b0 * d.ln_nparts + (size_t)trem * n_nblk + nblk) * 4;
                p[0] = cnt; p[1] = mean; p[2] = (red[8] + red[9]) + (red[10] + red[11]); p[3] = 0.f;
            }
        } else if (lane == 0 && wn == 0) {
            float* p = d.ln_part + ((size_t)(b0 + wm) * d.ln_nparts + (size_t)trem * n_nblk + nblk) * 4;
            p[0] = cnt; p[1] = mean; p[2] = red[8 + wm] + red[10 + wm]; p[3] = 0.f;
        }
    }
}

// =================================================================================================================================
// Three-piece ConvLSTM with the weights read STRAIGHT FROM L2 into the MFMA's operand registers: no weight ring, no loader waves, no block
// barrier in the tap loop.  The ring form above (16-channel blocks) spends a fifth of its k-step on the LDS-DMAs and the per-k-step barrier,
// and its 16-channel blocks need two rounds on 32 x 32 maps.  Here a block is 128 anchors x 32 channels x 4 gates and all eight waves
// multiply (2 x 4: wave tile = 64 anchors x the four gates of 8 channels, two waves per SIMD); the LDS holds only the three patch planes.
// A wave's B fragment of a (tap, k-step, plane) is 1 KB of the fragment-major pack: one coalesced global_load_dwordx4 per wave, requested
// FOUR k-steps (one tap) ahead into a register ring of 4 x 3 fragments: with two waves per SIMD a wave's k-step lasts ~0.45 us, so a tap of
// lookahead covers an L2 round trip of 1-2 us; the two waves that share a fragment (wm = 0 / 1) ask for the same lines at about the same time.  Same arithmetic, term for term, as the ring form.
// =================================================================================================================================
// NWN = 2: the same with four waves (2 x 2), 16 channels per block, for layers whose 32-channel blocks would leave CUs idle: one wave per
// SIMD (up to 512 registers), so the fragment ring is EIGHT k-steps (two taps) deep -- a lone wave's k-step lasts ~0.22 us.
// NWM x NWN waves over the 128 anchors x (8 NWN channels x 4 gates) of a block: 2 x 4 (32 channels), 4 x 2 (16 channels, still two waves per
// SIMD: a wave's tile is 32 anchors), 2 x 2 (16 channels, four waves).
// LSTM = false: the plain 5x5 convolution with the same loop (the data gradient): a wave's 32 columns are consecutive output columns, the channel
// groups may be split over gridDim.y (partial sums then meet in `out` by atomic adds), the epilogue stores / adds the accumulators.
// PCS = 2: TWO FP16 pieces per operand instead of three bf16 ones (22 bits of operand mantissa; the weights arrive times 2^8 and the sum is scaled
// back) and three MFMAs per product: hi*hi on the main accumulator, lo*hi + hi*lo on the second.  scripts/split_fp16_study.py: the truncation is a
// quarter of the fp32 path's own error.  Forward gate convolutions only (gradients are too small for fp16's exponent range).
// IN_LN: the x operand (d.x0, c0 <= 64 channels) is a RAW ConvLSTM output whose LayerNorm (per-element gamma / beta d.in_g / d.in_b [H W][c0], statistics
// merged from the producer's partials d.in_part) is applied while the patch is staged -- the expression of ln_apply_kernel; out-of-image pixels load 0
// for v, gamma and beta alike and stay 0.  Inference rollouts: hidden1 -> lstm2 and hidden3 -> lstm4 lose their ln_apply launch.
// THX = 16 (two fp16 pieces, 4 x 2 waves): tiles of 16 x 16 anchors -- 256 anchors x 16 channels per block, a wave tile of 64 anchors.  The weight bytes a
// block pulls from L2 per multiply-add halve (the 8-row forms pull ~6.0-6.6 TB/s of unique fragment bytes out of L2 in every layer); the two 20 x 20
// patch planes take 120 KB.  Correct and tested (nch = 256), and not faster: see convlstm_bf16()'s note.  An option, off.
template <int NWM, int NWN, bool LSTM = true, int PCS = 3, bool IN_LN = false, int THX = TH>
__global__ __launch_bounds__(64 * NWM * NWN, 1) void convlstm_x6g_kernel(const IgemmDesc d, const unsigned short* __restrict__ wb, int wbytes, int ncols) {
    constexpr int PHX = THX + 4;                       // patch rows
    constexpr int PB = PHX * RP16;                     // one patch plane: 36,864 B (61,440 B with 16 anchor rows)
    static_assert(PCS == 3 || PCS == 2, "pieces");
    static_assert(THX == TH || (THX == 16 && PCS == 2 && NWM == 4 && NWN == 2), "16-row tiles: the fp16 form with 4 x 2 waves");
    constexpr int PW = 20;
    constexpr int NW = NWM * NWN;                      // waves
    constexpr int MT = THX / 2 / NWM;                  // 32-anchor M tiles per wave
    constexpr int NT = 64 * NW;                        // threads
    constexpr int PPP = NT / 8;                        // patch pixels per staging pass
    constexpr int NPJX = IN_LN ? 2 : 4;                // staging passes per round: 4 x 64 pixels cover the patch's 240 with 512 threads (IN_LN: gamma and beta
                                                       // travel with the pixels: two rounds of 2, or the prologue spills);
    constexpr int NRND = (PHX * PW + NPJX * PPP - 1) / (NPJX * PPP);     // 256 threads take two rounds of 4 x 32 (eight passes in one round put the staged pixels in scratch)
    constexpr int RD = (NW == 8 && !(PCS == 2 && NWM == 4 && PIVP_X3_RD8)) ? 4 : 8;      // k-steps of B fragments in registers
    PIVP_SET_MAIN_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const patch = lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave8 % NWM, wn = wave8 / NWM;
    const int half = lane >> 5, l31 = lane & 31;
    const int C = d.C;
    const int n_nblk = LSTM ? C / (8 * NWN) : d.N / (32 * NWN);      // (plain: d.N = rows of the padded pack, a multiple of 64)
    const int H = d.Hin, W = d.Win;
    const int tpr = W / 16, tpi = (H / THX) * tpr;
    const int n_tiles = d.B * tpi;
    int lid = blockIdx.x;
    if ((gridDim.x & 7) == 0) lid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-aware, column-block major
    const int nblk = lid / n_tiles, tile = lid - nblk * n_tiles;
    const int b0 = tile / tpi, trem = tile - b0 * tpi;
    const int y0 = (trem / tpr) * THX, x0 = (trem - (trem / tpr) * tpr) * 16;
    BF_STAMP(0);
    const int c0 = d.c0, ld0 = d.ld0, ld1 = d.ld1;
    const int cin = c0 + d.c1;
    const int ncg_all = (cin + 63) >> 6;
    const int cgbase = (int)blockIdx.y * ncg_all / (int)gridDim.y;                 // this block's channel groups: [cgbase, cgbase + ncg)
    const int ncg = ((int)blockIdx.y + 1) * ncg_all / (int)gridDim.y - cgbase;
    const int nchunks = 25 * ncg;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x0), 0, d.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.c1 ? d.x1 : d.x0), 0, d.c1 ? d.bytes1 : d.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wb), 0, wbytes, 0x00020000);
    constexpr unsigned OOB = 0xC0000000u;

    // ---- patch staging (all 8 waves), as in convlstm_bf16_kernel: thread = (pixel (tid >> 3) + 64 j, 8-channel piece tid & 7), three planes ----
    const int cpiece = tid & 7;
    // (pixel -> image / patch offsets are recomputed where they are used: ten index registers held across the tap loop cost more than the divisions)
    auto pix_of = [&](int j, int& a_pix, int& a_lds) {      // j: pass index over all rounds
        const int p = (tid >> 3) + PPP * j;
        const int py = p / PW, px = p - py * PW;
        const int iy = y0 - 2 + py, ix = x0 - 2 + px;
        const bool ok = p < PHX * PW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        a_pix = ok ? (b0 * H + iy) * W + ix : -1;
        a_lds = p < PHX * PW ? py * RP16 + px * PP : PW * PP;
    };
    float a_scale = 1.0f;                              // (plain form with fp16 pieces: see inv_wscale below)
    if constexpr (PCS == 2 && !LSTM) a_scale = pivp_x3_scale_wave(d.wscale_part);
    f32x4 plo[NPJX], phi[NPJX];
    f32x4 glo[IN_LN ? NPJX : 1], ghi[IN_LN ? NPJX : 1], blo[IN_LN ? NPJX : 1], bhi[IN_LN ? NPJX : 1];      // IN_LN: gamma / beta of the staged pieces
    float ln_mean = 0.f, ln_rstd = 1.f;
    if constexpr (IN_LN) ln_merge_partials(d.in_part, b0, d.in_np, d.in_eps, ln_mean, ln_rstd);      // every wave for itself: a few partials per sample
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(IN_LN ? d.in_g : d.x0), 0, IN_LN ? H * W * c0 * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(IN_LN ? d.in_b : d.x0), 0, IN_LN ? H * W * c0 * 4 : 0, 0x00020000);
    auto patch_load = [&](int cg, int rnd) {
        const int ch = cg * 64 + cpiece * 8;
        const bool s0 = ch < c0, s1 = !s0 && ch < cin;
        const int ld = s0 ? ld0 : ld1, co = s0 ? ch : ch - c0;
#pragma unroll
        for (int j = 0; j < NPJX; ++j) {
            int a_pix, a_lds;
            pix_of(rnd * NPJX + j, a_pix, a_lds);
            const unsigned off = (a_pix >= 0 && (s0 || s1)) ? (unsigned)((a_pix * ld + co) * 4) : OOB;
            if constexpr (IN_LN) {          // (pieces of h channels and pixels outside the image: the zeros of an out-of-range load)
                const unsigned go = (a_pix >= 0 && s0) ? (unsigned)(((a_pix - b0 * H * W) * c0 + co) * 4) : OOB;
                glo[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsg, go, 0, 0));
                ghi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsg, go, 16, 0));
                blo[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, go, 0, 0));
                bhi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, go, 16, 0));
            }
            if (s0) {
                plo[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0));
                phi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 16, 0));
            } else {
                plo[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0));
                phi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 16, 0));
            }
        }
    };
    auto patch_store = [&](int rnd, int cg = 0) {      // cg: which 64-channel group was loaded (IN_LN: its x pieces are normalised)
#pragma unroll
        for (int j = 0; j < NPJX; ++j) {
            int a_pix, a_lds;
            pix_of(rnd * NPJX + j, a_pix, a_lds);
            float r[8] = {plo[j][0], plo[j][1], plo[j][2], plo[j][3], phi[j][0], phi[j][1], phi[j][2], phi[j][3]};
            if constexpr (IN_LN) {
                if (cg * 64 + cpiece * 8 < c0) {
                    const float gm[8] = {glo[j][0], glo[j][1], glo[j][2], glo[j][3], ghi[j][0], ghi[j][1], ghi[j][2], ghi[j][3]};
                    const float bt[8] = {blo[j][0], blo[j][1], blo[j][2], blo[j][3], bhi[j][0], bhi[j][1], bhi[j][2], bhi[j][3]};
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = (r[e] - ln_mean) * ln_rstd * gm[e] + bt[e];
                }
            }
            if constexpr (PCS == 2) {
                if constexpr (!LSTM) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] *= a_scale;
                }
                uint4 hh, ll;
                hh.x = pivp_pack2h_rest(r[0], r[1]); hh.y = pivp_pack2h_rest(r[2], r[3]); hh.z = pivp_pack2h_rest(r[4], r[5]); hh.w = pivp_pack2h_rest(r[6], r[7]);
                ll.x = pivp_pack2h_rest(r[0], r[1]); ll.y = pivp_pack2h_rest(r[2], r[3]); ll.z = pivp_pack2h_rest(r[4], r[5]); ll.w = pivp_pack2h_rest(r[6], r[7]);
                *reinterpret_cast<uint4*>(patch + a_lds + cpiece * 16) = hh;
                *reinterpret_cast<uint4*>(patch + PB + a_lds + cpiece * 16) = ll;
                continue;
            }
            uint4 v;
            v.x = pack2(r[0], r[1]); v.y = pack2(r[2], r[3]); v.z = pack2(r[4], r[5]); v.w = pack2(r[6], r[7]);
            *reinterpret_cast<uint4*>(patch + a_lds + cpiece * 16) = v;
            auto rest = [&](unsigned p2, int i) {          // r[i], r[i + 1] become the remainders (exact in fp32); returns them as bf16
                r[i] -= __builtin_bit_cast(float, p2 << 16); r[i + 1] -= __builtin_bit_cast(float, p2 & 0xffff0000u);
                return pack2(r[i], r[i + 1]);
            };
            uint4 l, q;
            l.x = rest(v.x, 0); l.y = rest(v.y, 2); l.z = rest(v.z, 4); l.w = rest(v.w, 6);
            *reinterpret_cast<uint4*>(patch + PB + a_lds + cpiece * 16) = l;
            q.x = rest(l.x, 0); q.y = rest(l.y, 2); q.z = rest(l.z, 4); q.w = rest(l.w, 6);
            *reinterpret_cast<uint4*>(patch + 2 * PB + a_lds + cpiece * 16) = q;
        }
    };
    const int tap0 = (lid * 7) % 25;
    float inv_wscale = 1.0f;
    if constexpr (PCS == 2) inv_wscale = 1.0f / *reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(wb) + wbytes);   // the pack's tail
    // plain form with fp16 pieces (the data gradient): the activations are gradients, far below fp16's normal range -- they are staged times the power of
    // two that puts the tensor's largest |value| into [2^14, 2^15) (d.wscale_part = absmax_partials of d.x0, one partial per lane), and the sums scaled back
    if constexpr (PCS == 2 && !LSTM) inv_wscale *= 1.0f / a_scale;

    f32x16 acc[MT], accl[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[mt][r] = 0.f; accl[mt][r] = 0.f; }
    int a_off[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int i = 32 * MT * wm + 32 * mt + l31;
        a_off[mt] = (i >> 4) * RP16 + (i & 15) * PP + half * 16;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;

    // ---- the weights: fragment (group, tap, k-step, plane, c8 = nblk * 4 + wn) of the pack, 1 KB in lane order ------------------------
    const unsigned pls = (unsigned)(LSTM ? C / 8 : d.N / 32) * 1024u;    // bytes between the planes of a k-step (one KB per 32-column fragment)
    const unsigned kss = (unsigned)PCS * pls, tps = 4u * kss;     // ... between k-steps, between taps
    const unsigned voff = (unsigned)((nblk * NWN + wn) * 1024 + lane * 16);
    bf16x8 Bf[RD][PCS];                                  // [k-step (of the even / odd tap when RD = 8)][plane]: behind each k-step its registers take the fragments RD k-steps on
    auto bload = [&](bf16x8 (&dst)[PCS], unsigned soff) {
        if constexpr (PIVP_X6_ABL & 8) return;
#pragma unroll
        for (int pl = 0; pl < PCS; ++pl)
            dst[pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, voff, (int)(soff + pl * pls), 0));
    };
    auto adv = [&](int& tp, int& cg) { tp = tp == 24 ? 0 : tp + 1; cg += tp == tap0 ? 1 : 0; };

    // ---- prologue ---------------------------------------------------------------------------------------------------------------------
    patch_load(cgbase, 0);
    int tap = tap0, cg = cgbase, tap1 = tap0, cg1 = cgbase;
    adv(tap1, cg1);
    constexpr bool SHB = PCS == 2 && NW == 8 && LSTM && PIVP_X3_SHARE_B && !PIVP_X3_DOUBLE && !PIVP_X3_RD8;
    constexpr int NF = PCS * NWN;                      // fragments of a k-step of this block: [plane][wave column]
    unsigned char* const bslot = lds + PCS * PB;       // SHB: two slots of a tap's fragments (4 NF KB each) behind the patch
    const bool floader = wave8 < NF;                   // SHB: this wave fetches fragment wave8 = (plane wave8 / NWN, column wave8 % NWN) of every k-step
    const unsigned fvoff = (unsigned)((wave8 / NWN) * pls + (nblk * NWN + (wave8 % NWN)) * 1024 + lane * 16);
    bf16x8 Of[4];                                      // SHB: the own fragment of chunks c (register c & 3), four k-steps of lookahead
    auto oload = [&](bf16x8& dst, unsigned soff) {
        dst = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, fvoff, (int)soff, 0));
    };
#pragma unroll
    for (int ks = 0; ks < (SHB ? 0 : ((PIVP_X6_MIDLOAD && RD == 4) ? 3 : 4)); ++ks) bload(Bf[ks], (unsigned)(cg * 25 + tap) * tps + ks * kss);
    if constexpr (RD == 8) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bload(Bf[4 + ks], (unsigned)(cg1 * 25 + tap1) * tps + ks * kss);
    }
    const int chl = wn * 8 + (l31 & 7);
    const int ch = nblk * 8 * NWN + chl;
    const int grp = l31 >> 3;
    float bj = 0.f, bi = 0.f, bf = 0.f, bo = 0.f;
    float cpre[MT][4];
    patch_store(0, cgbase);
#pragma unroll
    for (int rnd = 1; rnd < NRND; ++rnd) { patch_load(cgbase, rnd); patch_store(rnd, cgbase); }
    BF_STAMP(1);
    __syncthreads();
    BF_STAMP(2);

    bf16x8 fa[2][MT], fal[2][MT], fa3[2][MT];          // [register set][M tile]: the A fragments of a k-step, hi / mid / lo planes
    auto wait_a = [&](auto SET) {
        constexpr int st = decltype(SET)::value;
        if constexpr (PCS == 2 && MT == 2) wait_lgkm(fa[st][0], fa[st][1], fal[st][0], fal[st][1]);
        else if constexpr (PCS == 2) wait_lgkm(fa[st][0], fal[st][0]);
        else if constexpr (MT == 2) wait_lgkm(fa[st][0], fa[st][1], fal[st][0], fal[st][1], fa3[st][0], fa3[st][1]);
        else wait_lgkm(fa[st][0], fal[st][0], fa3[st][0]);
    };
    auto read_a = [&](auto SET, auto KS, auto I, unsigned ab) {       // read I of the 3 MT: plane I / MT, M tile I % MT
        constexpr int st = decltype(SET)::value, ks = decltype(KS)::value, i = decltype(I)::value, pl = i / MT, mt = i % MT;
        if constexpr ((PIVP_X6_ABL & 4) && pl > 0) return;
        bf16x8 v = lds_read_b128<ks * 32>(ab + pl * PB + a_off[mt]);
        if constexpr (pl == 0) fa[st][mt] = v; else if constexpr (pl == 1) fal[st][mt] = v; else fa3[st][mt] = v;
    };
    // twelve MFMAs of register set CUR against the fragments b[3] (hi, mid, lo); corrections into accl, the leading term into acc
    auto mfma = [&](auto CUR, auto I, const bf16x8 (&b)[PCS]) {
        constexpr int st = decltype(CUR)::value, i = decltype(I)::value, term = i / MT, mt = i % MT;
        if constexpr (PCS == 2) {      // fp16 pieces: lo * hi, hi * lo into the corrections, hi * hi into the main accumulator
            auto h = [](const bf16x8& v) { return __builtin_bit_cast(f16x8, v); };
            if constexpr (PIVP_X6_ABL & 32) {   // timing only: the loads are issued and kept alive (sink below), the MFMAs take the A fragments for B too
                if constexpr (term == 0) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fal[st][mt]), h(fa[st][0]), accl[mt], 0, 0, 0);
                else if constexpr (term == 1) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fa[st][mt]), h(fal[st][0]), accl[mt], 0, 0, 0);
                else acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fa[st][mt]), h(fa[st][0]), acc[mt], 0, 0, 0);
                // 32: the loaded registers are consumed (an empty asm: the compiler waits for the load as it would for an MFMA); 32 | 64: never consumed (loads
                // issued, nobody waits -- the compiler still has to keep the ring registers, so the loads are not removed: they feed the next reload's WAW order)
                if constexpr (i == 0 && !(PIVP_X6_ABL & 64)) asm volatile("" :: "v"(b[0]), "v"(b[1]));
                return;
            }
            if constexpr (term == 0) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fal[st][mt]), h(b[0]), accl[mt], 0, 0, 0);
            else if constexpr (term == 1) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fa[st][mt]), h(b[1]), accl[mt], 0, 0, 0);
            else acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fa[st][mt]), h(b[0]), acc[mt], 0, 0, 0);
            return;
        } else {
        if constexpr ((PIVP_X6_ABL & 16) && term != 5) return;
        if constexpr (term == 0) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa3[st][mt], b[0], accl[mt], 0, 0, 0);        // lo * hi
        else if constexpr (term == 1) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][mt], b[2], accl[mt], 0, 0, 0);    // hi * lo
        else if constexpr (term == 2) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[st][mt], b[1], accl[mt], 0, 0, 0);   // mid * mid
        else if constexpr (term == 3) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[st][mt], b[0], accl[mt], 0, 0, 0);   // mid * hi
        else if constexpr (term == 4) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][mt], b[1], accl[mt], 0, 0, 0);    // hi * mid
        else acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][mt], b[0], acc[mt], 0, 0, 0);                               // hi * hi
        }
    };
    // one k-step: wait for its A fragments, then the MFMAs with the six A reads of the NEXT k-step (set NXT, k-step KSN at patch offset abn)
    // behind the first three
    auto kstep = [&](auto CUR, auto NXT, auto KSN, unsigned abn, const bf16x8 (&b)[PCS], auto RD, auto mid) {     // mid(): issued behind the first MFMA
        constexpr bool rd = decltype(RD)::value;
        wait_a(CUR);
#define PIVP_X6_M(I) mfma(CUR, std::integral_constant<int, I>{}, b);
#define PIVP_X6_R(I) if constexpr (rd) read_a(NXT, KSN, std::integral_constant<int, I>{}, abn);
#define PIVP_X6_S __builtin_amdgcn_sched_barrier(0);
        PIVP_X6_S
        if constexpr (PIVP_X6_READS_FIRST && PCS == 2) {          // all reads of the next k-step in front of this one's MFMAs
            PIVP_X6_R(0) PIVP_X6_R(1)
            if constexpr (MT == 2) { PIVP_X6_R(2) PIVP_X6_R(3) }
            PIVP_X6_S
            PIVP_X6_M(0) PIVP_X6_M(1) mid(); PIVP_X6_M(2)
            if constexpr (MT == 2) { PIVP_X6_M(3) PIVP_X6_M(4) PIVP_X6_M(5) }
        } else if constexpr (PCS == 2 && MT == 2) {
            PIVP_X6_M(0) PIVP_X6_R(0) PIVP_X6_R(1) PIVP_X6_S
            PIVP_X6_M(1) PIVP_X6_R(2) PIVP_X6_R(3) PIVP_X6_S
            PIVP_X6_M(2) mid(); PIVP_X6_S
            PIVP_X6_M(3) PIVP_X6_M(4) PIVP_X6_M(5)
        } else if constexpr (PCS == 2) {
            PIVP_X6_M(0) PIVP_X6_R(0) PIVP_X6_R(1) PIVP_X6_S
            PIVP_X6_M(1) mid(); PIVP_X6_S
            PIVP_X6_M(2)
        } else if constexpr (MT == 2) {
            PIVP_X6_M(0) PIVP_X6_R(0) PIVP_X6_R(1) PIVP_X6_S
            PIVP_X6_M(1) PIVP_X6_R(2) PIVP_X6_R(3) PIVP_X6_S
            PIVP_X6_M(2) PIVP_X6_R(4) PIVP_X6_R(5) PIVP_X6_S
            PIVP_X6_M(3) mid(); PIVP_X6_S
            PIVP_X6_M(4) PIVP_X6_M(5) PIVP_X6_M(6) PIVP_X6_M(7) PIVP_X6_M(8) PIVP_X6_M(9) PIVP_X6_M(10) PIVP_X6_M(11)
        } else {
            PIVP_X6_M(0) PIVP_X6_R(0) PIVP_X6_R(1) PIVP_X6_S
            PIVP_X6_M(1) PIVP_X6_R(2) PIVP_X6_S
            PIVP_X6_M(2) mid(); PIVP_X6_S
            PIVP_X6_M(3) PIVP_X6_M(4) PIVP_X6_M(5)
        }
        PIVP_X6_S
#undef PIVP_X6_M
#undef PIVP_X6_R
#undef PIVP_X6_S
    };
    // (the shared-B loop waits for its fragments itself, in front of its barrier)
    auto kstep_nowait = [&](auto CUR, auto NXT, auto KSN, unsigned abn, const bf16x8 (&b)[PCS], auto RD) {
        constexpr bool rd = decltype(RD)::value;
        __builtin_amdgcn_sched_barrier(0);
        static_for<3 * MT>([&](auto I) {
            constexpr int i = decltype(I)::value;
            mfma(CUR, I, b);
            if constexpr (rd && 2 * i < PCS * MT) read_a(NXT, KSN, std::integral_constant<int, 2 * i>{}, abn);
            if constexpr (rd && 2 * i + 1 < PCS * MT) read_a(NXT, KSN, std::integral_constant<int, 2 * i + 1>{}, abn);
            if constexpr (2 * i < PCS * MT + 2) __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    auto read_a_all = [&](unsigned ab) {               // the first k-step of a tap into set 0 (prologue, and behind a restaged patch)
        read_a(S0{}, K0{}, I0{}, ab); read_a(S0{}, K0{}, I1{}, ab);
        if constexpr (PCS * MT > 2) read_a(S0{}, K0{}, I2{}, ab);
        if constexpr (PCS * MT > 3) read_a(S0{}, K0{}, I3{}, ab);
        if constexpr (PCS * MT > 4) { read_a(S0{}, K0{}, I4{}, ab); read_a(S0{}, K0{}, I5{}, ab); }
    };
    auto a_base = [&](int tp) { const int ty = tp / 5; return lds0 + ty * RP16 + (tp - ty * 5) * PP; };

    // ---- two fp16 pieces, eight waves: two k-steps per wait -------------------------------------------------------------------------------------
    constexpr bool DBL = PCS == 2 && NW == 8 && PIVP_X3_DOUBLE;
    constexpr int NRD = 2 * 2 * MT;                    // A reads of a k-step pair: [k-step][plane][M tile]
    constexpr int NMD = 2 * 3 * MT;                    // MFMAs of a k-step pair: [k-step][term][M tile]
    bf16x8 ga[2][2][2][MT];                            // [register set][k-step of the pair][plane][M tile]
    auto wait_d = [&](auto SET) {
        constexpr int st = decltype(SET)::value;
        if constexpr (MT == 2) wait_lgkm(ga[st][0][0][0], ga[st][0][0][1], ga[st][0][1][0], ga[st][0][1][1], ga[st][1][0][0], ga[st][1][0][1], ga[st][1][1][0], ga[st][1][1][1]);
        else wait_lgkm(ga[st][0][0][0], ga[st][0][1][0], ga[st][1][0][0], ga[st][1][1][0]);
    };
    auto read_d = [&](auto SET, auto PAIR, auto I, unsigned ab) {      // PAIR: which half of the tap (k-steps 2 PAIR, 2 PAIR + 1)
        constexpr int st = decltype(SET)::value, pair = decltype(PAIR)::value, i = decltype(I)::value;
        constexpr int kk = i / (2 * MT), pl = (i / MT) % 2, mt = i % MT, ks = 2 * pair + kk;
        ga[st][kk][pl][mt] = lds_read_b128<ks * 32>(ab + pl * PB + a_off[mt]);
    };
    auto mfma_d = [&](auto CUR, auto I, const bf16x8 (&b0)[PCS], const bf16x8 (&b1)[PCS]) {
        constexpr int st = decltype(CUR)::value, i = decltype(I)::value, kk = i / (3 * MT), term = (i / MT) % 3, mt = i % MT;
        auto h = [](const bf16x8& v) { return __builtin_bit_cast(f16x8, v); };
        const bf16x8& bh = kk ? b1[0] : b0[0];
        const bf16x8& bl = kk ? b1[PCS - 1] : b0[PCS - 1];
        if constexpr (term == 0) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(ga[st][kk][1][mt]), h(bh), accl[mt], 0, 0, 0);       // lo * hi
        else if constexpr (term == 1) accl[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(ga[st][kk][0][mt]), h(bl), accl[mt], 0, 0, 0);  // hi * lo
        else acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(ga[st][kk][0][mt]), h(bh), acc[mt], 0, 0, 0);                             // hi * hi
    };
    auto dstep = [&](auto CUR, auto NXT, auto PAIRN, unsigned abn, const bf16x8 (&b0)[PCS], const bf16x8 (&b1)[PCS], auto RD) {
        constexpr bool rd = decltype(RD)::value;
        wait_d(CUR);
        __builtin_amdgcn_sched_barrier(0);
        static_for<NMD>([&](auto I) {                  // two reads of the next pair behind each of the first MFMAs
            constexpr int i = decltype(I)::value;
            mfma_d(CUR, I, b0, b1);
            if constexpr (rd && 2 * i < NRD) read_d(NXT, PAIRN, std::integral_constant<int, 2 * i>{}, abn);
            if constexpr (rd && 2 * i + 1 < NRD) read_d(NXT, PAIRN, std::integral_constant<int, 2 * i + 1>{}, abn);
            if constexpr (2 * i < NRD + 2) __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_sched_barrier(0);
    };

    if constexpr (RD == 8) {
        // one wave per SIMD: the taps of all groups as one sequence, two per iteration (the ring's halves), fragments requested two taps ahead
        int tap2 = tap1, cg2 = cg1;
        adv(tap2, cg2);
        auto tap_body = [&](auto P, int it) __attribute__((always_inline)) {
            constexpr int p = decltype(P)::value;
            if (LSTM && it == nchunks - 25) {          // in front of the last group's taps: the epilogue's operands
                bj = d.bias[ch]; bi = d.bias[C + ch]; bf = d.bias[2 * C + ch] + 1.0f; bo = d.bias[3 * C + ch];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int r = k * 4 + grp;
                        const int i = 32 * MT * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half;
                        const int m = (b0 * H + y0 + (i >> 4)) * W + x0 + (i & 15);
                        cpre[mt][k] = d.cstate_in[(size_t)m * C + ch];
                    }
            }
            const unsigned ab = a_base(tap), ab1 = a_base(tap1);
            const bool has1 = it + 1 < nchunks, has2 = it + 2 < nchunks;
            const bool regroup = has1 && cg1 != cg;
            const unsigned so2 = (unsigned)(cg2 * 25 + tap2) * tps;
            kstep(S0{}, S1{}, K1{}, ab, Bf[4 * p + 0], std::true_type{}, []{});
            if (has2) bload(Bf[4 * p + 0], so2);
            kstep(S1{}, S0{}, K2{}, ab, Bf[4 * p + 1], std::true_type{}, []{});
            if (has2) bload(Bf[4 * p + 1], so2 + kss);
            kstep(S0{}, S1{}, K3{}, ab, Bf[4 * p + 2], std::true_type{}, []{});
            if (has2) bload(Bf[4 * p + 2], so2 + 2 * kss);
            if (has1 && !regroup) kstep(S1{}, S0{}, K0{}, ab1, Bf[4 * p + 3], std::true_type{}, []{});
            else kstep(S1{}, S0{}, K0{}, ab1, Bf[4 * p + 3], std::false_type{}, []{});
            if (has2) bload(Bf[4 * p + 3], so2 + 3 * kss);
            if (regroup) {                             // next 64 input channels: every wave is done with the old patch
                __syncthreads();
#pragma unroll
                for (int rnd = 0; rnd < NRND; ++rnd) { patch_load(cg1, rnd); patch_store(rnd, cg1); }
                __syncthreads();
                read_a_all(ab1);
            }
            tap = tap1; cg = cg1; tap1 = tap2; cg1 = cg2;
            adv(tap2, cg2);
        };
        read_a_all(a_base(tap));
        for (int it = 0; it < nchunks; it += 2) {
            tap_body(std::integral_constant<int, 0>{}, it);
            if (it + 1 < nchunks) tap_body(std::integral_constant<int, 1>{}, it + 1);
        }
    } else {
    for (int g = 0; g < ncg; ++g) {                    // 64 input channels of concat(x, h) at a time
        if (g > 0) {                                   // every wave is done with the old patch
            __syncthreads();
#pragma unroll
            for (int rnd = 0; rnd < NRND; ++rnd) { patch_load(cgbase + g, rnd); patch_store(rnd, cgbase + g); }
            __syncthreads();
        }
        if (LSTM && g == ncg - 1) {
            // the epilogue's operands, requested in front of the last 25 taps (read in the epilogue they cost an exposed HBM round trip per row;
            // requested in the prologue they hold 12 registers through every tap loop)
            bj = d.bias[ch]; bi = d.bias[C + ch]; bf = d.bias[2 * C + ch] + 1.0f; bo = d.bias[3 * C + ch];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = k * 4 + grp;
                    const int i = 32 * MT * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int m = (b0 * H + y0 + (i >> 4)) * W + x0 + (i & 15);
                    cpre[mt][k] = d.cstate_in[(size_t)m * C + ch];
                }
        }
        if constexpr (SHB) {
            // Tap T of this block's sequence (all groups) has NF fragments per k-step.  NF waves fetch one each (a tap ahead, registers Of[k-step]) and
            // write tap T + 1's into LDS slot (T + 1) & 1 during the first three k-steps of tap T; every wave reads its two planes from slot T & 1.
            // ONE barrier per tap, at the top of its last k-step: behind it slot (T + 1) & 1 is complete (so the first fragments of tap T + 1 can be
            // requested) and slot T & 1, whose last fragments are in registers by then, is free for tap T + 2.
            const int NT25 = nchunks;
            bf16x8 bq[2][PCS];                          // [register set][plane]: the B fragments of the k-step in flight / in use
            constexpr int SLOT = 4 * NF * 1024;         // bytes of a tap's fragments
            auto tap_off = [&](int T) {                 // byte offset of tap T's weights (group, rotated tap)
                const int gg = T / 25, tt = T - gg * 25;
                int tp = tap0 + tt; tp -= tp >= 25 ? 25 : 0;
                return (unsigned)((cgbase + gg) * 25 + tp) * tps;
            };
            auto bread = [&](auto SET, int T, int ks) {  // fragments of (tap T, k-step ks): this wave's column, both planes
                constexpr int st = decltype(SET)::value;
                const unsigned sb = lds0 + PCS * PB + (T & 1) * SLOT + ks * (NF * 1024) + wn * 1024 + lane * 16;
#pragma unroll
                for (int pl = 0; pl < PCS; ++pl) bq[st][pl] = lds_read_b128<0>(sb + pl * NWN * 1024);
            };
            auto bwrite = [&](const bf16x8& v, int T, int ks) {
                *reinterpret_cast<bf16x8*>(bslot + (T & 1) * SLOT + ks * (NF * 1024) + wave8 * 1024 + lane * 16) = v;
            };
            if (g == 0) {                                // tap 0 into slot 0, tap 1 into the registers
                if (floader) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) oload(Of[ks], tap_off(0) + ks * kss);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) bwrite(Of[ks], 0, ks);
                    if (NT25 > 1) {
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) oload(Of[ks], tap_off(1) + ks * kss);
                    }
                }
                __syncthreads();
                bread(S0{}, 0, 0);
            }
            read_a_all(a_base(tap));
            for (int t = 0; t < 25; ++t) {
                const unsigned ab = a_base(tap), ab1 = a_base(tap1);
                const int T = g * 25 + t;
                const bool w1 = floader && T + 1 < NT25, l2 = floader && T + 2 < NT25;
                const unsigned o2 = l2 ? tap_off(T + 2) : 0u;
                auto waitf = [&](auto CUR) {
                    constexpr int st = decltype(CUR)::value;
                    if constexpr (MT == 2) wait_lgkm(fa[st][0], fa[st][1], fal[st][0], fal[st][1], bq[st][0], bq[st][1]);
                    else wait_lgkm(fa[st][0], fal[st][0], bq[st][0], bq[st][1]);
                };
                // k-step 0: tap T + 1's fragments of k-steps 0, 1 go to their slot
                waitf(S0{});
                bread(S1{}, T, 1);
                if (w1) { bwrite(Of[0], T + 1, 0); bwrite(Of[1], T + 1, 1); }
                if (l2) { oload(Of[0], o2); oload(Of[1], o2 + kss); }
                kstep_nowait(S0{}, S1{}, K1{}, ab, bq[0], std::true_type{});
                // k-step 1
                waitf(S1{});
                bread(S0{}, T, 2);
                if (w1) bwrite(Of[2], T + 1, 2);
                if (l2) oload(Of[2], o2 + 2 * kss);
                kstep_nowait(S1{}, S0{}, K2{}, ab, bq[1], std::true_type{});
                // k-step 2
                waitf(S0{});
                bread(S1{}, T, 3);
                if (w1) bwrite(Of[3], T + 1, 3);
                if (l2) oload(Of[3], o2 + 3 * kss);
                kstep_nowait(S0{}, S1{}, K3{}, ab, bq[0], std::true_type{});
                // k-step 3: the tap's barrier (every LDS write above is complete: lgkmcnt(0) in waitf)
                waitf(S1{});
                __builtin_amdgcn_s_barrier();
                if (T + 1 < NT25) bread(S0{}, T + 1, 0);
                if (t < 24) kstep_nowait(S1{}, S0{}, K0{}, ab1, bq[1], std::true_type{});
                else kstep_nowait(S1{}, S0{}, K0{}, ab1, bq[1], std::false_type{});
                tap = tap1; cg = cg1;
                adv(tap1, cg1);
            }
            continue;
        }
        if constexpr (DBL) {
            // two k-steps per wait: set 0 = k-steps 0, 1 of a tap, set 1 = k-steps 2, 3
            static_for<NRD>([&](auto I) { read_d(S0{}, K0{}, I, a_base(tap)); });
            for (int t = 0; t < 25; ++t) {
                const unsigned ab = a_base(tap), ab1 = a_base(tap1);
                const bool has1 = t < 24 || g + 1 < ncg;
                const unsigned so1 = (unsigned)(cg1 * 25 + tap1) * tps;
                dstep(S0{}, S1{}, K1{}, ab, Bf[0], Bf[1], std::true_type{});
                if (has1) { bload(Bf[0], so1); bload(Bf[1], so1 + kss); }
                if (t < 24) dstep(S1{}, S0{}, K0{}, ab1, Bf[2], Bf[3], std::true_type{});
                else dstep(S1{}, S0{}, K0{}, ab1, Bf[2], Bf[3], std::false_type{});
                if (has1) { bload(Bf[2], so1 + 2 * kss); bload(Bf[3], so1 + 3 * kss); }
                tap = tap1; cg = cg1;
                adv(tap1, cg1);
            }
            continue;
        }
        read_a_all(a_base(tap));
        for (int t = 0; t < 25; ++t) {                 // one tap = four k-steps; behind each k-step its registers take the next tap's fragments
            const unsigned ab = a_base(tap), ab1 = a_base(tap1);
            const bool has1 = t < 24 || g + 1 < ncg;
            const unsigned so1 = (unsigned)(cg1 * 25 + tap1) * tps;
#if PIVP_X6_MIDLOAD
            // the fragment loads go out in the MIDDLE of a k-step's MFMAs (their issue then overlaps the wave's own matrix work), into the registers
            // of the k-step before: three k-steps ahead of their use
            const unsigned soc = (unsigned)(cg * 25 + tap) * tps;
            kstep(S0{}, S1{}, K1{}, ab, Bf[0], std::true_type{}, [&] { bload(Bf[3], soc + 3 * kss); });
            kstep(S1{}, S0{}, K2{}, ab, Bf[1], std::true_type{}, [&] { if (has1) bload(Bf[0], so1); });
            kstep(S0{}, S1{}, K3{}, ab, Bf[2], std::true_type{}, [&] { if (has1) bload(Bf[1], so1 + kss); });
            if (t < 24) kstep(S1{}, S0{}, K0{}, ab1, Bf[3], std::true_type{}, [&] { if (has1) bload(Bf[2], so1 + 2 * kss); });
            else kstep(S1{}, S0{}, K0{}, ab1, Bf[3], std::false_type{}, [&] { if (has1) bload(Bf[2], so1 + 2 * kss); });
#else
            kstep(S0{}, S1{}, K1{}, ab, Bf[0], std::true_type{}, []{});
            if (has1) bload(Bf[0], so1);
            kstep(S1{}, S0{}, K2{}, ab, Bf[1], std::true_type{}, []{});
            if (has1) bload(Bf[1], so1 + kss);
            kstep(S0{}, S1{}, K3{}, ab, Bf[2], std::true_type{}, []{});
            if (has1) bload(Bf[2], so1 + 2 * kss);
            if (t < 24) kstep(S1{}, S0{}, K0{}, ab1, Bf[3], std::true_type{}, []{});
            else kstep(S1{}, S0{}, K0{}, ab1, Bf[3], std::false_type{}, []{});      // (the next tap's A fragments come from the next patch)
            if (has1) bload(Bf[3], so1 + 3 * kss);
#endif
            tap = tap1; cg = cg1;
            adv(tap1, cg1);
        }
    }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[mt][r] += accl[mt][r];
            if constexpr (PCS == 2) acc[mt][r] *= inv_wscale;            // (the weights were packed times a power of two)
        }
    BF_STAMP(3);

    if constexpr (!LSTM) {
        // ---- plain epilogue: accumulator row = anchor, column = output channel; 32 lanes write 128 contiguous bytes -------------------
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = 32 * MT * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half;
                const size_t m = (size_t)((b0 * H + y0 + (i >> 4)) * W + x0 + (i & 15));
                const int col = (nblk * NWN + wn) * 32 + l31;
                if (col < ncols) {                      // the pack's rows past the real column count are zero padding
                    float* o = d.out + m * d.ldo + col;
                    if (gridDim.y > 1) atomicAdd(o, acc[mt][r]);
                    else if (d.accum) *o += acc[mt][r];
                    else *o = acc[mt][r];
                }
            }
        return;
    }
    // ---- epilogue: the gate math of convlstm_bf16_kernel's 16-channel blocks (a wave's 32 columns = 4 gates x 8 channels) -------------
    auto pick = [&](const float (&v)[4], int idx) -> float {
        const float lo = (idx & 1) ? v[1] : v[0], hi = (idx & 1) ? v[3] : v[2];
        return (idx & 2) ? hi : lo;
    };
    float sv[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float rows[4], val[4], g4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) rows[g] = acc[mt][k * 4 + g];
            val[0] = pick(rows, grp);
#pragma unroll
            for (int x = 1; x < 4; ++x) val[x] = __shfl_xor(pick(rows, grp ^ x), x * 8, 64);
#pragma unroll
            for (int g = 0; g < 4; ++g) g4[g] = pick(val, g ^ grp);
            const int r = k * 4 + grp;
            const int i = 32 * MT * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int m = (b0 * H + y0 + (i >> 4)) * W + x0 + (i & 15);
            const size_t o = (size_t)m * C + ch;
            const float aj = b_tanh(g4[0] + bj), ai = b_sigmoid(g4[1] + bi);
            const float af = b_sigmoid(g4[2] + bf), ao = b_sigmoid(g4[3] + bo);
            const float cn = cpre[mt][k] * af + ai * aj;
            d.cstate_out[o] = cn;
            const float hn = b_tanh(cn) * ao;
            d.hout[o] = hn;
            sv[mt][k] = hn;
            if (d.gates_out) {
                float* gp = d.gates_out + (size_t)m * 4 * C + ch;
                gp[0] = aj; gp[C] = ai; gp[2 * C] = af; gp[3 * C] = ao;
            }
        }
    BF_STAMP(4);
#ifdef PIVP_BF16_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BF_STAMP(5);
#endif
    if (d.ln_part) {                                   // (count, mean, M2) of the block's h tile, two passes, fixed order over the eight waves
        float* red = reinterpret_cast<float*>(lds);
        float s1 = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int k = 0; k < 4; ++k) s1 += sv[mt][k];
        s1 = wave_sum(s1);
        __syncthreads();
        if (lane == 0) red[wave8] = s1;
        __syncthreads();
        const float cnt = (float)NW * 64.f * 4.f * MT;
        float ssum = (red[0] + red[1]) + (red[2] + red[3]);
        if constexpr (NW == 8) ssum += (red[4] + red[5]) + (red[6] + red[7]);
        const float mean = ssum / cnt;
        float q = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float dd = sv[mt][k] - mean; q = fmaf(dd, dd, q); }
        q = wave_sum(q);
        if (lane == 0) red[8 + wave8] = q;
        __syncthreads();
        if (tid == 0) {
            float* p = d.ln_part + ((size_t)b0 * d.ln_nparts + (size_t)trem * n_nblk + nblk) * 4;
            float qs = (red[8] + red[9]) + (red[10] + red[11]);
            if constexpr (NW == 8) qs += (red[12] + red[13]) + (red[14] + red[15]);
            p[0] = cnt; p[1] = mean; p[2] = qs; p[3] = 0.f;
        }
    }
}

#ifdef PIVP_BF16_STAMPS
}
extern "C" int pivp_debug_bf16_stamps(long long* out, int n) {   // n <= 2048 * 8: [block][entry, prologue done, first barrier passed, tap loop done, cells done]
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pivp::pivp_bf16_stamps), (size_t)n * sizeof(long long)) == hipSuccess ? 0 : -2;
}
namespace pivp {
#endif

// (the kernel reads whole float4s: n a multiple of 4 and w 16-byte aligned -- every tensor it is used on is; anything else is refused, a tail
// left out of the maximum would let the fp16 hi piece saturate)
int absmax_partials(const float* w, long n, float* tail, hipStream_t stream) {
    PIVP_CHECK_ARG(w && tail && n > 0 && n % 4 == 0 && ((uintptr_t)w & 15) == 0);
    hipLaunchKernelGGL(absmax_partials_kernel, dim3(64), dim3(512), 0, stream, w, n, tail);
    return PIVP_LAUNCH_STATUS();
}

size_t lstm_bf16_weight_elems(int wcin, int N) { return (size_t)((wcin + 63) / 64) * 25 * N * 64; }
// rows of the bf16 pack of a plain 5x5 convolution with N output channels: whole 128- or 64-column blocks
int conv5x5_bf16_rows(int N) { return N % 128 == 0 ? N : (N + 63) / 64 * 64; }

// w: fp32 K-inner packed [25][wcin/32][N][32]; wb: [ceil(wcin/64)][25][planes][Np][64] bf16 (Np >= N rows, the extra ones zero; 0 = N;
// planes = 2: the hi / lo split of the split mode, lstm_bf16_weight_elems(wcin, Np) * 2 elements)
// planes = 3: three bf16 pieces, fragment-major; planes = -2: two fp16 pieces of 256 w, fragment-major (lstm_bf16_weight_elems * 2 elements)
int pack_lstm_bf16(const float* w, unsigned short* wb, int wcin, int N, hipStream_t s, int Np, int planes, int plain) {
    if (Np == 0) Np = N;
    PIVP_CHECK_ARG(w && wb && wcin > 0 && wcin % 32 == 0 && N > 0 && Np >= N && ((planes >= 1 && planes <= 3) || planes == -2));
    const int pieces = planes == -2 ? 2 : planes;
    const long total = (long)lstm_bf16_weight_elems(wcin, Np) * pieces;
    if (planes == -2 && plain == 2) {     // two fp16 pieces in the RING kernel's layout (layers on 8-wide maps: convlstm_bf16_kernel<NCH, true, 2, true>)
        hipLaunchKernelGGL(absmax_partials_kernel, dim3(64), dim3(512), 0, s, w, (long)25 * wcin * N, reinterpret_cast<float*>(wb + total));
        hipLaunchKernelGGL(pack_lstm_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, wb, wcin, N, Np, 2, total, 1);
        return PIVP_LAUNCH_STATUS();
    }
    if (planes == -2)        // the tensor's scale first: 64 partial maxima into the pack's tail
        hipLaunchKernelGGL(absmax_partials_kernel, dim3(64), dim3(512), 0, s, w, (long)25 * wcin * N, reinterpret_cast<float*>(wb + total));
    if (planes == 3 || planes == -2) {       // the three-piece kernels' fragment-major pack: the cell's (N = 4 C, gate-interleaved fragments) or a plain conv's
        PIVP_CHECK_ARG(Np % 64 == 0 && (plain || (Np == N && N % 32 == 0)));
        const long nthreads = (long)lstm_bf16_weight_elems(wcin, Np) / 8;
        hipLaunchKernelGGL(pack_lstm_x6_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s, w, wb, wcin, N, Np, plain, pieces, nthreads);
        return PIVP_LAUNCH_STATUS();
    }
    hipLaunchKernelGGL(pack_lstm_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, wb, wcin, N, Np, planes, total, 0);
    return PIVP_LAUNCH_STATUS();
}

static bool bf16_geometry_ok(const IgemmDesc& d) {
    if (d.ksize != 5 || d.pad != 2 || d.in_step != 1 || d.Hin % TH) return false;
    if (d.c0 % 8 || d.c1 % 8 || d.ld0 % 4 || d.ld1 % 4) return false;
    if (d.Win % 16 == 0) return true;
    return d.Win % 8 == 0 && d.B % 2 == 0;
}
bool convlstm_bf16_ok(const IgemmDesc& d) { return bf16_geometry_ok(d) && d.C > 0 && d.C % 16 == 0; }
bool convlstm_bf16x6_ok(const IgemmDesc& d) { return convlstm_bf16_ok(d) && d.Win % 16 == 0; }

template <int NCH, bool LSTM, int PL = 1, bool F16 = false>
static int launch_bf16(const IgemmDesc& d, const unsigned short* wb, hipStream_t stream, int* ln_nparts, int nb, int ksplit, int ncols) {
    constexpr int lds_bytes = PL * patch_plane_bytes<PL>() + (PL == 3 ? 8 * X6_CHUNK : (ring_depth<NCH, PL>() + 1) * PL * 4 * NCH * 128);
    static PerDeviceOnce once;
    if (pivp_ensure_dyn_lds(once, reinterpret_cast<const void*>(&convlstm_bf16_kernel<NCH, LSTM, PL, F16>), lds_bytes) != PIVP_OK) return PIVP_ERR_LAUNCH;
    IgemmDesc dd = d;
    const int tw = d.Win % 16 == 0 ? 16 : 8, ti_n = tw == 16 ? 1 : 2;
    const int tpi = (d.Hin / TH) * (d.Win / tw);
    const int np = tpi * nb;
    dd.ln_nparts = (LSTM && d.ln_part && np <= d.ln_cap) ? np : 0;
    if (!dd.ln_nparts) dd.ln_part = nullptr;
    if (ln_nparts) *ln_nparts = dd.ln_nparts;
    const int blocks = (d.B / ti_n) * tpi * nb;
    hipLaunchKernelGGL((convlstm_bf16_kernel<NCH, LSTM, PL, F16>), dim3(blocks, ksplit), dim3(512), lds_bytes, stream, dd, wb, tw, ncols);
    return PIVP_LAUNCH_STATUS();
}

template <int NWM, int NWN, int PCS, bool IN_LN, int THX = TH>
static int launch_x6g_impl(const IgemmDesc& d, const unsigned short* wb, hipStream_t stream, int* ln_nparts) {
    constexpr int lds_bytes = PCS * (THX + 4) * RP16 + ((PCS == 2 && NWM * NWN == 8 && THX == TH && PIVP_X3_SHARE_B) ? 2 * 4 * PCS * NWN * 1024 : 0);       // (+ the shared-B tap slots of the fp16 forms)
    static_assert(lds_bytes <= 160 * 1024, "LDS");
    static PerDeviceOnce once;
    if (pivp_ensure_dyn_lds(once, reinterpret_cast<const void*>(&convlstm_x6g_kernel<NWM, NWN, true, PCS, IN_LN, THX>), lds_bytes) != PIVP_OK) return PIVP_ERR_LAUNCH;
    PIVP_CHECK_ARG(d.Hin % THX == 0);
    IgemmDesc dd = d;
    const int tpi = (d.Hin / THX) * (d.Win / 16), nb = d.C / (8 * NWN);
    const int np = tpi * nb;
    dd.ln_nparts = (d.ln_part && np <= d.ln_cap) ? np : 0;
    if (!dd.ln_nparts) dd.ln_part = nullptr;
    if (ln_nparts) *ln_nparts = dd.ln_nparts;
    const long long wbytes = (long long)lstm_bf16_weight_elems(d.c0 + (d.c1 ? d.c1 : d.C), 4 * d.C) * PCS * 2;
    if (wbytes >= (1LL << 31)) return PIVP_ERR_BADARG;
    hipLaunchKernelGGL((convlstm_x6g_kernel<NWM, NWN, true, PCS, IN_LN, THX>), dim3(d.B * tpi * nb), dim3(64 * NWM * NWN), lds_bytes, stream, dd, wb, (int)wbytes, 0);
    return PIVP_LAUNCH_STATUS();
}
template <int NWM, int NWN, int PCS = 3, int THX = TH>
static int launch_x6g(const IgemmDesc& d, const unsigned short* wb, hipStream_t stream, int* ln_nparts) {
    if (d.in_g) {       // the x operand's LayerNorm applied while staging (eight-wave forms only; x in one 64-channel group)
        if constexpr (NWM * NWN == 8) {
            PIVP_CHECK_ARG(d.in_b && d.in_part && d.in_np > 0 && d.c0 <= 64 && d.ld0 == d.c0 && d.in_part != d.ln_part);
            return launch_x6g_impl<NWM, NWN, PCS, true, THX>(d, wb, stream, ln_nparts);
        } else return PIVP_ERR_BADARG;
    }
    return launch_x6g_impl<NWM, NWN, PCS, false, THX>(d, wb, stream, ln_nparts);
}

// the plain 5x5 convolution on the same kernel: dd.N = rows of the padded pack, nb = its 64-column blocks, ks = split of the channel groups
template <int PCS>
static int launch_x6g_plain(const IgemmDesc& dd, const unsigned short* wb, hipStream_t stream, int nb, int ks, int ncols) {
    constexpr int lds_bytes = PCS * PH * RP16;
    static PerDeviceOnce once;
    if (pivp_ensure_dyn_lds(once, reinterpret_cast<const void*>(&convlstm_x6g_kernel<4, 2, false, PCS>), lds_bytes) != PIVP_OK) return PIVP_ERR_LAUNCH;
    const int tpi = (dd.Hin / TH) * (dd.Win / 16);
    const long long wbytes = (long long)lstm_bf16_weight_elems(dd.c0 + dd.c1, dd.N) * PCS * 2;
    if (wbytes >= (1LL << 31)) return PIVP_ERR_BADARG;
    hipLaunchKernelGGL((convlstm_x6g_kernel<4, 2, false, PCS>), dim3(dd.B * tpi * nb, ks), dim3(512), lds_bytes, stream, dd, wb, (int)wbytes, ncols);
    return PIVP_LAUNCH_STATUS();
}

// d as for igemm_lstm (validated by the caller's igemm_validate(d, true) equivalent); wb = pack_lstm_bf16(d.w, ..., planes).
int convlstm_bf16(const IgemmDesc& d, const unsigned short* wb, hipStream_t stream, int* ln_nparts, int nch, int planes) {
    PIVP_CHECK_ARG(wb && convlstm_bf16_ok(d) && (nch == 0 || nch == 16 || (nch == 32 && d.C % 32 == 0) || ((nch == 1 || nch == 2) && planes == 3) ||
                                                 (nch == 256 && planes == -2 && d.Hin % 16 == 0 && d.Win % 16 == 0)) &&
                   ((planes >= 1 && planes <= 3) || planes == -2));
    // LayerNorm-on-load (d.in_g) exists in the eight-wave L2-direct kernels only (launch_x6g): every other form would consume the raw tensor
    PIVP_CHECK_ARG(!d.in_g || ((planes == 3 || planes == -2) && convlstm_bf16x6_ok(d) && nch != 1));
    if (planes == -2 && !convlstm_bf16x6_ok(d)) {   // 8-wide maps: the ring kernel with fp16 pieces (wb = pack_lstm_bf16(..., planes = -2, plain = 2))
        const int tw2 = 8, ti2 = 2;
        const long b32 = d.C % 32 ? 0 : (long)(d.B / ti2) * (d.Hin / TH) * (d.Win / tw2) * (d.C / 32);
        (void)tw2; (void)ti2; (void)b32;      // 16-channel blocks only: with the second accumulator a 32-channel block would need 286 registers
        return launch_bf16<16, true, 2, true>(d, wb, stream, ln_nparts, d.C / 16, 1, 0);
    }
    if (planes == -2) {   // two fp16 pieces, three MFMAs per product (wb = pack_lstm_bf16(..., planes = -2)): the L2-direct kernel, 16-wide tiles
        PIVP_CHECK_ARG(convlstm_bf16x6_ok(d));
        // 16 x 16 tiles (256 anchors x 16 channels per block: half the weight bytes out of L2 per multiply-add) where they still give every CU a block:
        // PIVP_X3_TH16=1 or nch = 256.  MEASURED (profiles/r04/fp16x3_layers_tile16_ab.txt): lstm1 / 2 / 7 41.8 / 41.7 / 72.9 us against 40.5 / 40.7 / 71.1 with
        // 8-row tiles -- the weight stream was never the bound (clock_power_operand_values.txt: these loops run power-limited at 1.82 GHz).  Off.
        static const int th16 = [] { const char* e = getenv("PIVP_X3_TH16"); return e ? atoi(e) : 0; }();
        const long b256 = d.Hin % 16 ? 0 : (long)d.B * (d.Hin / 16) * (d.Win / 16) * (d.C / 16);
        if (nch == 256 || (nch == 0 && th16 && b256 >= pivp_cu_count())) return launch_x6g<4, 2, 2, 16>(d, wb, stream, ln_nparts);
        const long b32 = d.C % 32 ? 0 : (long)d.B * (d.Hin / TH) * (d.Win / 16) * (d.C / 32);
        if ((nch == 32 && b32 > 0) || (nch == 0 && b32 >= pivp_cu_count())) return launch_x6g<2, 4, 2>(d, wb, stream, ln_nparts);
        return launch_x6g<4, 2, 2>(d, wb, stream, ln_nparts);
    }
    if (planes == 3) {   // three pieces: 16-wide tiles and 16-channel blocks only (convlstm_bf16x6_ok); 8-wide maps are the caller's to route elsewhere
        PIVP_CHECK_ARG(convlstm_bf16x6_ok(d));
        // weights straight from L2: 32-channel blocks (eight waves) where they give every CU a block, else 16-channel blocks (four waves);
        // nch = 1 (op tests): the 16-channel blocks on the LDS-ring kernel instead
        const long b32 = d.C % 32 ? 0 : (long)d.B * (d.Hin / TH) * (d.Win / 16) * (d.C / 32);
        if ((nch == 32 && b32 > 0) || (nch == 0 && b32 >= pivp_cu_count())) return launch_x6g<2, 4>(d, wb, stream, ln_nparts);
        if (nch == 1) return launch_bf16<16, true, 3>(d, wb, stream, ln_nparts, d.C / 16, 1, 0);
        if (nch == 2) return launch_x6g<2, 2>(d, wb, stream, ln_nparts);      // (four waves, one per SIMD: measured against the eight-wave form)
        return launch_x6g<4, 2>(d, wb, stream, ln_nparts);
    }
    if (planes == 2) {   // split mode: 32-channel blocks (two ring slots) when they still give every CU a block, else 16-channel ones
        const int tw2 = d.Win % 16 == 0 ? 16 : 8, ti2 = tw2 == 16 ? 1 : 2;
        const long b32 = d.C % 32 ? 0 : (long)(d.B / ti2) * (d.Hin / TH) * (d.Win / tw2) * (d.C / 32);
        if (nch == 32 || (nch == 0 && b32 >= 256)) return launch_bf16<32, true, 2>(d, wb, stream, ln_nparts, d.C / 32, 1, 0);
        return launch_bf16<16, true, 2>(d, wb, stream, ln_nparts, d.C / 16, 1, 0);
    }
    const int tw = d.Win % 16 == 0 ? 16 : 8, ti_n = tw == 16 ? 1 : 2;
    const long blocks32 = (long)(d.B / ti_n) * (d.Hin / TH) * (d.Win / tw) * (d.C / 32);
    if (nch == 0) nch = (d.C % 32 || blocks32 < 256) ? 16 : 32;
    return nch == 16 ? launch_bf16<16, true>(d, wb, stream, ln_nparts, d.C / 16, 1, 0)
                     : launch_bf16<32, true>(d, wb, stream, ln_nparts, d.C / 32, 1, 0);
}

// Plain 5x5 stride-1 "same" convolution with bf16 operands: out[m][n] (+)= sum_{tap, k} x[m + tap][k] w[tap][k][n], n < d.N, written
// with pixel stride d.ldo.  x = d.x0 | d.x1 (fp32 NHWC, rounded to bf16 on the way into LDS); wb = pack_lstm_bf16(w, c0 + c1, d.N,
// conv5x5_bf16_rows(d.N)).  d.accum adds into out; d.ksplit_ok (out pre-zeroed, no accum) lets grids that would leave CUs idle split
// the channel groups over gridDim.y and meet in out by atomic adds.  This is the ConvLSTM data gradient (x = dG, 4C channels).
// ks > 1 (the K split conv5x5_bf16 will use) needs a zeroed destination: the caller asks first so that it only clears when needed
int conv5x5_bf16_ksplit(const IgemmDesc& d, int planes) {
    const int Np = conv5x5_bf16_rows(d.N);
    const int tw = d.Win % 16 == 0 ? 16 : 8, ti_n = tw == 16 ? 1 : 2;
    const int tiles = (d.B / ti_n) * (d.Hin / TH) * (d.Win / tw);
    const int ncg = (d.c0 + d.c1 + 63) / 64, nb = Np / ((Np % 128 == 0 && planes != 3 && planes != -2) ? 128 : 64);     // (three pieces / fp16 pieces: 64-column blocks only)
    // split only up to ONE round of blocks (the kernel is one 8-wave block per CU): 512 blocks = two rounds of half-length blocks with
    // atomics and a zeroed destination were slower than 256 whole ones (bf16 train step 12.56 -> 12.36 ms)
    static const int forced = [] { const char* e = getenv("PIVP_BF16_KS_BLOCKS"); return e ? atoi(e) : 0; }();   // tuning
    const int target = forced > 0 ? forced : pivp_cu_count();
    int ks = 1;
    if (d.ksplit_ok && !d.accum)
        while (ks * 2 <= ncg && (long)tiles * nb * ks * 2 <= target) ks *= 2;
    return ks;
}

int conv5x5_bf16(const IgemmDesc& d, const unsigned short* wb, hipStream_t stream, int planes) {
    PIVP_CHECK_ARG(wb && bf16_geometry_ok(d) && d.out && d.N > 0 && d.ldo >= d.N && d.x0 && d.c0 > 0 && ((planes >= 1 && planes <= 3) || planes == -2) &&
                   (planes != 3 || d.Win % 16 == 0) && (planes != -2 || (d.wscale_part && d.c1 == 0)));
    const int Np = conv5x5_bf16_rows(d.N);
    IgemmDesc dd = d;
    dd.N = Np;                                         // the kernel's weight-row count
    const bool wide = Np % 128 == 0;
    const int nb = Np / (wide ? 128 : 64);
    const int ks = conv5x5_bf16_ksplit(d, planes);
    if (planes == 3)     // three pieces (wb packed with planes = 3, plain = 1): 64-column blocks, weights from L2 into the operand registers, eight
        return launch_x6g_plain<3>(dd, wb, stream, Np / 64, ks, d.N);      // waves (the k-step-ring form of it measured 118 us per launch in the sweep against 100)
    if (planes == -2 && d.Win % 16)     // ... on an 8-wide map (an even batch): the ring kernel's two-image tiles, wb packed with plain = 2
        return launch_bf16<16, false, 2, true>(dd, wb, stream, nullptr, Np / 64, ks, d.N);
    if (planes == -2)    // two fp16 pieces (wb packed with planes = -2, plain = 1; d.wscale_part = absmax_partials(d.x0): the activations' scale)
        return launch_x6g_plain<2>(dd, wb, stream, Np / 64, ks, d.N);
    if (planes == 2)     // split mode (wb packed with planes = 2): 128-column blocks run the two-slot schedule, 64-column ones the four-slot one
        return wide ? launch_bf16<32, false, 2>(dd, wb, stream, nullptr, nb, ks, d.N)
                    : launch_bf16<16, false, 2>(dd, wb, stream, nullptr, nb, ks, d.N);
    return wide ? launch_bf16<32, false>(dd, wb, stream, nullptr, nb, ks, d.N)
                : launch_bf16<16, false>(dd, wb, stream, nullptr, nb, ks, d.N);
}

}  // namespace pivp

PIVP_DEFINE_MAIN_PRIO_SETTER(convlstm_bf16)
