// The bf16 / split-precision ConvLSTM cell (BASELINE.json config 3 and the fp32-grade split modes): weight packs, the tensors' power-of-two scales,
// and the dispatcher over the two kernel families -- convlstm_bf16_kernel (csrc/convlstm_ring.h) and convlstm_x6g_kernel (csrc/convlstm_l2direct.h).
// Their plain-convolution forms (the data gradient) are instantiated in conv5x5_bf16.hip.  Reference op: BasicConvLSTMCell.__call__, TM:234-276.
#include <string.h>

#include "convlstm_ring.h"
#include "convlstm_l2direct.h"

namespace pivp {

namespace {
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
    m = wave_max(m);
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
}  // namespace

// fp32 K-inner packed weights [25][wcin/32][N][32] -> bf16 [ceil(wcin/64)][25][PL][Np][64], channels past wcin and rows past N zero.
// PL = 1: plane 0 = bf16(w).  PL = 2 (split mode): plane 0 = hi = bf16(w), plane 1 = lo = bf16(w - hi).  PL = 3 (three pieces): plane 1 =
// mid = bf16(w - hi), plane 2 = lo = bf16((w - hi) - mid); both differences are exact in fp32, so hi + mid + lo = w.
__global__ void pack_lstm_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wb, int wcin, int N, int Np, int PL, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
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
    __bf16 h = (__bf16)v;
    if (pl >= 1) { v -= (float)h; h = (__bf16)v; }
    if (pl == 2) { v -= (float)h; h = (__bf16)v; }
    wb[i] = __builtin_bit_cast(unsigned short, h);
}

// ---- several weight preparations in one launch (WeightPrepJob, pivp_kernels.h) ----------------------------------------------------------------
struct WeightPrepTable { int n; int blk0[WEIGHT_PREP_MAX + 1]; WeightPrepJob j[WEIGHT_PREP_MAX]; };
__global__ __launch_bounds__(256) void weight_prep_kernel(const WeightPrepTable t) {
    __shared__ float tile[32][33];
    int k = 0;
    while (k + 1 < t.n && (int)blockIdx.x >= t.blk0[k + 1]) ++k;        // block-uniform: which job this block belongs to
    const WeightPrepJob& job = t.j[k];
    const int blk = (int)blockIdx.x - t.blk0[k];
    if (job.kind == 0) {      // repack_transpose_kernel's tile: W packed [tap][cin/32][N][32] -> Wt packed [tap'][N/32][cin][32]
        const int taps = job.p0, cin = job.p1, N = job.p2, flip = job.p3;
        const int nn = N >> 5, nc_ = cin >> 5;
        const int nc = blk % nn, cc = (blk / nn) % nc_, tap = blk / (nn * nc_);
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        const float* src = job.src + (((size_t)tap * nc_ + cc) * N + nc * 32) * 32;
        for (int r = ty; r < 32; r += 8) tile[r][tx] = src[r * 32 + tx];
        __syncthreads();
        const int tap2 = flip ? taps - 1 - tap : tap;
        float* dst = reinterpret_cast<float*>(job.dst) + (((size_t)tap2 * nn + nc) * cin + cc * 32) * 32;
        for (int r = ty; r < 32; r += 8) dst[r * 32 + tx] = tile[tx][r];
        return;
    }
    // one bf16 plane [ceil(wcin/64)][25][Np][64] of pack_lstm_bf16_kernel
    const int wcin = job.p0, N = job.p1, Np = job.p2;
    const long total = (long)((wcin + 63) / 64) * 25 * Np * 64;
    const long i = (long)blk * 256 + threadIdx.x;
    if (i >= total) return;
    const int c64 = (int)(i & 63);
    long r = i >> 6;
    const int n = (int)(r % Np); r /= Np;
    const int tap = (int)(r % 25);
    const int ch = (int)(r / 25) * 64 + c64;
    float v = 0.f;
    if (ch < wcin && n < N) v = job.src[(((long)tap * (wcin >> 5) + (ch >> 5)) * N + n) * 32 + (ch & 31)];
    reinterpret_cast<unsigned short*>(job.dst)[i] = __builtin_bit_cast(unsigned short, (__bf16)v);
}
int weight_prep_batch(const WeightPrepJob* jobs, int n, hipStream_t s) {
    PIVP_CHECK_ARG(jobs && n > 0 && n <= WEIGHT_PREP_MAX);
    WeightPrepTable t;
    memset(&t, 0, sizeof(t));
    t.n = n;
    long blocks = 0;
    for (int k = 0; k < n; ++k) {
        const WeightPrepJob& j = jobs[k];
        PIVP_CHECK_ARG(j.src && j.dst && (j.kind == 0 || j.kind == 1));
        t.blk0[k] = (int)blocks;
        if (j.kind == 0) {
            PIVP_CHECK_ARG(j.p0 > 0 && j.p1 > 0 && j.p1 % 32 == 0 && j.p2 > 0 && j.p2 % 32 == 0);
            blocks += (long)j.p0 * (j.p1 / 32) * (j.p2 / 32);
        } else {
            PIVP_CHECK_ARG(j.p0 > 0 && j.p0 % 32 == 0 && j.p1 > 0 && j.p2 >= j.p1);
            blocks += ((long)lstm_bf16_weight_elems(j.p0, j.p2) + 255) / 256;
        }
        t.j[k] = j;
    }
    t.blk0[n] = (int)blocks;
    PIVP_CHECK_ARG(blocks > 0 && blocks < (1L << 31));
    hipLaunchKernelGGL(weight_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, s, t);
    return PIVP_LAUNCH_STATUS();
}

// PL = 3 pack, FRAGMENT-MAJOR: [group][tap][k-step][plane][8-channel group c8][lane][8 bf16] -- one B fragment of
// v_mfma_f32_32x32x16_bf16 whose 32 columns are the four gates of eight channels (lane (half, l31): gate l31 / 8, channel c8 * 8 + l31 % 8;
// k = the k-step's channels half * 8 .. + 8) is 1 KB in lane order: one global_load_lds_dwordx4 of a wave moves exactly one fragment into
// a lane-linear (conflict-free) kilobyte of the ring, one global_load_dwordx4 of a wave loads it straight into the MFMA's operand registers.
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
    PIVP_CHECK_ARG(plain == 0 || plain == 1);      // a layout selector, not a flag: a future third layout must not be read as "plain"
    const int pieces = planes == -2 ? 2 : planes;
    const long total = (long)lstm_bf16_weight_elems(wcin, Np) * pieces;
    if (planes == -2)        // the tensor's scale first: 64 partial maxima into the pack's tail
        hipLaunchKernelGGL(absmax_partials_kernel, dim3(64), dim3(512), 0, s, w, (long)25 * wcin * N, reinterpret_cast<float*>(wb + total));
    if (planes == 3 || planes == -2) {       // the three-piece kernels' fragment-major pack: the cell's (N = 4 C, gate-interleaved fragments) or a plain conv's
        PIVP_CHECK_ARG(Np % 64 == 0 && (plain || (Np == N && N % 32 == 0)));
        const long nthreads = (long)lstm_bf16_weight_elems(wcin, Np) / 8;
        hipLaunchKernelGGL(pack_lstm_x6_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s, w, wb, wcin, N, Np, plain, pieces, nthreads);
        return PIVP_LAUNCH_STATUS();
    }
    hipLaunchKernelGGL(pack_lstm_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, wb, wcin, N, Np, planes, total);
    return PIVP_LAUNCH_STATUS();
}

bool convlstm_bf16_ok(const IgemmDesc& d) { return bf16_geometry_ok(d) && d.C > 0 && d.C % 16 == 0; }
bool convlstm_bf16x6_ok(const IgemmDesc& d) { return convlstm_bf16_ok(d) && d.Win % 16 == 0; }

// d as for igemm_lstm (validated by the caller's igemm_validate(d, true) equivalent); wb = pack_lstm_bf16(d.w, ..., planes).
int convlstm_bf16(const IgemmDesc& d, const unsigned short* wb, hipStream_t stream, int* ln_nparts, int nch, int planes) {
    PIVP_CHECK_ARG(wb && convlstm_bf16_ok(d) && (nch == 0 || nch == 16 || (nch == 32 && d.C % 32 == 0)) && ((planes >= 1 && planes <= 3) || planes == -2));
    // LayerNorm-on-load (d.in_g) exists in the L2-direct kernels only (launch_x6g): every other form would consume the raw tensor
    PIVP_CHECK_ARG(!d.in_g || ((planes == 3 || planes == -2) && convlstm_bf16x6_ok(d)));
    if (planes == 3 && !convlstm_bf16x6_ok(d)) {    // three pieces on an 8-wide map (an even batch: convlstm_bf16_ok): the L2-direct kernel on tiles of two images
        const long b32 = d.C % 32 ? 0 : (long)(d.B / 2) * (d.Hin / TH) * (d.Win / 8) * (d.C / 32);
        if ((nch == 32 && b32 > 0) || (nch == 0 && b32 >= pivp_cu_count())) return launch_x6g_w8<2, 4>(d, wb, stream, ln_nparts);
        return launch_x6g_w8<4, 2>(d, wb, stream, ln_nparts);
    }
    if (planes == -2 && !convlstm_bf16x6_ok(d)) {   // two fp16 pieces on an 8-wide map (an even batch): the L2-direct kernel on tiles of two images
        const long b32 = d.C % 32 ? 0 : (long)(d.B / 2) * (d.Hin / TH) * (d.Win / 8) * (d.C / 32);
        if ((nch == 32 && b32 > 0) || (nch == 0 && b32 >= pivp_cu_count())) return launch_x6g_w8<2, 4, 2>(d, wb, stream, ln_nparts);
        return launch_x6g_w8<4, 2, 2>(d, wb, stream, ln_nparts);
    }
    if (planes == -2) {   // two fp16 pieces, three MFMAs per product (wb = pack_lstm_bf16(..., planes = -2)): the L2-direct kernel, 16-wide tiles
        PIVP_CHECK_ARG(convlstm_bf16x6_ok(d));
        const long b32 = d.C % 32 ? 0 : (long)d.B * (d.Hin / TH) * (d.Win / 16) * (d.C / 32);
        if ((nch == 32 && b32 > 0) || (nch == 0 && b32 >= pivp_cu_count())) return launch_x6g<2, 4, 2>(d, wb, stream, ln_nparts);
        return launch_x6g<4, 2, 2>(d, wb, stream, ln_nparts);
    }
    if (planes == 3) {   // three pieces, 16-wide tiles
        // weights straight from L2: 32-channel blocks where they give every CU a block, else 16-channel blocks
        const long b32 = d.C % 32 ? 0 : (long)d.B * (d.Hin / TH) * (d.Win / 16) * (d.C / 32);
        if ((nch == 32 && b32 > 0) || (nch == 0 && b32 >= pivp_cu_count())) return launch_x6g<2, 4>(d, wb, stream, ln_nparts);
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

}  // namespace pivp

PIVP_DEFINE_MAIN_PRIO_SETTER(convlstm_bf16)
