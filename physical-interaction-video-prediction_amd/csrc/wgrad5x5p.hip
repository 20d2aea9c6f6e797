// ConvLSTM weight gradient in fp32, round 6 form (backward of BasicConvLSTMCell's 5x5 conv, TM:262-266, under optimizer.update TM:950):
//     dW[tap][ci][n] += sum over pixels m (and timesteps) of  X[m + tap][ci] * dG[m][n]
//
// Same arithmetic unit as wgrad5x5_kernel (igemm_wgrad.hip): a WAVE works alone on 16-pixel chunks of one (kernel row ky, 32 input channels,
// 32 NTW gate columns) tile -- it stages the dG tile [16][32 NTW] and the X strip of kernel row ky with its 2-pixel halo into its OWN LDS buffers and
// runs the 5 taps kx of the row against the same dG fragments on v_mfma_f32_32x32x2_f32 (rows = ci, columns = n, k = pixels); the four waves
// of a block share the tile over interleaved chunks and meet through LDS at the end.  What round 6 changes around that unit:
//   * staging by LDS-DMA (global_load_lds_dwordx4 per 1 KiB piece, issued from inline asm so that hipcc's vmcnt(0)-before-every-LDS-read does
//     not drain the pipeline): no staging registers, no ds_write; NTW = 1: four 5-KiB buffers per wave = three chunks in flight behind a counted
//     vmcnt, NTW = 2 (64 columns per wave, 0.7 x the staged bytes and DMAs per MFMA; 160 accumulator registers fit two waves per SIMD now that the
//     staging registers are gone): two 7-KiB buffers; out-of-image strip pixels read a 16-byte zero word;
//   * the grid is G = 2 blocks per CU, always, and the (tile, chunk) work is dealt to them in equal contiguous ranges (a block whose range crosses
//     a tile boundary runs two or more SEGMENTS, each with its own block reduction): no partially filled last round (the tile x split grid of
//     round 2 filled 480 of 512 slots on every layer of the model), and the blocks of one XCD (linear id mod 8) work on one part of the pixels
//     and one part of the tiles, so that what an XCD streams it streams through its own L2;
//   * no atomics: a segment's result is added (or, for a sweep's first launch, stored) into the segment's own slot of a partial buffer with plain
//     loads issued before the K loop; wgrad5x5p_reduce sums the slots of a tile in a fixed order into the packed gradient.  The partition is a
//     function of one timestep's geometry alone, so every launch of a sweep, the buffer and the reduction agree, and the result is bit-identical
//     from sweep to sweep.  The bias gradient (column sums of dG) rides along in the slots' tails.
// Measured (profiles/r06/NOTES.md 1; scripts/bench_lstm_backward.py, B = 32, us per launch, round-2 kernel -> this one): lstm1/2 117 -> 116, lstm3 89 -> 90,
// lstm4 116 -> 115, lstm5 89 -> 90, lstm6 166 -> 167, lstm7 221 -> 218: 0.746 -> 0.740 of the fp32 MFMA peak over the seven cells -- no faster.  The stamps
// say why: the clock holds 2.38 GHz in both kernels (not the limiter), prologue + epilogue shrank to 2 us and the grid is full, but the K loop itself runs
// 203 us for 171 us of MFMAs on lstm7, and timing-only builds price its parts: without the DMAs 186 us, also without the LDS operand reads 178 -- wherever in
// the chunk the DMAs are issued (between chunks or one per k-step under the MFMAs: 221 vs 218 us).  5 KB staged and 1.2 ds_read2_b32 per 40 / per MFMA is what a
// 32 x 32 x 5-tap wave tile costs; the 64-column tile (NTW = 2, 0.7 x both) needs 256 VGPRs with 45 spilled to keep two waves per SIMD and two LDS buffers:
// 4-17 % slower on every cell.  In the backward sweep the slots cost 0.7 ms per train step (27.3 -> 28.0 ms): 42 MB of slot traffic per launch stream through
// HBM beside the main stream's memory-bound kernels, where the round-2 kernel's atomics hit a gradient that stays in L2.  So the SWEEP keeps the round-2 kernel
// (pivp_plan.hip), and this one is the bit-reproducible form behind pivp_wgrad5x5_f32_batch (part != NULL).
#include <type_traits>

#include "pivp_kernels.h"

#ifdef PIVP_WG_STAMPS
__device__ long long pivp_wgp_stamps[2048 * 8];
#define WGP_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 2048) { \
    pivp_wgp_stamps[blockIdx.x * 4 + (i)] = (long long)wall_clock64(); pivp_wgp_stamps[2048 * 4 + blockIdx.x * 4 + (i)] = (long long)clock64(); } } while (0)
#else
#define WGP_STAMP(i)
#endif

namespace pivp {

namespace {
__device__ float g_wgp_zero[4];      // what out-of-image strip pixels and the strip image's padding lanes load

constexpr int WP_CH = 16;                  // pixels per chunk
constexpr int WP_IT = 32 * 33;             // one 32 x 32 tile image of the block reduction (rows of 33: see wgrad5x5_kernel)
constexpr int WP_IMG = 5 * WP_IT;
constexpr int wp_nbuf(int ntw) { return ntw == 1 ? 3 : 2; }                       // LDS buffers per wave
constexpr int wp_buf(int ntw) { return (2 * ntw + 3) * 256; }                      // floats per buffer: the dG tile's 2 NTW pieces, then the strip's three
constexpr int wp_lds_floats(int ntw) { return 4 * wp_nbuf(ntw) * wp_buf(ntw); }   // 80 KiB / 56 KiB: two blocks per CU
constexpr int wp_slot(int ntw) { return 5 * ntw * 1024 + 64; }                    // floats per segment slot: [t][kx][n][ci] + the bias tail (32 NTW used)
static_assert(2 * WP_IMG + 4 * 64 <= wp_lds_floats(1) && 2 * WP_IMG + 4 * 64 <= wp_lds_floats(2), "the block reduction's images live in the staging buffers");

// The partition: shared by the kernel, the reduction and the host (slot count).  All quantities describe ONE timestep.
struct WpGeom {
    int J;              // blocks per XCD (grid = 8 J)
    int PP, TP;         // pixel parts x tile parts = 8 (one pair per XCD)
    int NTW;            // 32-column MFMA tiles per wave
    int T;              // tiles = 5 * (cin / 32) * (N / (32 NTW)), ordered (nb, cb, ky)
    int cpt;            // 16-pixel chunks per timestep
    int maxseg;         // slots per block
    int logW, logHW;    // the map is a power of two wide and high
};
struct WpRange { int t_lo, nt, c_lo, len; long long i0, i1; };
__host__ __device__ inline WpRange wp_range(const WpGeom& g, int x, int j) {
    WpRange r;
    const int px = x % g.PP, tx = x / g.PP;
    r.t_lo = (int)((long long)g.T * tx / g.TP);
    r.nt = (int)((long long)g.T * (tx + 1) / g.TP) - r.t_lo;
    r.c_lo = (int)((long long)g.cpt * px / g.PP);
    r.len = (int)((long long)g.cpt * (px + 1) / g.PP) - r.c_lo;
    const long long tot = (long long)r.nt * r.len;
    r.i0 = tot * j / g.J; r.i1 = tot * (j + 1) / g.J;
    return r;
}

// the next segment of a block's item range [i, i1): tile (local index tl) and chunk range [k0, k1) of the part; advances i
__host__ __device__ inline void wp_next_segment(const WpRange& r, long long& i, int& tl, int& k0, int& k1) {
    tl = (int)(i / r.len); k0 = (int)(i - (long long)tl * r.len);
    const long long left = r.i1 - i;
    k1 = left < (long long)(r.len - k0) ? k0 + (int)left : r.len;
    i += k1 - k0;
}
// the slots (block * maxseg + segment) that hold partial sums of `tile`, in the reduction's fixed order: pixel parts ascending, blocks ascending
template <class F>
__host__ __device__ inline int wp_for_each_slot(const WpGeom& g, int tile, F f) {
    int tx = 0, n = 0;                   // the tile part that holds this tile
    while (tx + 1 < g.TP && (int)((long long)g.T * (tx + 1) / g.TP) <= tile) ++tx;
    for (int px = 0; px < g.PP; ++px) {
        const int x = tx * g.PP + px;
        const WpRange r0 = wp_range(g, x, 0);
        if (r0.len <= 0 || r0.nt <= 0) continue;
        const int tl = tile - r0.t_lo;
        const long long tot = (long long)r0.nt * r0.len, lo = (long long)tl * r0.len, hi = lo + r0.len;
        int j = (int)(lo * g.J / tot);
        if (j > 0) --j;
        long long b0 = tot * j / g.J;
        for (; j < g.J && b0 < hi; ++j) {
            const long long b1 = tot * (j + 1) / g.J;
            if (b1 > lo && b1 > b0) { f(n, (unsigned)((j * 8 + x) * g.maxseg + (tl - (int)(b0 / r0.len)))); ++n; }
            b0 = b1;
        }
    }
    return n;
}

inline bool wp_ok_shape(const WgradDesc& d) {
    if (d.deconv || d.ksize != 5 || d.pad != 2 || d.stride != 1 || d.Hx != d.Hy || d.Wx != d.Wy) return false;
    if (d.Wg < 8 || (d.Wg & (d.Wg - 1)) || (d.Hg & (d.Hg - 1)) || d.Hg < 2) return false;
    if (d.M % WP_CH || d.N % 32 || d.cin % 32 || d.c0 % 32 || d.c1 % 32 || d.ld0 % 4 || d.ldy % 4 || (d.c1 && d.ld1 % 4)) return false;
    return true;
}
inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
// d.form: 0 = by shape, 1 / 2 = 32 / 64 columns per wave (64 needs N % 64 == 0)
inline int wp_ntw(const WgradDesc& d) {
    if (d.N % 64) return 1;
    if (d.form == 1 || d.form == 2) return d.form;
    return 1;
}
inline WpGeom wp_geom(const WgradDesc& d) {
    WpGeom g;
    g.J = (2 * pivp_cu_count()) / 8;
    if (g.J < 1) g.J = 1;
    g.NTW = wp_ntw(d);
    g.T = 5 * (d.cin / 32) * (d.N / (32 * g.NTW));
    g.cpt = d.M / WP_CH;
    // pixel parts: as many as keep the segment count near the block count (every tile of a part is cut once more per part) and a part's chunk
    // range long enough for four waves
    int pp = 8;
    while (pp > 1 && ((long long)g.T * pp > 8LL * g.J || g.cpt / pp < 16)) pp >>= 1;
    // (r06_c05, all six shapes at B = 32: one part against this rule within +-2 % -- lstm7 222.5 against 218.1 us --, eight parts on the small maps
    // 5-25 % slower: lstm5 112.4 against 90.1)
    while (8 / pp > g.T) pp <<= 1;               // (fewer tiles than tile parts: never with this model's layers)
    g.PP = pp; g.TP = 8 / pp;
    g.logW = ilog2(d.Wg); g.logHW = ilog2(d.Hg * d.Wg);
    int ms = 1;
    for (int x = 0; x < 8; ++x)
        for (int j = 0; j < g.J; ++j) {
            const WpRange r = wp_range(g, x, j);
            if (r.i1 <= r.i0) continue;
            const int segs = (int)((r.i1 - 1) / r.len - r.i0 / r.len) + 1;
            if (segs > ms) ms = segs;
        }
    g.maxseg = ms;
    return g;
}
template <int N> __device__ __forceinline__ void wp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
}  // namespace

// SW: pixels of one image row inside a chunk: 16 (maps at least 16 wide) or 8 (8-wide maps: two rows per chunk); NTW: 32-column tiles per wave
template <int SW, int NTW>
__global__ __launch_bounds__(256, 2) void wgrad5x5p_kernel(const WgradDesc d, const WpGeom g) {
    constexpr int R = WP_CH / SW;                // image rows per chunk
    constexpr int SP = R * (SW + 4);             // strip pixels used (20 or 24)
    constexpr int NY = 2 * NTW, NPIECE = NY + 3; // DMA pieces of a chunk: the dG tile's, then the strip's
    constexpr int NBUF = wp_nbuf(NTW), DEPTH = NBUF - 1, BUF = wp_buf(NTW), YP = 32 * NTW, YF = NY * 256, SLOT = wp_slot(NTW);
    constexpr int YL = 8 * NTW;                  // float4 lanes per dG pixel
    extern __shared__ __attribute__((aligned(16))) float sm[];
    WGP_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int ncb = d.cin >> 5;
    const int Wd = d.Wg, Hd = d.Hg;
    const int tcount = d.tcount > 1 ? d.tcount : 1;
    const int L = blockIdx.x, xcd = L & 7, jb = L >> 3;
    const WpRange rg = wp_range(g, xcd, jb);
    float* const wbase = sm + wave * (NBUF * BUF);
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)wbase));      // LDS byte address of the wave's buffers

    // ---- staging roles: piece p of a chunk = LDS floats [256 p, 256 p + 256) of the buffer ---------------------------------------------------
    // dG pieces p < NY: lane -> (pixel p * (64 / YL) + lane / YL, 4 columns); X pieces: lane -> (strip pixel 8 (p - NY) + lane / 8, 4 channels)
    const int yq = lane / YL, yc4 = (lane % YL) * 4;
    const int q8 = lane >> 3, c4 = (lane & 7) * 4;
    int xrel[3];                                  // the lane's strip pixel relative to the chunk's first pixel (in pixels), kernel row excluded
    unsigned m_pad = 0, m_left = 0, m_right = 0, m_r1 = 0;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int sp = 8 * p + q8;
        const int r = sp / (SW + 4), xx = sp - r * (SW + 4);
        xrel[p] = 0;
        if (sp >= SP) { m_pad |= 1u << p; continue; }
        if (xx < 2) m_left |= 1u << p;
        if (xx >= SW + 2) m_right |= 1u << p;
        if (r) m_r1 |= 1u << p;
        xrel[p] = r * Wd + xx - 2;
    }
    const unsigned m_r0 = ~(m_r1 | m_pad) & 7u;
    const char* const zero = reinterpret_cast<const char*>(g_wgp_zero);

    f32x16 acc[5 * NTW];                          // [t][kx]
    float bsum[NTW];
    long long i = rg.i0;
    for (int seg = 0; i < rg.i1; ++seg) {
        // ---- this segment: tile and chunk range (block-uniform) ---------------------------------------------------------------------------
        int tl, k0, k1;
        wp_next_segment(rg, i, tl, k0, k1);
        const int tile = rg.t_lo + tl;
        const int ky = tile % 5, cb = (tile / 5) % ncb, nb = tile / (5 * ncb);
        const int ci0 = cb * 32, n0 = nb * YP;
        const bool from0 = ci0 < d.c0;               // a 32-channel block never straddles the two sources (c0 % 32 == 0)
        const char* const xbase = reinterpret_cast<const char*>(from0 ? d.x0 : d.x1) + (from0 ? ci0 : ci0 - d.c0) * 4;
        const long long xts = from0 ? d.ts_x0 : d.ts_x1;
        const int xld4 = (from0 ? d.ld0 : d.ld1) * 4;                // bytes per pixel of the X source
        const char* const ybase = reinterpret_cast<const char*>(d.dy) + n0 * 4;
        const int yld4 = d.ldy * 4;
        const int seglen = k1 - k0;
        const int Q = tcount * seglen;               // chunks of the segment, timestep-major
        const int n_my = Q > wave ? (Q - wave + 3) >> 2 : 0;
        float* const slot = d.part + ((size_t)L * g.maxseg + seg) * SLOT;
        const bool bias_tile = d.db != nullptr && ky == 2 && cb == 0;

#pragma unroll
        for (int t = 0; t < 5 * NTW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int t = 0; t < NTW; ++t) bsum[t] = 0.f;

        // ---- the wave's chunk walk: q = wave, wave + 4, ...; (tj, kk) = (timestep, chunk of the segment) ------------------------------------
        // The chunk loop is software-pipelined so that nothing but MFMAs sits between MFMAs' issue slots: while chunk `it` is multiplied, the DMA
        // pieces of chunk it + DEPTH go out behind its first k-steps (NTW pieces and their address arithmetic per k-step, in the shadow of that
        // k-step's MFMAs), the scalar bookkeeping of the chunk after that runs behind k-step 5, the counted wait for chunk it + 1 behind k-step 6,
        // and chunk it + 1's first operands are read behind k-step 7.  (The first version issued, waited and read between two chunks: no faster and
        // no slower, 218 against 221 us on lstm7 -- timing-only builds price the DMAs at 8 % of the loop wherever they are issued, the LDS reads at 4 %.)
#ifdef PIVP_WGP_ABLATE
        bool steady_now = false;
#endif
        int tj_i = 0, kk_i = wave;                   // position of the NEXT chunk to prepare
        while (kk_i >= seglen && tj_i < tcount) { kk_i -= seglen; ++tj_i; }
        const char* yb = nullptr; const char* xb = nullptr;      // prepared chunk: wave-uniform bases, this lane's out-of-image mask
        unsigned oob = 0;
        auto prep = [&]() {
            const int p0 = __builtin_amdgcn_readfirstlane((rg.c_lo + k0 + kk_i) * WP_CH);     // first pixel of the chunk, linear in [0, M)
            const int x0 = p0 & (Wd - 1), y0 = (p0 & ((1 << g.logHW) - 1)) >> g.logW;
            yb = ybase + (long long)tj_i * d.ts_dy + (long long)p0 * yld4;
            xb = xbase + (long long)tj_i * xts + ((long long)p0 + (long long)(ky - 2) * Wd) * xld4;
            const bool row0 = (unsigned)(y0 + ky - 2) < (unsigned)Hd, row1 = (unsigned)(y0 + 1 + ky - 2) < (unsigned)Hd;
            oob = m_pad | (x0 == 0 ? m_left : 0u) | (x0 + SW == Wd ? m_right : 0u) | (row0 ? 0u : m_r0) | (row1 ? 0u : m_r1);
            kk_i += 4;
            while (kk_i >= seglen && tj_i < tcount) { kk_i -= seglen; ++tj_i; }
        };
        auto piece = [&](auto P, int buf) {
            constexpr int p = decltype(P)::value;
            if constexpr (p < NPIECE) {
#if defined(PIVP_WGP_ABLATE) && PIVP_WGP_ABLATE >= 1      // timing-only builds (results are wrong): what the loop costs without its DMAs / LDS reads
                if (steady_now) return;
#endif
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_wave + (unsigned)buf * (BUF * 4) + p * 1024);
                const char* src;
                if constexpr (p < NY) src = yb + (p * (64 / YL) + yq) * yld4 + yc4 * 4;
                else {
                    src = xb + xrel[p - NY] * xld4 + c4 * 4;
                    if ((oob >> (p - NY)) & 1) src = zero;
                }
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            }
        };
        auto pieces_all = [&](int buf) {
            piece(std::integral_constant<int, 0>{}, buf); piece(std::integral_constant<int, 1>{}, buf); piece(std::integral_constant<int, 2>{}, buf);
            piece(std::integral_constant<int, 3>{}, buf); piece(std::integral_constant<int, 4>{}, buf); piece(std::integral_constant<int, 5>{}, buf);
            piece(std::integral_constant<int, 6>{}, buf);
        };
        // (Operands read TWO k-steps ahead of their MFMAs -- four register sets, the wait for the next chunk one k-step earlier -- measured the same,
        // r06_c09: lstm7 219.4 against 219.8 us, lstm5 91.3 against 89.8: the loop does not wait for LDS reads.)
        float av[2][5], bv[2][NTW];
        auto read_step = [&](auto S2, int s, int buf) {
            constexpr int s2 = decltype(S2)::value;
#if defined(PIVP_WGP_ABLATE) && PIVP_WGP_ABLATE >= 2
            if (steady_now) return;
#endif
            constexpr int sidx0 = (2 * s2 / SW) * (SW + 4) + (2 * s2 % SW);      // strip index of (pixel 2 s2, kx = 0)
            const float* ys = wbase + buf * BUF + l31;
            const float* xs = ys + YF;
#pragma unroll
            for (int t = 0; t < NTW; ++t) bv[s][t] = ys[(2 * s2 + half) * YP + 32 * t];
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) av[s][kx] = xs[(sidx0 + half + kx) * 32];
        };
        auto mfmas = [&](int cur) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) bsum[t] += bv[cur][t];
#pragma unroll
            for (int kx = 0; kx < 5; ++kx)
#pragma unroll
                for (int t = 0; t < NTW; ++t)
                    acc[t * 5 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][kx], bv[cur][t], acc[t * 5 + kx], 0, 0, 0);
        };
        // one chunk.  STEADY: chunk it + DEPTH exists (its pieces are issued here) -- then the chunks between exist as well.
        auto chunk = [&](auto STEADY_, int it) {
            constexpr bool STEADY = decltype(STEADY_)::value;
            const int buf = it % NBUF, bufn = (it + DEPTH) % NBUF;
            const bool more = STEADY || it + 1 < n_my;               // chunk it + 1 exists: wait for it and read its first operands at the end
#define WP_PIECES(S) do { if constexpr (STEADY) { piece(std::integral_constant<int, (S) * NTW>{}, bufn); \
                                                  if constexpr (NTW == 2) piece(std::integral_constant<int, (S) * NTW + 1>{}, bufn); } } while (0)
#define WP_KSTEP(S, EXTRA) do { read_step(std::integral_constant<int, (S) + 1>{}, ((S) + 1) & 1, buf); __builtin_amdgcn_sched_barrier(0); \
                                mfmas((S) & 1); EXTRA; __builtin_amdgcn_sched_barrier(0); } while (0)
            WP_KSTEP(0, WP_PIECES(0));
            WP_KSTEP(1, WP_PIECES(1));
            WP_KSTEP(2, WP_PIECES(2));
            WP_KSTEP(3, WP_PIECES(3));
            WP_KSTEP(4, WP_PIECES(4));
            WP_KSTEP(5, if (STEADY && it + DEPTH + 1 < n_my) prep());
#undef WP_KSTEP
            read_step(std::integral_constant<int, 7>{}, 1, buf);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(0);
            if constexpr (STEADY) wp_wait_vm<(DEPTH - 1) * NPIECE>();      // the chunks behind it + 1 may still be in flight
            else if (DEPTH >= 3 && it + 3 == n_my) wp_wait_vm<NPIECE>();      // (three buffers ahead: it + 2 is the last chunk)
            else wp_wait_vm<0>();
            __builtin_amdgcn_sched_barrier(0);
            if (more) read_step(std::integral_constant<int, 0>{}, 0, (it + 1) % NBUF);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(1);
            __builtin_amdgcn_sched_barrier(0);
#undef WP_PIECES
        };
        if (seg == 0) WGP_STAMP(1);
        if (n_my > 0) {
            // prologue: up to DEPTH chunks in flight, the first one landed, its first operands read
            const int npro = n_my < DEPTH ? n_my : DEPTH;
            for (int c = 0; c < npro; ++c) { prep(); pieces_all(c); }
            if (npro == 3) wp_wait_vm<2 * NPIECE>();
            else if (npro == 2) wp_wait_vm<NPIECE>();
            else wp_wait_vm<0>();
            if (n_my > DEPTH) prep();                   // chunk DEPTH: issued during chunk 0
            read_step(std::integral_constant<int, 0>{}, 0, 0);
            int it = 0;
#ifdef PIVP_WGP_ABLATE
            steady_now = true;
#endif
            for (; it + DEPTH < n_my; ++it) chunk(std::true_type{}, it);
#ifdef PIVP_WGP_ABLATE
            steady_now = false;
#endif
            for (; it < n_my; ++it) chunk(std::false_type{}, it);
        }

        // ---- block reduction of the four workers (as wgrad5x5_kernel), 32 columns at a time, then the slot ------------------------------------------
        // (the slot's old contents are requested in front of the reduction and met behind it: plain loads, no atomics)
        __syncthreads();                               // every wave is done with its staging buffers
        float* const bred = sm + 2 * WP_IMG;           // [4 waves][32 NTW columns]: the workers' column sums of dG
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const float v = bsum[t] + __shfl_xor(bsum[t], 32, 64);
            if (half == 0) bred[wave * YP + t * 32 + l31] = v;
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            f32x16* const a5 = acc + t * 5;
            auto put = [&](float* img) {
#pragma unroll
                for (int k = 0; k < 5; ++k)
#pragma unroll
                    for (int r = 0; r < 16; ++r) img[k * WP_IT + l31 * 33 + (r & 3) + 8 * (r >> 2) + 4 * half] = a5[k][r];
            };
            auto add = [&](const float* img) {
#pragma unroll
                for (int k = 0; k < 5; ++k)
#pragma unroll
                    for (int r = 0; r < 16; ++r) a5[k][r] += img[k * WP_IT + l31 * 33 + (r & 3) + 8 * (r >> 2) + 4 * half];
            };
            float old[20];
            float* const sl = slot + t * 5 * 1024;
            if (!d.part_overwrite) {
#pragma unroll
                for (int e = 0; e < 20; ++e) old[e] = sl[e * 256 + tid];
            } else {
#pragma unroll
                for (int e = 0; e < 20; ++e) old[e] = 0.f;
            }
            if (wave >= 2) put(sm + (wave - 2) * WP_IMG);
            __syncthreads();
            if (wave < 2) add(sm + wave * WP_IMG);
            __syncthreads();
            if (wave == 1) put(sm);
            __syncthreads();
            if (wave == 0) add(sm);
            __syncthreads();
            if (wave == 0) put(sm);
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 20; ++e) {
                const int idx = e * 256 + tid, k = idx >> 10, rem = idx & 1023, n = rem >> 5, ci = rem & 31;
                sl[idx] = old[e] + sm[k * WP_IT + n * 33 + ci];
            }
            if (t + 1 < NTW) __syncthreads();          // the next 32 columns' images overwrite these
        }
        if (bias_tile && tid < YP) {
            float* const q = slot + 5 * NTW * 1024 + tid;
            const float v = (bred[tid] + bred[YP + tid]) + (bred[2 * YP + tid] + bred[3 * YP + tid]);
            *q = d.part_overwrite ? v : *q + v;
        }
        __syncthreads();                               // the next segment's DMAs overwrite the images
    }
    WGP_STAMP(2);
#ifdef PIVP_WG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WGP_STAMP(3);
#endif
}

// dw (and db) += the sum of the segment slots of every tile, in a fixed order: pixel parts ascending, blocks ascending.  Block = one (tile, t, kx): 1024
// contiguous elements of the packed gradient [tap][wcin/32][N][32], four per thread.  One thread lists the tile's slots (the partition's arithmetic), then
// every thread loads them four at a time (a thread that walked the list with one dependent load per slot took 13-19 us per cell at B = 32).
__global__ __launch_bounds__(256) void wgrad5x5p_reduce_kernel(const WgradDesc d, const WpGeom g) {
    constexpr int MAXC = 128;                    // slots per tile the list holds (8 pixel parts x the blocks that cut one part's tile range)
    __shared__ unsigned offs[MAXC];              // slot index = block * maxseg + segment
    __shared__ int cnt;
    const int NTW = g.NTW, SLOT = wp_slot(NTW);
    const int tile = blockIdx.x / (5 * NTW), sub = blockIdx.x - tile * 5 * NTW, t = sub / 5, kx = sub - t * 5;
    const int ncb = d.cin >> 5;
    const int ky = tile % 5, cb = (tile / 5) % ncb, nb = tile / (5 * ncb);
    const int tid = threadIdx.x;
    const bool bias = d.db != nullptr && ky == 2 && cb == 0 && kx == 0 && tid < 32;
    const size_t eo = (size_t)(t * 5 + kx) * 1024 + tid * 4, bo = (size_t)5 * NTW * 1024 + t * 32 + tid;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    float sb = 0.f;
    for (int base = 0;; base += MAXC) {          // (one window with this model's layers: a tile has 2..24 slots)
        if (tid == 0) cnt = wp_for_each_slot(g, tile, [&](int n, unsigned slot) { if (n >= base && n < base + MAXC) offs[n - base] = slot; });
        __syncthreads();
        const int total = cnt, n = total - base < MAXC ? total - base : MAXC;
        int c = 0;
        for (; c + 3 < n; c += 4) {
            const float* p0 = d.part + (size_t)offs[c] * SLOT; const float* p1 = d.part + (size_t)offs[c + 1] * SLOT;
            const float* p2 = d.part + (size_t)offs[c + 2] * SLOT; const float* p3 = d.part + (size_t)offs[c + 3] * SLOT;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p0 + eo), v1 = *reinterpret_cast<const f32x4*>(p1 + eo);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(p2 + eo), v3 = *reinterpret_cast<const f32x4*>(p3 + eo);
            if (bias) sb += (p0[bo] + p1[bo]) + (p2[bo] + p3[bo]);
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; c < n; ++c) {
            const float* p0 = d.part + (size_t)offs[c] * SLOT;
            s0 += *reinterpret_cast<const f32x4*>(p0 + eo);
            if (bias) sb += p0[bo];
        }
        if (base + MAXC >= total) break;
        __syncthreads();                         // the list is rewritten
    }
    const int tap = ky * 5 + kx;
    float* o = d.dw + (((size_t)tap * (d.wcin >> 5) + cb) * d.N + (nb * NTW + t) * 32) * 32 + tid * 4;
    f32x4 v = *reinterpret_cast<f32x4*>(o);
    v += (s0 + s1) + (s2 + s3);
    *reinterpret_cast<f32x4*>(o) = v;
    if (bias) d.db[(nb * NTW + t) * 32 + tid] += sb;
}

bool wgrad5x5p_ok(const WgradDesc& d) { return wp_ok_shape(d); }

// The partition as the kernel and the reduction see it, walked on the host (tests/test_host.py: no GPU).  segs: one (block, segment, tile, first chunk, end chunk)
// quintuple per segment in the kernel's order; slots: per tile, the reduction's list as (tile, slot) pairs.  Returns the counts; nothing is written past the caps.
int wgrad5x5p_partition(const WgradDesc& d, int* geom8, int* segs, int seg_cap, int* nsegs, int* slots, int slot_cap, int* nslots) {
    PIVP_CHECK_ARG(wp_ok_shape(d) && geom8 && nsegs && nslots);
    const WpGeom g = wp_geom(d);
    geom8[0] = g.J; geom8[1] = g.PP; geom8[2] = g.TP; geom8[3] = g.NTW; geom8[4] = g.T; geom8[5] = g.cpt; geom8[6] = g.maxseg; geom8[7] = wp_slot(g.NTW);
    int ns = 0;
    for (int L = 0; L < 8 * g.J; ++L) {
        const WpRange rg = wp_range(g, L & 7, L >> 3);
        long long i = rg.i0;
        for (int seg = 0; i < rg.i1; ++seg) {
            int tl, k0, k1;
            wp_next_segment(rg, i, tl, k0, k1);
            if (segs && ns < seg_cap) { int* q = segs + 5 * ns; q[0] = L; q[1] = seg; q[2] = rg.t_lo + tl; q[3] = rg.c_lo + k0; q[4] = rg.c_lo + k1; }
            ++ns;
        }
    }
    *nsegs = ns;
    int nl = 0;
    for (int tile = 0; tile < g.T; ++tile)
        nl += wp_for_each_slot(g, tile, [&](int n, unsigned slot) { if (slots && nl + n < slot_cap) { slots[2 * (nl + n)] = tile; slots[2 * (nl + n) + 1] = (int)slot; } });
    *nslots = nl;
    return PIVP_OK;
}

long long wgrad5x5p_part_floats(const WgradDesc& d) {
    if (!wp_ok_shape(d)) return 0;
    const WpGeom g = wp_geom(d);
    return (long long)8 * g.J * g.maxseg * wp_slot(g.NTW);
}

template <int SW, int NTW>
static int launch_wp(const WgradDesc& d, const WpGeom& g, hipStream_t s) {
    constexpr int lds_bytes = wp_lds_floats(NTW) * 4;
    static PerDeviceOnce once;
    if (pivp_ensure_dyn_lds(once, reinterpret_cast<const void*>(&wgrad5x5p_kernel<SW, NTW>), lds_bytes) != PIVP_OK) return PIVP_ERR_LAUNCH;
    hipLaunchKernelGGL((wgrad5x5p_kernel<SW, NTW>), dim3(8 * g.J), dim3(256), lds_bytes, s, d, g);
    return PIVP_LAUNCH_STATUS();
}

int wgrad5x5p(const WgradDesc& d, hipStream_t s) {
    PIVP_CHECK_ARG(wp_ok_shape(d) && d.x0 && d.dy && d.part && (d.c1 == 0 || d.x1) && d.cin == d.c0 + d.c1 && d.M == d.B * d.Hg * d.Wg);
    const WpGeom g = wp_geom(d);
    if (g.NTW == 2) return d.Wg >= 16 ? launch_wp<16, 2>(d, g, s) : launch_wp<8, 2>(d, g, s);
    return d.Wg >= 16 ? launch_wp<16, 1>(d, g, s) : launch_wp<8, 1>(d, g, s);
}

int wgrad5x5p_reduce(const WgradDesc& d, hipStream_t s) {
    PIVP_CHECK_ARG(wp_ok_shape(d) && d.part && d.dw && d.wcin >= d.cin && d.wcin % 32 == 0);
    const WpGeom g = wp_geom(d);
    hipLaunchKernelGGL(wgrad5x5p_reduce_kernel, dim3(g.T * 5 * g.NTW), dim3(256), 0, s, d, g);
    return PIVP_LAUNCH_STATUS();
}

}  // namespace pivp

#ifdef PIVP_WG_STAMPS
extern "C" int pivp_debug_wgp_stamps(long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pivp_wgp_stamps), sizeof(long long) * n) == hipSuccess ? 0 : -2;
}
#endif
