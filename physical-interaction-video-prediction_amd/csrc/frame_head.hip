// frame_head: the output side of one timestep in ONE launch (round 4).
//   masks = relu(Deconv1x1(enc6)) and enc7 = Deconv1x1(enc6) with the variant's activation     (TM:718-719, TM:315-317 / 454-455 / 387-388)
//   on enc6 = relu(LayerNorm(raw enc6)) (norm_enc6, TM:601), applied while the tile is staged,
//   the motion head's finisher: CDNA kernels = normalise(relu(Linear(hidden5) - eps) + eps) (TM:326-329) or the STP parameters
//   (TM:458-468), from the K-slice partial sums of skinny_linear_partials_kernel (heads.hip),
//   and the flat-(num_masks+1) softmax + transform + compositing of the next frame              (TM:720-728 with TM:341-349 / 469-470 / 392-415).
// Before: heads_1x1_kernel -> skinny_linear_partials_kernel -> *_finish_kernel -> composite_kernel, four launches and a round trip of the
// 14 logit / enc7 / layer0 planes through HBM per timestep (425 us of a config-2 rollout).  Now the partial sums are the only launch in
// front of this one.  Round 6: the finisher is per-SAMPLE work that every band of a sample repeated (128 KB of partial sums per block); the
// plan now runs it as B "rider" blocks of enc5's launch (deconv_tile.hip) and this kernel reads the sample's kernels / parameters from a.aux
// with its first loads.  The in-kernel finisher (a.partials != NULL) remains for callers without that launch in front (PIVP_FINISH_RIDER=0,
// the per-op API).  Round 6 also took the block's chain from 23 to 12.7 us: profiles/r06/NOTES.md 5.
//
// Block = one sample x FH_TR image rows.  The flat softmax groups of a band reach NP - 1 elements past either end of the band IN THE
// FLAT [plane][pixel] ORDER (TM:720-722 reshapes the NCHW tensor to (-1, NP)): for plane m these are the NP - 1 pixels in front of /
// behind the band, and at the first / last band of a sample the last pixels of plane m - 1 / the first pixels of plane m + 1.  So the
// block computes the 1x1 mixes for its band plus a "halo tile" of 2 (NP - 1) wrapped pixels (<= 22 pixels of 64-channel recompute per
// 256-pixel band) and files each halo logit under the plane whose window it belongs to.  The arithmetic of every output is that of the
// separate kernels, in the same order: the fused frame is BIT-IDENTICAL to heads_1x1 + cdna_kernels / stp_params + composite
// (tests/test_gpu_ops.py::test_frame_head_matches_the_separate_kernels).
//
// Each WAVE owns 64 consecutive pixels per tile, loads them with coalesced 16-B loads, applies the LayerNorm, and turns "lane = 4
// channels of a pixel" into "lane = pixel" through a private LDS tile -- 32 channels at a time (two passes), so that the wave tiles
// are 37 KB instead of 70 and two blocks share a CU.  The blend phase reuses the tiles for the blended CDNA kernels (MFMA outputs back to
// "lane = pixel"); the group maxima live in the (dead) halo rows.
#include "pivp_kernels.h"

namespace pivp {

constexpr int FH_TR = 4;              // image rows per block (composite_kernel: 4 rows 12.2 us, 8 rows 14.3, 2 rows 14.2 at 64 x 64)
constexpr int FH_XP = 36;             // floats per pixel row of the transposition tile (144 B: conflict-free per-lane ds_read_b128)
constexpr int FH_TILE = 64 * FH_XP;   // floats per wave tile
constexpr int FH_KL = 11 * 28;        // CDNA kernel table [11][28]
constexpr int FH_MAXKS = 128;         // most K slices the in-kernel finisher takes (every block of a sample re-reads its KS x 1 KB of partial sums)
constexpr int FH_HP = 68;             // floats per halo row (272 B)
constexpr int FH_WP = 68;             // floats per row of the weight table [32 outputs][64 k]: per-lane rows (MFMA B operand, halo dots) read conflict-free
constexpr int FH_HW = 4;              // waves per block: at 64-wide frames a band is four 64-pixel tiles + the halo tile (wave 0 takes it second)
constexpr int FH_NT = 64 * FH_HW;     // 256 threads x 256 registers: two blocks per CU wherever the dispatcher puts their waves

#ifdef PIVP_FH_STAMPS   // phase stamps (constant-rate 100 MHz counter) of block (3, 5): [wave][slot]; scripts/r04/fh_stamps.py
__device__ long long pivp_fh_stamps[8 * 16];
#define FH_STAMP(i) do { if (blockIdx.x == 3 && blockIdx.y == 5 && (threadIdx.x & 63) == 0) pivp_fh_stamps[(threadIdx.x >> 6) * 16 + (i)] = (long long)wall_clock64(); } while (0)
#else
#define FH_STAMP(i)
#endif

// Block shape, as measured (scripts/r04/fh_stamps.py, profiles/r04/NOTES.md): the first versions ran five heads waves + a side wave for the
// finisher at 168 registers (three waves per SIMD): ONE such block was resident per CU -- 62 us per launch for 30 us per block -- and
// every spill reload inside the multiply loop waited, through its vmcnt(0), for the whole prefetch in flight (pass 0: 6.3 us, pass 1
// with no prefetch behind it: 2.1).  Four waves with 256 registers each: no spills, and two blocks fit a CU whatever SIMDs their waves get.
// NMC / WC (round 6): num_masks and the frame width as compile-time constants (0 = from the arguments) for the geometry of the reference's configurations
// (10 masks -- DNA: 1 --, 64-wide frames).  A block's waves run alone on their SIMDs (two waves per SIMD at most), so its time is its INSTRUCTION COUNT
// at ~4.5 cycles each: with NP, W, G, win as run-time values the plane / output loops carried a predicate, an index clamp and SGPR spills per
// iteration, divisions by W and G were 40-instruction sequences, and the tile epilogue alone was 1,000 instructions.
template <int MODE, int NMC, int WC>   // MODE: 0 CDNA, 1 STP, 2 DNA
__global__ __launch_bounds__(FH_NT, 2) void frame_head_kernel(const FrameHeadArgs a) {
    constexpr int NE = MODE == 2 ? 25 : 3;
    constexpr int MAXO = MODE == 2 ? 2 + 25 : 12 + 3;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = WC ? WC : a.W, HW = H * W, NM = NMC ? NMC : a.NM, NP = NM + 1, NO = NP + NE;
    const int b = blockIdx.y, y0 = blockIdx.x * FH_TR;
    const int p0 = y0 * W, np = FH_TR * W;
    const int hn = NP - 1;
    const int win = np + 2 * hn;
    const int G = np / NP + 2;
    const int PW = W + 4;
    // ---- LDS carve -----------------------------------------------------------------------------------------------------------
    float* wl = sm;                              // [32][FH_WP] weights (rows past NO: zeros) + [32] bias
    float* bl = wl + 32 * FH_WP;
    float* lg = bl + 32;                         // [NP][win] mask logits of the band's windows
    float* l0t = lg + ((NP * win + 3) & ~3);     // [3][np] sigmoid(enc7) (CDNA / STP)
    float* kl = l0t + (MODE == 2 ? 0 : 3 * np);  // [11][28] CDNA kernels
    float* vs = kl + FH_KL;                      // [256] finisher scratch
    float* th = vs + 256;                        // [8] STP theta
    float* prevt = th + 8;                       // [3][FH_TR + 4][PW] previous-frame tile with its 2-pixel halo
    float* hal = prevt + ((3 * (FH_TR + 4) * PW + 3) & ~3);   // [2 (NP - 1)][FH_HP] normalised enc6 rows of the halo pixels
    float* un = hal + max(24 * FH_HP, (2 * NP * G + 3) & ~3);                // FH_HW wave tiles (transposition of the heads' operands / outputs, the finisher's chain sums, the blended kernels)
    float* gmx = hal;                            // (gmx, ginv) reuse the halo rows, dead behind barrier (2): the wave tiles serve the blend phase once more
    float* ginv = gmx + NP * G;
    const unsigned magic = 0xFFFFFFFFu / (unsigned)NP + 1u;     // exact x / NP for x * NP < 2^32
    auto div_np = [&](int x) { return (int)__umulhi((unsigned)x, magic); };
    const float* pb = a.prev + (size_t)b * 3 * HW;
    const int ntile = np / 64;                   // band tiles (the halo pixels are handled by all threads, below)
    FH_STAMP(0);

    // ---- heads: the first pass's loads (e6, gamma, beta: 24 float4 per lane) go out first of all --------------------------------------
    const float* eb = a.e6raw + (size_t)b * HW * 64;
    float* yb = a.y_out ? a.y_out + (size_t)b * HW * 64 : nullptr;
    float* mt = un + wave * FH_TILE;
    auto pix_of = [&](int t, int pl) -> int { return p0 + t * 64 + pl; };       // pixel (inside the sample) of slot pl of band tile t
    f32x4 rx[8], rg[8], rb[8];
    auto issue = [&](int t, int hh) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f = lane + 64 * j;
            const size_t gi = (size_t)pix_of(t, f >> 3) * 64 + hh * 32 + (f & 7) * 4;
            rx[j] = *reinterpret_cast<const f32x4*>(eb + gi);
            rg[j] = *reinterpret_cast<const f32x4*>(a.gamma + gi);
            rb[j] = *reinterpret_cast<const f32x4*>(a.beta + gi);
        }
    };
    // ---- weights of the 1x1 mixes, the previous-frame tile and the halo pixels' rows: every thread's loads go out before its first LDS store
    // (a loop of lds[i] = g[i] compiles to one L2 round trip per element: an early version of this kernel spent 26 of them on the frame
    // tile), and in front of the first pass's 24 loads per lane (loads return in order: behind them the table waited for HBM) ----------
    // halo slot hs: [0, hn) = the hn pixels in front of the band, [hn, 2 hn) = behind it, wrapped into the neighbouring plane's rows
    auto halo_pix = [&](int hs) -> int {
        const int raw = hs < hn ? p0 - hn + hs : p0 + np + (hs - hn);
        return raw < 0 ? raw + HW : raw >= HW ? raw - HW : raw;
    };
    const f32x4 ln_first = ln_partial_first(a.ln_part, b, a.ln_nparts);      // merged behind the staging below
    // the sample's CDNA kernels / STP parameters when a finisher in front of this launch left them in a.aux (the rider blocks of enc5's launch, or
    // the separate finish kernels): requested with the first loads, filed with the weight table
    float kv0[2] = {0.f, 0.f};
    if (MODE != 2 && !a.partials) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {                    // 308 table entries
                const int i2 = min(tid + FH_NT * u, FH_KL - 1), k = i2 / 28, e = i2 - k * 28;
                kv0[u] = a.aux[((size_t)b * NM + min(k, NM - 1)) * 25 + min(e, 24)];
            }
        } else {
            kv0[0] = a.aux[(size_t)b * 6 + min(tid, 5)];
        }
    }
    constexpr int WIT = 32 / FH_HW;              // thread = (k = lane, outputs wave, wave + 4, ...): no run-time division
    float wv[WIT];
#pragma unroll
    for (int u = 0; u < WIT; ++u) {
        const int o = min(wave + FH_HW * u, NO - 1);
        wv[u] = o < NP ? a.wm[lane * NP + o] : a.we[lane * NE + (o - NP)];
    }
    const float bv = tid < NO ? (tid < NP ? a.bm[tid] : a.be[tid - NP]) : 0.f;
    f32x4 hx[2], hg[2], hb[2];                   // halo rows: 2 hn pixels x 16 float4 <= 352 pieces, two per thread
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = min(tid + FH_NT * u, 2 * hn * 16 - 1);
        const size_t gi = (size_t)halo_pix(i >> 4) * 64 + (i & 15) * 4;
        hx[u] = *reinterpret_cast<const f32x4*>(eb + gi);
        hg[u] = *reinterpret_cast<const f32x4*>(a.gamma + gi);
        hb[u] = *reinterpret_cast<const f32x4*>(a.beta + gi);
    }
    constexpr int PR = (FH_TR + 4 + 2) / 3;      // frame-tile rows per thread when 256 / PW >= 3 (W <= 81), the rest looped below
    float tp[3][PR];
    const int tx = tid % PW, tr0 = tid / PW, rstep = FH_NT / PW;
    const bool prow = MODE != 1 && tid < rstep * PW;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int u = 0; u < PR; ++u) {
            const int r = tr0 + u * rstep;
            const int iy = min(max(y0 + r - 2, 0), H - 1), ix = min(max(tx - 2, 0), W - 1);
            tp[c][u] = pb[(size_t)c * HW + iy * W + ix];           // clamped address, zeroed below: no predicate around the load
        }
    if (wave < ntile) issue(wave, 0);
    // the statistics merge here, under the round trip of everything requested above (its one partial per lane was requested first): three wave
    // sums, two divisions and a square root that sat between the staging stores and barrier (1) before (1.1 us of the block's chain)
    float mean, rstd;
    ln_merge_partials(ln_first, a.ln_part, b, a.ln_nparts, a.eps, mean, rstd);
    if (prow) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int u = 0; u < PR; ++u) {
                const int r = tr0 + u * rstep, iy = y0 + r - 2, ix = tx - 2;
                if (r < FH_TR + 4) prevt[(c * (FH_TR + 4) + r) * PW + tx] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? tp[c][u] : 0.f;
            }
        for (int r = tr0 + PR * rstep; r < FH_TR + 4; r += rstep) {   // wide frames: remaining rows
            const int iy = y0 + r - 2, ix = tx - 2;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                prevt[(c * (FH_TR + 4) + r) * PW + tx] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? pb[(size_t)c * HW + iy * W + ix] : 0.f;
        }
    }
#pragma unroll
    for (int u = 0; u < WIT; ++u) {
        const int o = wave + FH_HW * u;
        wl[o * FH_WP + lane] = o < NO ? wv[u] : 0.f;       // rows NO .. 31: the unused columns of the 32-wide MFMA tile
    }
    if (tid < 32) bl[tid] = bv;
    if (MODE == 0) {                                          // (with the finisher in this kernel: zeros now, the kernels behind barrier (2) --
#pragma unroll                                                // the blend's MFMAs read all 10 rows)
        for (int u = 0; u < 2; ++u) {
            const int i2 = tid + FH_NT * u, k = i2 / 28, e = i2 - k * 28;
            if (i2 < FH_KL) kl[i2] = (!a.partials && k < NM && e < 25) ? kv0[u] : 0.f;
        }
    } else if (MODE == 1 && !a.partials && tid < 6) {
        th[tid] = kv0[0];
    }
    FH_STAMP(1);
    if (a.stat_out && blockIdx.x == 0 && tid == 0) { a.stat_out[b * 2] = mean; a.stat_out[b * 2 + 1] = rstd; }
    FH_STAMP(2);
    // the halo pixels' rows, normalised, into LDS: relu((v - mean) rstd gamma + beta), the band tiles' expression
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + FH_NT * u;
        if (i < 2 * hn * 16) {
            f32x4 v = hx[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf((v[e] - mean) * rstd * hg[u][e] + hb[u][e], 0.f);
            *reinterpret_cast<f32x4*>(hal + (i >> 4) * FH_HP + (i & 15) * 4) = v;
        }
    }
    __syncthreads();          // (1) weight table and halo rows in place
    FH_STAMP(3);
    // halo logits: thread = (halo pixel, mask plane), a 64-long dot product in the band tiles' order (bias first, k ascending).  A halo logit
    // of plane o at window position jx belongs to the window of plane o - carry, carry = -1 / +1 where the flat index wrapped into the
    // previous / next plane.  (Done here, while the first pass's e6 is still on its way from HBM.)
    for (int i = tid; i < 2 * hn * NP; i += FH_NT) {
        const int hs = i / NP, o = i - hs * NP;
        const float* xrow = hal + hs * FH_HP;
        const float* wo = wl + o * FH_WP;
        float sacc = bl[o];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xrow + q * 4);
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wo + q * 4);
            sacc = fmaf(xv[0], w4[0], sacc); sacc = fmaf(xv[1], w4[1], sacc); sacc = fmaf(xv[2], w4[2], sacc); sacc = fmaf(xv[3], w4[3], sacc);
        }
        const int raw = hs < hn ? p0 - hn + hs : p0 + np + (hs - hn);
        const int row = o + (raw < 0 ? 1 : raw >= HW ? -1 : 0);
        const int jx = hs < hn ? hs : np + hs;
        if ((unsigned)row < (unsigned)NP) lg[row * win + jx] = fmaxf(sacc, 0.f);
    }

    // ---- heads: one 64-pixel tile per wave and turn, 32 channels per pass; the loads of pass i + 1 are in flight while pass i is multiplied ----
    // The 1x1 mixes are the dense contractions of this model's heads (BASELINE.json north_star): they run on the fp32 matrix cores.
    // v_mfma_f32_16x16x4_f32 (round 6; rounds 4-5: 32x32x2, whose 32 columns were half empty -- 2 x 2,048 cycles of the matrix pipe per tile, shared
    // with the other resident block's wave on the SIMD: 3.5 us of the block's chain) is a k-ordered fmaf chain: C first, then k = 0 .. 3 of lane
    // groups 0 .. 3.  With the bias as C and instruction s fed channels 4s .. 4s + 3 an output is BIT-identical to heads_1x1_kernel's scalar chain.
    // Tile: 16 pixels x 16 outputs (CDNA / STP: NO <= 15 of them real, one column tile; DNA: 27, two); operands by ds_read_b32 (lane = row i, k group
    // g: bank 36 i + g, conflict-free).
    {
        constexpr int NT = MODE == 2 ? 2 : 1;
        f32x4 macc[NT][4];
        const int li = lane & 15, lg4 = lane >> 4;
        const float* arow = mt + li * FH_XP + lg4;
        const float* brow = wl + li * FH_WP + lg4;
        for (int t = wave; t < ntile; t += FH_HW) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                // LayerNorm + ReLU of the pass in registers, into the wave's transposition tile
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int f = lane + 64 * j;
                    f32x4 v = rx[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf((v[e] - mean) * rstd * rg[j][e] + rb[j][e], 0.f);
                    if (yb) *reinterpret_cast<f32x4*>(yb + (size_t)pix_of(t, f >> 3) * 64 + hh * 32 + (f & 7) * 4) = v;
                    *reinterpret_cast<f32x4*>(mt + (f >> 3) * FH_XP + (f & 7) * 4) = v;
                }
                FH_STAMP(4 + 2 * hh);
                // the next pass's loads: the other half of this tile, or the first half of the wave's next tile
                if (hh == 0) issue(t, 1);
                else if (t + FH_HW < ntile) issue(t + FH_HW, 0);
                if (hh == 0) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float bias = bl[16 * nt + li];
#pragma unroll
                        for (int m4 = 0; m4 < 4; ++m4) macc[nt][m4] = f32x4{bias, bias, bias, bias};
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {                // instruction q: channels 4q + (0 .. 3 by lane group) of the pass
                    float bq[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bq[nt] = brow[16 * nt * FH_WP + hh * 32 + 4 * q];
#pragma unroll
                    for (int m4 = 0; m4 < 4; ++m4) {
                        const float av = arow[16 * m4 * FH_XP + 4 * q];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) macc[nt][m4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bq[nt], macc[nt][m4], 0, 0, 0);
                    }
                }
                FH_STAMP(5 + 2 * hh);
            }
            // accumulator register r of M tile m4 = pixel 16 m4 + 4 (lane >> 4) + r of the tile, output column = 16 nt + (lane & 15): back through the
            // wave's (dead) transposition tile to "lane = pixel", so that every lane finishes one pixel's outputs (with the outputs left on their
            // columns, 3 lanes did all of enc7's sigmoids and scattered stores: 4.3 us against 1.3)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int m4 = 0; m4 < 4; ++m4)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mt[(16 * m4 + 4 * lg4 + r) * FH_XP + 16 * nt + li] = macc[nt][m4][r];
            float acc[MAXO];
#pragma unroll
            for (int q = 0; q < (MAXO + 3) / 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(mt + lane * FH_XP + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) if (q * 4 + e < MAXO) acc[q * 4 + e] = v[e];
            }
            {
                const int pp = t * 64 + lane, p = p0 + pp;
#pragma unroll
                for (int o = 0; o < MAXO; ++o) {
                    if (o < NP) {
                        const float v = fmaxf(acc[o], 0.f);
                        lg[o * win + pp + hn] = v;
                        if (a.logits_out) a.logits_out[((size_t)b * NP + o) * HW + p] = v;
                    } else if (o < NO) {
                        const int oe = o - NP;
                        float v = acc[o];
                        if (MODE != 1) v = fmaxf(v, 0.f);
                        a.enc7[((size_t)b * NE + oe) * HW + p] = v;
                        if (MODE != 2) {
                            const float sg = sigmoidf_(v);
                            l0t[oe * np + pp] = sg;
                            if (a.layer0_out) a.layer0_out[((size_t)b * NE + oe) * HW + p] = sg;
                        }
                    }
                }
            }
        }
    }
    FH_STAMP(8);

    // ---- the motion head's finisher, part 1 (fixed summation order: that of sum_partials in heads.hip: four chains, chain c = slices c,
    // c + 4, ... in increasing order, joined as (c0 + c1) + (c2 + c3)).  Wave c sums chain c for all 256 outputs -- a lane owns 4
    // consecutive ones, its loads are whole 16-B pieces of the 1-KB partial rows, the chain's 32 slices in flight at once -- and leaves the
    // sums at the end of its own (dead) transposition tile.  (One wave for all four chains: 8 us behind its tile; scripts/r04/fh_stamps.py.)
    double* chd = reinterpret_cast<double*>(un + wave * FH_TILE + (FH_TILE - 512));
    // (Requested in front of the wave's last multiplies instead, the 128 registers stayed allocated across the whole tile loop: 337 spills.)
    if (MODE != 2 && a.partials) {
        const int KS = a.KS;
        const float* crow = a.partials + ((size_t)b * KS + wave) * 256 + lane * 4;
        f32x4 cv[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) cv[u] = *reinterpret_cast<const f32x4*>(crow + (size_t)u * 1024);   // slices past KS: the buffer's tail padding, not summed
        double c4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 32; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) c4[e] += wave + 4 * u < KS ? (double)cv[u][e] : 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) chd[lane * 4 + e] = c4[e];
    }
    __syncthreads();          // (2) logits, layer0, chain sums (LDS) and, for DNA, enc7 (global, this block's own stores) complete; the tiles are dead
    FH_STAMP(9);

    // ---- per-group max and 1 / sum (groups of NP consecutive flat elements, TM:720-722); the transposition tiles are dead -------------
    for (int i = tid; i < NP * G; i += FH_NT) {
        const int m = i / G, gi = i - m * G;
        const int gfirst = div_np(m * HW + p0);
        const int glast = div_np(m * HW + p0 + np - 1);
        if (gfirst + gi <= glast) {
            const int j0 = (gfirst + gi) * NP - (m * HW + p0 - hn);
            const float* e = lg + m * win + j0;
            float ev[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) { const float t = e[min(u, NP - 1)]; ev[u] = u < NP ? t : -3.0e38f; }   // clamped, not predicated: the 12 LDS reads go out together
            float mx = ev[0];
#pragma unroll
            for (int u = 1; u < 12; ++u) mx = fmaxf(mx, ev[u]);
            float sum = 0.f;
#pragma unroll
            for (int u = 0; u < 12; ++u) sum += u < NP ? __expf(ev[u] - mx) : 0.f;
            gmx[i] = mx;
            ginv[i] = 1.0f / sum;
        }
    }
    FH_STAMP(12);
    // finisher, part 2: thread = output; join the four chains, bias, activation
    if (MODE != 2) {
        const int o = tid;
        if (a.partials) {
            const double* c0 = reinterpret_cast<const double*>(un + (FH_TILE - 512));
            constexpr int CS = FH_TILE / 2;                  // doubles between two waves' chain sums
            const double sv = (c0[o] + c0[CS + o]) + (c0[2 * CS + o] + c0[3 * CS + o]);
            if (MODE == 0) {
                float accv = 0.f;
                if (o < 25 * NM) {
                    const double t = (double)a.hbias[o] + sv;
                    if (a.vpre_out && blockIdx.x == 0) a.vpre_out[(size_t)b * 256 + o] = (float)t;
                    accv = fmaxf((float)t - 1e-12f, 0.f) + 1e-12f;
                }
                vs[o] = accv;
            } else if (o < 100) {
                const double t = (double)a.hbias[o] + sv;
                vs[o] = fmaxf((float)t, 0.f);
                if (a.vpre_out && blockIdx.x == 0) a.vpre_out[(size_t)b * 256 + o] = vs[o];
            }
        }                                                    // (no partials: kl / th were filed with the weight table, above)
    }
    __syncthreads();
    // finisher, part 3: per-kernel normalisation (TM:327-329) / the shared Linear(100 -> 6) + identity (TM:462-468)
    if (MODE != 2 && a.partials) {
        const int o = tid;
        if (MODE == 0) {
            if (o < 25 * NM) {
                const int g = (o / 25) * 25;
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < 25; ++i) sum += vs[g + i];
                const float kv = vs[o] / sum;
                kl[(o / 25) * 28 + (o - g)] = kv;
                if (a.kerns_out && blockIdx.x == 0) a.kerns_out[(size_t)b * 25 * NM + o] = kv;
            }
        } else if (o < 6) {
            double t = a.b2[o];
            for (int i = 0; i < 100; ++i) t = fma((double)a.w2[o * 100 + i], (double)vs[i], t);
            const float tv = (float)(t + ((o == 0 || o == 4) ? 1.0 : 0.0));
            th[o] = tv;
            if (a.kerns_out && blockIdx.x == 0) a.kerns_out[b * 6 + o] = tv;
        }
        __syncthreads();      // (block-uniform condition)
    }
    FH_STAMP(10);

    // ---- softmaxed masks, motion transform, blend (the body of composite_kernel, operands from LDS) --------------------------------------
    for (int pp = tid; pp < np; pp += FH_NT) {
        const int p = p0 + pp;
        const int y = p / W, x = p - y * W;
        const int ry = y - y0;
        float l0v[3] = {0.f, 0.f, 0.f};
        if (MODE != 2) {
#pragma unroll
            for (int c = 0; c < 3; ++c) l0v[c] = l0t[c * np + pp];
        }
        float mk[12];
#pragma unroll
        for (int m = 0; m < 12; ++m) {                       // plane index clamped, not predicated: a predicate around the LDS reads made twelve
            const int mc = min(m, NP - 1);                   // dependent read -> wait -> exp chains of this loop (1.6 us by the stamps)
            const int F = mc * HW + p;
            const int gi = div_np(F) - div_np(mc * HW + p0);
            const float v = lg[mc * win + pp + hn];
            const float mv = __expf(v - gmx[mc * G + gi]) * ginv[mc * G + gi];
            mk[m] = m < NP ? mv : 0.f;
        }
        if (a.masks_out) {
#pragma unroll
            for (int m = 0; m < 12; ++m)
                if (m < NP) a.masks_out[((size_t)b * NP + m) * HW + p] = mk[m];
        }
        FH_STAMP(13);
        float o3[3];
        if (MODE == 0) {
            // The per-pixel blended kernel keff[tap] = sum_k mask[k + 2] kern[k][tap] (composite_kernel: a chain of fmaf over k ascending from 0) on the
            // matrix cores (round 6): a [64 pixels x 10] x [10 x 25] product per wave as 2 x 5 v_mfma_f32_32x32x2_f32 -- C = 0, instruction s fed
            // k = 2s (lanes 0-31) and 2s + 1 (lanes 32-63): the same chain, bit-identical -- and back through the wave's tile to "lane = pixel".
            // Before: every lane read the whole table (70 wave-uniform ds_read_b128 = 560 LDS cycles per wave, eight waves on the CU's one LDS
            // pipe) for 250 FMAs: 2.2 us of the block's 4.2-us blend by the stamps.  k = NM - 1 (the generated kernel no mask pairs with, TM:725
            // zip) meets mk[NP] = 0; table rows >= NM are zero.
            float keff[25];
            {
                const int half = lane >> 5, l31 = lane & 31;
                f32x16 ka[2];
#pragma unroll
                for (int r = 0; r < 16; ++r) { ka[0][r] = 0.f; ka[1][r] = 0.f; }
#pragma unroll
                for (int s5 = 0; s5 < 5; ++s5) {
                    const float own_lo = mk[2 * s5 + 2], own_hi = mk[2 * s5 + 3];
                    float sa, sb;
                    wave_swap_halves(half ? own_lo : own_hi, true, sa, sb);
                    const float xch = half ? sa : sb;                                // pixel lane ^ 32's mask of the k this half feeds
                    const float bvk = kl[(2 * s5 + half) * 28 + l31];                // (columns 28-31: the next row's entries, unused outputs)
                    ka[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? xch : own_lo, bvk, ka[0], 0, 0, 0);
                    ka[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? own_hi : xch, bvk, ka[1], 0, 0, 0);
                }
                float* kt = un + wave * FH_TILE;
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) kt[(m2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * FH_XP + l31] = ka[m2][r];
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(kt + lane * FH_XP + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (q * 4 + e < 25) keff[q * 4 + e] = t4[e];
                }
            }
            FH_STAMP(14);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* pt = prevt + (c * (FH_TR + 4) + ry) * PW + x;
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int j = 0; j < 5; ++j) t = fmaf(keff[i * 5 + j], pt[i * PW + j], t);
                const float pc = pt[2 * PW + 2];
                o3[c] = mk[0] * pc + mk[1] * l0v[c] + t;
            }
        } else if (MODE == 1) {
            const double xs = -1.0 + 2.0 * (double)x / (double)(W - 1);
            const double ys = -1.0 + 2.0 * (double)y / (double)(H - 1);
            double gu = (double)th[0] * xs + (double)th[1] * ys + (double)th[2];
            double gv = (double)th[3] * xs + (double)th[4] * ys + (double)th[5];
            if (!a.stp_zero) { gu = fmin(fmax(gu, -1.0), 1.0); gv = fmin(fmax(gv, -1.0), 1.0); }
            const double u = (gu + 1.0) * (double)(W - 1) * 0.5;
            const double v = (gv + 1.0) * (double)(H - 1) * 0.5;
            double u0 = floor(u), v0 = floor(v);
            if (!a.stp_zero) { u0 = fmin(fmax(u0, 0.0), (double)(W - 2)); v0 = fmin(fmax(v0, 0.0), (double)(H - 2)); }
            const float wu1 = (float)(u - u0), wv1 = (float)(v - v0);
            const int iu = (int)u0, iv = (int)v0;
            float msum = 0.f;
#pragma unroll
            for (int q = 2; q < 12; ++q) msum += mk[q];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float t = 0.f;
#pragma unroll
                for (int dv = 0; dv < 2; ++dv)
#pragma unroll
                    for (int du = 0; du < 2; ++du) {
                        const int uu = iu + du, vv = iv + dv;
                        const float wgt = (dv ? wv1 : 1.f - wv1) * (du ? wu1 : 1.f - wu1);
                        if ((unsigned)uu < (unsigned)W && (unsigned)vv < (unsigned)H) t = fmaf(wgt, pb[(size_t)c * HW + vv * W + uu], t);
                    }
                o3[c] = mk[0] * pb[(size_t)c * HW + p] + mk[1] * l0v[c] + msum * t;
            }
        } else {
            const float* e7 = a.enc7 + (size_t)b * 25 * HW;
            float kn[25];
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 25; ++i) {
                kn[i] = fmaxf(e7[(size_t)i * HW + p] - 1e-12f, 0.f) + 1e-12f;
                sum += kn[i];
            }
            const float inv = 1.0f / sum;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* pt = prevt + (c * (FH_TR + 4) + ry) * PW + x;
                float t = 0.f;
#pragma unroll
                for (int xk = 0; xk < 5; ++xk)
#pragma unroll
                    for (int yk = 0; yk < 5; ++yk) {
                        const bool ok = (y + xk < H) && (x + yk < W);   // TM:400 slice quirk
                        t = fmaf(kn[xk * 5 + yk] * inv, ok ? pt[xk * PW + yk] : 0.f, t);
                    }
                o3[c] = mk[0] * pt[2 * PW + 2] + mk[1] * t;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) a.out[((size_t)b * 3 + c) * HW + p] = o3[c];
    }
    FH_STAMP(11);
}

static size_t frame_head_lds_floats(int mode, int W, int NM) {
    const int NP = NM + 1, np = FH_TR * W, win = np + 2 * (NP - 1), G = np / NP + 2, PW = W + 4;
    size_t f = (size_t)32 * FH_WP + 32 + ((NP * win + 3) & ~3) + (mode == 2 ? 0 : 3 * np) + FH_KL + 256 + 8;
    const size_t halo = 24 * FH_HP, groups = (size_t)(2 * NP * G + 3) & ~(size_t)3;      // the halo rows, then (gmx, ginv), share one region
    f += ((3 * (FH_TR + 4) * PW + 3) & ~3) + (halo > groups ? halo : groups);
    return f + (size_t)FH_HW * FH_TILE;
}

// can the fused launch serve this geometry?  (else: heads_1x1 + cdna_kernels / stp_params + composite)
bool frame_head_ok(int mode, int B, int H, int W, int num_masks) {
    if (mode < 0 || mode > 2 || B <= 0 || H <= 1 || W <= 1 || num_masks < 1 || num_masks > 11) return false;
    if (mode == 2 && num_masks != 1) return false;
    if (H % FH_TR || (FH_TR * W) % 64 || W + 4 > 256 || num_masks + 1 < 2) return false;
    return frame_head_lds_floats(mode, W, num_masks) * sizeof(float) <= 160 * 1024;
}
// ... and is it the faster form?  Only while two blocks share a CU (<= 80 KB of LDS each: frames up to 64 wide).  At 128 x 128 a block needs
// 92 KB, the 1,024 blocks of B = 32 run in four rounds, and the separate kernels win: rollout 65.7 against 66.5 ms (T = 20, profiles/r04).
bool frame_head_pays(int mode, int B, int H, int W, int num_masks) {
    return frame_head_ok(mode, B, H, W, num_masks) && frame_head_lds_floats(mode, W, num_masks) * sizeof(float) * 2 <= 160 * 1024;
}
// floats of a partial-sum buffer for B samples and K inputs: [B][KS][256] plus the tail the finisher's unclamped loads may touch
long long motion_partials_floats(int B, int K) { return ((long long)B * cdna_kernel_partials_slices(K) + FH_MAXKS) * 256; }
bool frame_head_finishes(int K) { return cdna_kernel_partials_slices(K) <= FH_MAXKS; }   // the in-kernel finisher takes this many K slices

int frame_head(const FrameHeadArgs& a, int mode, hipStream_t s) {
    PIVP_CHECK_ARG(frame_head_ok(mode, a.B, a.H, a.W, a.NM));
    PIVP_CHECK_ARG(a.e6raw && a.ln_part && a.ln_nparts > 0 && a.gamma && a.beta && a.wm && a.bm && a.we && a.be && a.prev && a.out && a.enc7);
    PIVP_CHECK_ARG(mode == 2 || a.partials || a.aux);
    PIVP_CHECK_ARG(!a.partials || mode == 2 || (a.KS >= 1 && a.KS <= FH_MAXKS && a.hbias && (mode != 1 || (a.w2 && a.b2))));
    const int lds = (int)(frame_head_lds_floats(mode, a.W, a.NM) * sizeof(float));
    const bool ref_geom = a.W == 64 && a.NM == (mode == 2 ? 1 : 10);      // the reference's configurations: the constant-folded instance
    static PerDeviceOnce once[6];
    const void* fns[6] = {reinterpret_cast<const void*>(&frame_head_kernel<0, 0, 0>), reinterpret_cast<const void*>(&frame_head_kernel<1, 0, 0>),
                          reinterpret_cast<const void*>(&frame_head_kernel<2, 0, 0>), reinterpret_cast<const void*>(&frame_head_kernel<0, 10, 64>),
                          reinterpret_cast<const void*>(&frame_head_kernel<1, 10, 64>), reinterpret_cast<const void*>(&frame_head_kernel<2, 1, 64>)};
    const int which = mode + (ref_geom ? 3 : 0);
    // the cap is raised once per device to the most any geometry asks for (160 KB); the launch passes the bytes this one needs
    if (pivp_ensure_dyn_lds(once[which], fns[which], 160 * 1024) != PIVP_OK) return PIVP_ERR_LAUNCH;
    const dim3 grid(a.H / FH_TR, a.B);
    switch (which) {
    case 0: hipLaunchKernelGGL((frame_head_kernel<0, 0, 0>), grid, dim3(FH_NT), lds, s, a); break;
    case 1: hipLaunchKernelGGL((frame_head_kernel<1, 0, 0>), grid, dim3(FH_NT), lds, s, a); break;
    case 2: hipLaunchKernelGGL((frame_head_kernel<2, 0, 0>), grid, dim3(FH_NT), lds, s, a); break;
    case 3: hipLaunchKernelGGL((frame_head_kernel<0, 10, 64>), grid, dim3(FH_NT), lds, s, a); break;
    case 4: hipLaunchKernelGGL((frame_head_kernel<1, 10, 64>), grid, dim3(FH_NT), lds, s, a); break;
    default: hipLaunchKernelGGL((frame_head_kernel<2, 1, 64>), grid, dim3(FH_NT), lds, s, a); break;
    }
    return PIVP_LAUNCH_STATUS();
}

}  // namespace pivp

#ifdef PIVP_FH_STAMPS
extern "C" int pivp_debug_fh_stamps(long long* out) {   // 8 x 16 values of block (3, 5), 10 ns ticks
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pivp::pivp_fh_stamps), sizeof(long long) * 128) == hipSuccess ? 0 : -2;
}
#endif
