// Backward of the output heads (CDNA variant) and of the small ops at both ends of the trunk.
// Forward definitions: heads.hip / small_kernels.hip; reference code: train_model.py ("TM") lines cited per kernel.
#include <stdlib.h>

#include "pivp_kernels.h"

namespace pivp {

// ------------------------------------------------------------------------------------------
// Loss gradient (TM:737-758): d loss / d gen[t] = 2 (gen[t] - images[t+1]) / (numel * (T-ctx)) for t >= ctx-1,
// and 1e-4 times the same for the predicted states.  out = scale * (a - b)  (+ out when accum).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scaled_diff_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                          long n, float scale, int accum, int vec) {
    PIVP_SET_MAIN_PRIO();
    const long n4 = vec ? n >> 2 : 0;       // float4 body when the three pointers are 16-byte aligned, scalar tail
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 va = reinterpret_cast<const f32x4*>(a)[i], vb = reinterpret_cast<const f32x4*>(b)[i];
        f32x4 v = (va - vb) * scale;
        if (accum) v += reinterpret_cast<const f32x4*>(out)[i];
        reinterpret_cast<f32x4*>(out)[i] = v;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = scale * (a[i] - b[i]);
        out[i] = accum ? out[i] + v : v;
    }
}
int scaled_diff(const float* a, const float* b, float* out, long n, float scale, int accum, hipStream_t s) {
    PIVP_CHECK_ARG(a && b && out && n > 0);
    const int vec = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0;
    const long blocks = ((vec ? n / 4 + 3 : n) + 255) / 256;
    hipLaunchKernelGGL(scaled_diff_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, a, b, out, n, scale, accum, vec);
    return PIVP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
// composite backward, CDNA (forward: composite_kernel<0>, TM:720-728 + TM:341-349).
// Per block = one sample x 8 rows (like the forward).  Given go = d loss / d out:
//   d mk[0] = sum_c go*prev, d mk[1] = sum_c go*L0, d mk[k+2] = sum_c go * (prev (*) kern_k)       -> dmk planar
//   d z (enc7 pre-activation) = go * mk[1] * L0 (1 - L0) [L0 > 1/2]   (L0 = sigmoid(relu(z)), TM:315-317)   -> dz planar
//   d kern[k][ij] partial = sum_{c,p in tile} go[c](p) mk[k+2](p) prev[c](p + ij - 2)                      -> dkpart
//   d prev[c](q) = mk[0](q) go[c](q) + sum_ij sum_k kern[k][ij] (mk[k+2] go[c])(q - (ij - 2))     (when dprev != null)
// The softmax Jacobian couples 11 consecutive flat elements and is applied by mask_softmax_bwd_kernel afterwards.
// ------------------------------------------------------------------------------------------
// rows of the frame per block tile: 8 up to 64-wide frames, 4 for 128-wide ones (the tile's LDS images grow with W)
static int composite_bwd_rows(int W) {
    return W <= 64 ? 8 : 4;
}

// threads per block: four waves per SIMD (the kernel is a chain of LDS-fed FMA loops and its grid is one ~100 KB block per CU; with 256
// threads every wait was exposed: 68.3 us per launch, 512 threads 45.5, 1024 threads 40.4)
constexpr int CBC_NT = 1024;

// NMC / WC / HC (round 6): num_masks and the frame size as compile-time constants (0 = from the arguments) for the reference's geometry (10 masks, 64 x 64):
// the kernel is a chain of LDS-fed loops whose trip counts and predicates hang on them (frame_head.hip has the measurement of what that costs)
template <int CB_TR, int NMC = 0, int WC = 0, int HC = 0>
__global__ __launch_bounds__(CBC_NT) void composite_bwd_cdna_kernel(const float* __restrict__ prev, const float* __restrict__ logits,
                                                                 const float* __restrict__ layer0, const float* __restrict__ kerns,
                                                                 const float* __restrict__ go, float* __restrict__ dmk,
                                                                 float* __restrict__ dz, float* __restrict__ dkpart,
                                                                 float* __restrict__ dprev, int dprev_accum, int H_, int W_, int NM_) {
    PIVP_SET_MAIN_PRIO();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H = HC ? HC : H_, W = WC ? WC : W_, NM = NMC ? NMC : NM_;
    const int NP = NM + 1, HW = H * W, NK = NM - 1;
    const int b = blockIdx.y, y0 = blockIdx.x * CB_TR;
    const int rows = min(CB_TR, H - y0);
    const int ey0 = max(0, y0 - 2), ey1 = min(H, y0 + rows + 2);       // extended rows (tile + 2-row halo inside the image)
    const int erows = ey1 - ey0, enp = erows * W, ep0 = ey0 * W;
    const int win = enp + 2 * (NP - 1);
    const int G = enp / NP + 2;
    const int PW = W + 4, PR = CB_TR + 4;
    float* lg = sm;                              // [NP][win]     logits window
    float* gmx = lg + NP * win;                  // [NP][G]
    float* ginv = gmx + NP * G;                  // [NP][G]
    float* mkx = ginv + NP * G;                  // [NP][(CB_TR+4)*W]  masks on the extended tile
    float* gox = mkx + NP * PR * W;              // [3][(CB_TR+4)*W]   go on the extended tile
    float* prevt = gox + 3 * PR * W;             // [3][PR][PW]        prev with zero halo, rows y0-2 ..
    float* kl = prevt + 3 * PR * PW;             // [NM*25]
    const int tid = threadIdx.x;
    const float* lgb = logits + (size_t)b * NP * HW;
    // exact x / NP for x * NP < 2^32 (see composite_kernel); staging loops without per-element division and with four
    // independent loads in flight per thread (the flat-index loops of the first version made one L2 round trip per element)
    const unsigned magic = 0xFFFFFFFFu / (unsigned)NP + 1u;
    auto div_np = [&](int x) { return (int)__umulhi((unsigned)x, magic); };
    // Staging: EVERY load of the block is requested before the first LDS store, by all 1,024 threads (round 6).  Before, 256 threads walked the
    // logits planes four loads at a time -- plane after plane: 11 serial L2 round trips --, then the go planes (3 more) behind the frame tile.
    const float* pb = prev + (size_t)b * 3 * HW;
    const float* gb = go + (size_t)b * 3 * HW;
    constexpr int LU = 2;                        // window / tile elements per thread and plane (win, enp, PR * PW <= 2,048: checked by the launcher)
    float tl[12][LU], tg[3][LU], tp[3][LU];
#pragma unroll
    for (int m = 0; m < 12; ++m)
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int j = tid + CBC_NT * u, F = m * HW + ep0 - (NP - 1) + j;
            tl[m][u] = (m < NP && j < win && F >= 0 && F < NP * HW) ? lgb[F] : 0.f;
        }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int pp = tid + CBC_NT * u;
            tg[c][u] = pp < enp ? gb[(size_t)c * HW + ep0 + pp] : 0.f;
            const int r = pp / PW, x = pp - r * PW, iy = y0 + r - 2, ix = x - 2;
            tp[c][u] = (r < PR && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? pb[(size_t)c * HW + iy * W + ix] : 0.f;
        }
    const float kv = tid < NM * 25 ? kerns[(size_t)b * NM * 25 + tid] : 0.f;
#pragma unroll
    for (int m = 0; m < 12; ++m)
#pragma unroll
        for (int u = 0; u < LU; ++u) { const int j = tid + CBC_NT * u; if (m < NP && j < win) lg[m * win + j] = tl[m][u]; }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int pp = tid + CBC_NT * u;
            if (pp < enp) gox[c * PR * W + pp] = tg[c][u];
            if (pp < PR * PW) prevt[c * PR * PW + pp] = tp[c][u];
        }
    if (tid < NM * 25) kl[tid] = kv;
    __syncthreads();
    for (int i = tid; i < NP * G; i += CBC_NT) {
        const int m = i / G, gi = i - m * G;
        const int gfirst = div_np(m * HW + ep0), glast = div_np(m * HW + ep0 + enp - 1);
        if (gfirst + gi <= glast) {
            const float* e = lg + m * win + (gfirst + gi) * NP - (m * HW + ep0 - (NP - 1));
            float ev[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) ev[u] = u < NP ? e[u] : -3.0e38f;
            float mx = ev[0];
#pragma unroll
            for (int u = 1; u < 12; ++u) mx = fmaxf(mx, ev[u]);
            float sum = 0.f;
#pragma unroll
            for (int u = 0; u < 12; ++u) sum += u < NP ? __expf(ev[u] - mx) : 0.f;
            gmx[i] = mx; ginv[i] = 1.0f / sum;
        }
    }
    __syncthreads();
    for (int m = 0; m < NP; ++m) {
        const int gbase = div_np(m * HW + ep0);
        for (int pp = tid; pp < enp; pp += CBC_NT) {
            const int gi = div_np(m * HW + ep0 + pp) - gbase;
            mkx[m * PR * W + pp] = __expf(lg[m * win + pp + (NP - 1)] - gmx[m * G + gi]) * ginv[m * G + gi];
        }
    }
    __syncthreads();
    const int toff = (y0 - ey0) * W;             // tile offset inside the extended arrays
    const int np = rows * W;
    // ---- per-pixel gradients of the masks and of enc7 ----
    for (int pp = tid; pp < np; pp += CBC_NT) {
        const int p = y0 * W + pp, y = p / W, x = p - y * W, ry = y - y0;
        const float g0 = gox[0 * PR * W + toff + pp], g1 = gox[1 * PR * W + toff + pp], g2 = gox[2 * PR * W + toff + pp];
        const float* p0 = prevt + (0 * PR + ry) * PW + x;
        const float* p1 = prevt + (1 * PR + ry) * PW + x;
        const float* p2 = prevt + (2 * PR + ry) * PW + x;
        float* dm = dmk + (size_t)b * NP * HW + p;
        dm[0] = g0 * p0[2 * PW + 2] + g1 * p1[2 * PW + 2] + g2 * p2[2 * PW + 2];
        const float l0 = layer0[((size_t)b * 3 + 0) * HW + p], l1 = layer0[((size_t)b * 3 + 1) * HW + p], l2 = layer0[((size_t)b * 3 + 2) * HW + p];
        dm[(size_t)HW] = g0 * l0 + g1 * l1 + g2 * l2;
        const float m1 = mkx[1 * PR * W + toff + pp];
        dz[((size_t)b * 3 + 0) * HW + p] = l0 > 0.5f ? g0 * m1 * l0 * (1.f - l0) : 0.f;
        dz[((size_t)b * 3 + 1) * HW + p] = l1 > 0.5f ? g1 * m1 * l1 * (1.f - l1) : 0.f;
        dz[((size_t)b * 3 + 2) * HW + p] = l2 > 0.5f ? g2 * m1 * l2 * (1.f - l2) : 0.f;
        float gp[25];   // sum_c go[c] * prev[c] at the 25 taps: shared by all NK kernels (the first version recomputed it per kernel)
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) gp[i * 5 + j] = g0 * p0[i * PW + j] + g1 * p1[i * PW + j] + g2 * p2[i * PW + j];
        for (int k = 0; k < NM; ++k) {
            float t = 0.f;
            if (k < NK) {
                const float* kk = kl + k * 25;
#pragma unroll
                for (int i = 0; i < 25; ++i) t = fmaf(kk[i], gp[i], t);
            }
            if (k + 2 < NP) dm[(size_t)(k + 2) * HW] = t;
        }
    }
    // ---- kernel gradient partials: dK[k][ij] = sum_pixels mk[k+2] * (sum_c go[c] * prev[c](+ij)).  Thread (row of the tile,
    // tap ij) forms the go*prev product ONCE per pixel and feeds all NK kernels (9 accumulators); the 8 rows are summed
    // through LDS.  (One thread per (k, ij) walking all 512 pixels redid the product per kernel: 3.6k LDS reads per thread.)
    // [CBC_NT / 256][CB_TR][256]: one image per quarter of the row.  Its own LDS region: the logits window it used to overlay (lg) is only
    // NP * win floats, smaller than this for num_masks < 10 (with 1024 threads the overlay ran into the masks: test_bptt_gradients_cdna_four_masks)
    float* red = kl + ((NM * 25 + 3) & ~3);
    const int xh = tid >> 8, t8 = tid & 255;
    const int xa = (W * xh) / (CBC_NT / 256), xb = (W * (xh + 1)) / (CBC_NT / 256);
    if (t8 < CB_TR * 25) {
        const int ry = t8 / 25, ij = t8 - ry * 25, i = ij / 5, j = ij - i * 5;
        float acc[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] = 0.f;
        if (ry < rows) {
            const float* g0r = gox + toff + ry * W;
            const float* q0 = prevt + (0 * PR + ry + i) * PW + j;
            const float* q1 = prevt + (1 * PR + ry + i) * PW + j;
            const float* q2 = prevt + (2 * PR + ry + i) * PW + j;
            const float* mrow = mkx + 2 * PR * W + toff + ry * W;
#pragma unroll 2
            for (int x = xa; x < xb; ++x) {
                const float v = g0r[x] * q0[x] + g0r[PR * W + x] * q1[x] + g0r[2 * PR * W + x] * q2[x];
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    if (k < NK) acc[k] = fmaf(mrow[k * PR * W + x], v, acc[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) red[(xh * CB_TR + ry) * 256 + k * 25 + ij] = acc[k];
    }
    __syncthreads();
    {
        float acc = 0.f;
        if (tid < NK * 25) {
#pragma unroll
            for (int r = 0; r < CB_TR * (CBC_NT / 256); ++r) acc += red[r * 256 + tid];
        }
        if (tid < 256) dkpart[((size_t)b * gridDim.x + blockIdx.x) * 256 + tid] = acc;
    }
    // ---- gradient w.r.t. the previous frame (feed-self only) ----
    if (dprev) {
        for (int pp = tid; pp < np; pp += CBC_NT) {
            const int p = y0 * W + pp, y = p / W, x = p - y * W;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int sy = y - (i - 2);                       // source pixel row: q - (ij - 2)
                if (sy < ey0 || sy >= ey1) continue;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int sx = x - (j - 2);
                    if ((unsigned)sx >= (unsigned)W) continue;
                    const int sp = (sy - ey0) * W + sx;
                    float wsum = 0.f;
                    for (int k = 0; k < NK; ++k) wsum = fmaf(kl[k * 25 + i * 5 + j], mkx[(k + 2) * PR * W + sp], wsum);
                    a0 = fmaf(wsum, gox[sp], a0); a1 = fmaf(wsum, gox[PR * W + sp], a1); a2 = fmaf(wsum, gox[2 * PR * W + sp], a2);
                }
            }
            const float m0 = mkx[toff + pp];
            float* dp = dprev + (size_t)b * 3 * HW + p;
            const float v0 = a0 + m0 * gox[toff + pp], v1 = a1 + m0 * gox[PR * W + toff + pp], v2 = a2 + m0 * gox[2 * PR * W + toff + pp];
            if (dprev_accum) { dp[0] += v0; dp[(size_t)HW] += v1; dp[2 * (size_t)HW] += v2; }
            else { dp[0] = v0; dp[(size_t)HW] = v1; dp[2 * (size_t)HW] = v2; }
        }
    }
}

int composite_bwd_tiles(int H, int W) { const int tr = composite_bwd_rows(W); return (H + tr - 1) / tr; }

int composite_bwd_cdna(const float* prev, const float* logits, const float* layer0, const float* kerns, const float* go,
                       float* dmk, float* dz, float* dkpart, float* dprev, int dprev_accum, int B, int H, int W, int NM, hipStream_t s) {
    PIVP_CHECK_ARG(prev && logits && layer0 && kerns && go && dmk && dz && dkpart && B > 0 && H > 1 && W > 1 && NM >= 1 && NM <= 10);
    const int CB_TR = composite_bwd_rows(W);
    const int NP = NM + 1, PR = CB_TR + 4;
    const int enp = PR * W, win = enp + 2 * (NP - 1), G = enp / NP + 2;
    const size_t lds = sizeof(float) * ((size_t)NP * win + 2 * NP * G + (size_t)NP * PR * W + 3 * PR * W + 3 * PR * (W + 4) + ((NM * 25 + 3) & ~3) +
                                        (size_t)CB_TR * CBC_NT);      // ... + the kernel-gradient row partials
    PIVP_CHECK_ARG(lds <= 160 * 1024 && W + 4 <= 256 && NM * 25 <= 256 && win <= 2 * CBC_NT && PR * (W + 4) <= 2 * CBC_NT);
    if (CB_TR == 8 && NM == 10 && W == 64 && H == 64) {      // the reference's geometry: the constant-folded instance
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&composite_bwd_cdna_kernel<8, 10, 64, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((composite_bwd_cdna_kernel<8, 10, 64, 64>), dim3(composite_bwd_tiles(H, W), B), dim3(CBC_NT), lds, s, prev, logits, layer0, kerns, go,
                           dmk, dz, dkpart, dprev, dprev_accum, H, W, NM);
    } else if (CB_TR == 8) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&composite_bwd_cdna_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(composite_bwd_cdna_kernel<8>, dim3(composite_bwd_tiles(H, W), B), dim3(CBC_NT), lds, s, prev, logits, layer0, kerns, go,
                           dmk, dz, dkpart, dprev, dprev_accum, H, W, NM);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&composite_bwd_cdna_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(composite_bwd_cdna_kernel<4>, dim3(composite_bwd_tiles(H, W), B), dim3(CBC_NT), lds, s, prev, logits, layer0, kerns, go,
                           dmk, dz, dkpart, dprev, dprev_accum, H, W, NM);
    }
    return PIVP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
// composite backward, DNA (forward: composite_kernel<2>, TM:392-415 + TM:720-728, num_masks = 1 so NP = 2):
//   kn_i = relu(e7_i - 1e-12) + 1e-12, w_i = kn_i / sum kn, t[c] = sum_i w_i in_i[c], out = mk0*prev + mk1*t
//   in_(xk,yk)[c](y,x) = prev[c](y+xk-2, x+yk-2) if (y+xk < H and x+yk < W) else 0      (the reference's slice quirk, TM:400)
//   d mk0 = sum_c go*prev, d mk1 = sum_c go*t;  a_i = sum_c go[c] mk1 in_i[c];  d kn_j = (a_j - sum_i a_i w_i) / S
//   d e7_j = d kn_j [e7_j - 1e-12 > 0]  (e7 = relu(conv): same mask)  -> dz planar [25]
//   d prev[c](q) = mk0 go + sum_i [q.y+2 < H and q.x+2 < W] (w_i mk1 go[c])(q - (xk-2, yk-2))
// ------------------------------------------------------------------------------------------
template <int CB_TR>
__global__ __launch_bounds__(256) void composite_bwd_dna_kernel(const float* __restrict__ prev, const float* __restrict__ logits,
                                                                const float* __restrict__ e7, const float* __restrict__ go,
                                                                float* __restrict__ dmk, float* __restrict__ dz,
                                                                float* __restrict__ dprev, int dprev_accum, int H, int W) {
    PIVP_SET_MAIN_PRIO();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NP = 2;
    const int HW = H * W;
    const int b = blockIdx.y, y0 = blockIdx.x * CB_TR;
    const int rows = min(CB_TR, H - y0);
    const int ey0 = max(0, y0 - 2), ey1 = min(H, y0 + rows + 2);
    const int erows = ey1 - ey0, enp = erows * W, ep0 = ey0 * W;
    const int PR = CB_TR + 4, PW = W + 4;
    float* wx = sm;                          // [25][PR*W]   normalised kernels on the extended tile
    float* dtx = wx + 25 * PR * W;           // [3][PR*W]    mk1 * go on the extended tile
    float* m0x = dtx + 3 * PR * W;           // [PR*W]       mk0 on the extended tile
    float* prevt = m0x + PR * W;             // [3][PR][PW]
    const int tid = threadIdx.x;
    const float* lgb = logits + (size_t)b * NP * HW;
    const float* pb = prev + (size_t)b * 3 * HW;
    const float* gb = go + (size_t)b * 3 * HW;
    const float* eb = e7 + (size_t)b * 25 * HW;
    for (int i = tid; i < 3 * PR * PW; i += 256) {
        const int c = i / (PR * PW), r = (i / PW) % PR, x = i % PW;
        const int iy = y0 + r - 2, ix = x - 2;
        prevt[i] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? pb[(size_t)c * HW + iy * W + ix] : 0.f;
    }
    for (int pp = tid; pp < enp; pp += 256) {
        const int p = ep0 + pp;
        // flat softmax over NP = 2 consecutive elements of the planar logits (TM:720-722)
        float mk[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int F = m * HW + p, g0 = (F / 2) * 2;
            const float a = lgb[g0], c2 = lgb[g0 + 1], mx = fmaxf(a, c2);
            const float ea = expf(a - mx), ec = expf(c2 - mx);
            mk[m] = (F == g0 ? ea : ec) / (ea + ec);
        }
        m0x[pp] = mk[0];
        float kn[25], S = 0.f;
#pragma unroll
        for (int i = 0; i < 25; ++i) { kn[i] = fmaxf(eb[(size_t)i * HW + p] - 1e-12f, 0.f) + 1e-12f; S += kn[i]; }
        const float inv = 1.0f / S;
#pragma unroll
        for (int i = 0; i < 25; ++i) wx[i * PR * W + pp] = kn[i] * inv;
#pragma unroll
        for (int c = 0; c < 3; ++c) dtx[c * PR * W + pp] = mk[1] * gb[(size_t)c * HW + p];
    }
    __syncthreads();
    const int toff = (y0 - ey0) * W, np = rows * W;
    for (int pp = tid; pp < np; pp += 256) {
        const int p = y0 * W + pp, y = p / W, x = p - y * W, ry = y - y0;
        const float g0 = gb[p], g1 = gb[(size_t)HW + p], g2 = gb[2 * (size_t)HW + p];
        const float* p0 = prevt + (0 * PR + ry) * PW + x;
        const float* p1 = prevt + (1 * PR + ry) * PW + x;
        const float* p2 = prevt + (2 * PR + ry) * PW + x;
        const float d0 = dtx[toff + pp], d1 = dtx[PR * W + toff + pp], d2 = dtx[2 * PR * W + toff + pp];
        float a[25], dot = 0.f, t0 = 0.f, t1 = 0.f, t2 = 0.f, S = 0.f;
#pragma unroll
        for (int xk = 0; xk < 5; ++xk)
#pragma unroll
            for (int yk = 0; yk < 5; ++yk) {
                const int i = xk * 5 + yk;
                const bool ok = (y + xk < H) && (x + yk < W);
                const float i0 = ok ? p0[xk * PW + yk] : 0.f, i1 = ok ? p1[xk * PW + yk] : 0.f, i2 = ok ? p2[xk * PW + yk] : 0.f;
                const float w = wx[i * PR * W + toff + pp];
                a[i] = d0 * i0 + d1 * i1 + d2 * i2;
                dot = fmaf(a[i], w, dot);
                t0 = fmaf(w, i0, t0); t1 = fmaf(w, i1, t1); t2 = fmaf(w, i2, t2);
                const float kn = fmaxf(eb[(size_t)i * HW + p] - 1e-12f, 0.f) + 1e-12f;
                S += kn;
            }
        const float inv = 1.0f / S;
#pragma unroll
        for (int i = 0; i < 25; ++i) {
            const float ev = eb[(size_t)i * HW + p];
            dz[((size_t)b * 25 + i) * HW + p] = (ev - 1e-12f > 0.f) ? (a[i] - dot) * inv : 0.f;
        }
        float* dm = dmk + (size_t)b * NP * HW + p;
        dm[0] = g0 * p0[2 * PW + 2] + g1 * p1[2 * PW + 2] + g2 * p2[2 * PW + 2];
        dm[(size_t)HW] = g0 * t0 + g1 * t1 + g2 * t2;
        if (dprev) {
            // TM:404 `kernel_inputs.append(tmp.data)`: the 25 shifted copies of the previous frame are DETACHED in the reference, so
            // nothing flows into the frame through the per-pixel kernels; only the mask-0 term m0 * go does (TM:725).
            const float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            const float m0 = m0x[toff + pp];
            float* dp = dprev + (size_t)b * 3 * HW + p;
            const float v0 = a0 + m0 * g0, v1 = a1 + m0 * g1, v2 = a2 + m0 * g2;
            if (dprev_accum) { dp[0] += v0; dp[(size_t)HW] += v1; dp[2 * (size_t)HW] += v2; }
            else { dp[0] = v0; dp[(size_t)HW] = v1; dp[2 * (size_t)HW] = v2; }
        }
    }
}

int composite_bwd_dna(const float* prev, const float* logits, const float* e7, const float* go, float* dmk, float* dz,
                      float* dprev, int dprev_accum, int B, int H, int W, hipStream_t s) {
    PIVP_CHECK_ARG(prev && logits && e7 && go && dmk && dz && B > 0 && H > 1 && W > 1);
    const int CB_TR = composite_bwd_rows(W);
    const int PR = CB_TR + 4;
    const size_t lds = sizeof(float) * ((size_t)29 * PR * W + 3 * PR * (W + 4));
    PIVP_CHECK_ARG(lds <= 160 * 1024);
    if (CB_TR == 8) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&composite_bwd_dna_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(composite_bwd_dna_kernel<8>, dim3(composite_bwd_tiles(H, W), B), dim3(256), lds, s, prev, logits, e7, go, dmk, dz, dprev,
                           dprev_accum, H, W);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&composite_bwd_dna_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(composite_bwd_dna_kernel<4>, dim3(composite_bwd_tiles(H, W), B), dim3(256), lds, s, prev, logits, e7, go, dmk, dz, dprev,
                           dprev_accum, H, W);
    }
    return PIVP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
// Backward of the flat softmax + ReLU of the mask head (TM:719-722): for each group of NP consecutive flat
// elements, d r = mk * (d mk - sum_group mk * d mk), masked by r > 0 (r = relu(masks conv)).  In place on dmk.
// ------------------------------------------------------------------------------------------
// A thread owns one group; the block's 256 groups are 256 * NP CONSECUTIVE floats of each array, so they travel through LDS in whole
// lines (the first version let every lane walk its own group in global memory: 44-B strides, 23 cache lines per wave-instruction, 30 us
// per launch for 17 MB).  In LDS the groups are NP floats apart: NP odd (11 for num_masks = 10) makes the lanes' banks distinct.
__global__ __launch_bounds__(256) void mask_softmax_bwd_kernel(const float* __restrict__ logits, float* __restrict__ dmk, long ngroups, int NP) {
    PIVP_SET_MAIN_PRIO();
    __shared__ float lr[256 * 12], ld[256 * 12];
    const long g0 = (long)blockIdx.x * 256;
    const int ng = (int)min((long)256, ngroups - g0), n = ng * NP;
    const float* rsrc = logits + g0 * NP;
    float* dsrc = dmk + g0 * NP;
    for (int i = threadIdx.x; i < n; i += 256) { lr[i] = rsrc[i]; ld[i] = dsrc[i]; }
    __syncthreads();
    if ((int)threadIdx.x < ng) {
        const float* r = lr + threadIdx.x * NP;
        float* d = ld + threadIdx.x * NP;
        float mx = r[0];
        for (int u = 1; u < NP; ++u) mx = fmaxf(mx, r[u]);
        float e[12], sum = 0.f;
#pragma unroll
        for (int u = 0; u < 12; ++u) { e[u] = u < NP ? expf(r[u] - mx) : 0.f; sum += e[u]; }
        const float inv = 1.0f / sum;
        float dot = 0.f;
#pragma unroll
        for (int u = 0; u < 12; ++u) if (u < NP) { e[u] *= inv; dot = fmaf(e[u], d[u], dot); }
#pragma unroll
        for (int u = 0; u < 12; ++u) if (u < NP) d[u] = r[u] > 0.f ? e[u] * (d[u] - dot) : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) dsrc[i] = ld[i];
}
int mask_softmax_bwd(const float* logits, float* dmk, int B, int HW, int NP, hipStream_t s) {
    PIVP_CHECK_ARG(logits && dmk && B > 0 && HW > 0 && NP >= 2 && NP <= 12);
    const long ng = (long)B * HW;          // B*NP*HW / NP groups
    hipLaunchKernelGGL(mask_softmax_bwd_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, s, logits, dmk, ng, NP);
    return PIVP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
// 1x1 heads backward (forward heads_1x1_kernel; masks TM:718, enc7 TM:315): with d pre-activation planes dp[o]
// (o < NP: masks after mask_softmax_bwd; o >= NP: enc7 dz):
//   d e6[pix][k] = sum_o W[k][o] dp[o][pix];  dW[k][o] += sum_pix e6[pix][k] dp[o][pix];  db[o] += sum_pix dp[o][pix]
// One block = 128 pixels.  dW/db through atomics (one [64][NO] tile per block).
// ------------------------------------------------------------------------------------------
constexpr int HB_PX = 128;       // pixels per sub-tile
constexpr int HB_SUB = 4;        // sub-tiles per block: dW / db are accumulated in registers over them, so each of the 910 gradient
                                 // addresses receives one atomic per 512 pixels (1024 blocks x 910 atomics on the same addresses made
                                 // the first version an L2-atomic queue).  (2, i.e. two blocks per CU at B = 32, measured slower.)
constexpr int HB_MAXOUT = 32;    // outputs (mask planes + enc7 planes): one 32-wide MFMA tile
constexpr int HB_DP = 129;       // pitch of the [output][pixel] tile: lanes that walk the outputs at one pixel hit 32 different banks
// Both contractions run on the fp32 matrix cores (v_mfma_f32_32x32x2_f32): these 1x1 channel mixes are the dense contractions the
// reference's heads consist of (TM:718, TM:315).  Wave w of the block owns pixels [32w, 32w + 32) of a 128-pixel sub-tile:
//   d e6 [32 px][64 k]  = dp^T [32 px][NO] . W^T [NO][64]       K = NO (zero-padded to even): A = dpt, B = wl, two 32-channel tiles
//   dW   [64 k][NO]    += e6^T [64 k][32 px] . dp [32 px][NO]    K = the wave's 32 pixels: A = xt, B = dpt (rows >= NO are zero)
// The scalar version spent 63 us per launch in LDS reads (one per FMA); this one is bound by reading e6 and writing d e6 (67 MB).
__global__ __launch_bounds__(256) void heads_bwd_kernel(const float* __restrict__ e6, const float* __restrict__ wm, const float* __restrict__ we,
                                                        const float* __restrict__ dpm, const float* __restrict__ dpe,
                                                        float* __restrict__ de6, float* __restrict__ dwm, float* __restrict__ dbm,
                                                        float* __restrict__ dwe, float* __restrict__ dbe, int total_px, int HW, int NP, int NE) {
    PIVP_SET_MAIN_PRIO();
    __shared__ __attribute__((aligned(16))) float xt[HB_PX * 68];        // [pixel][64 + 4]; reused for the block reduction of dW
    __shared__ float dpt[HB_MAXOUT * HB_DP];                              // [output][pixel]
    __shared__ float wl[HB_MAXOUT * 64];                                  // [output][k]
    const int tid = threadIdx.x, NO = NP + NE, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {   // the weight table: all eight loads of a thread go out before the first LDS store (a loop of lds[i] = g[i] is one L2 round trip per trip)
        float wv[HB_MAXOUT * 64 / 256];
#pragma unroll
        for (int u = 0; u < HB_MAXOUT * 64 / 256; ++u) {
            const int i = tid + 256 * u, o = i >> 6, k = i & 63, oc = min(o, NO - 1);
            const float v = oc < NP ? wm[k * NP + oc] : we[k * NE + (oc - NP)];
            wv[u] = o < NO ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < HB_MAXOUT * 64 / 256; ++u) wl[tid + 256 * u] = wv[u];
    }
    for (int i = tid; i < HB_MAXOUT * HB_DP; i += 256) dpt[i] = 0.f;      // rows >= NO stay zero for the whole block
    const int KO = (NO + 1) >> 1;                                         // k-steps of the d e6 contraction
    f32x16 dwacc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dwacc[t][r] = 0.f;
    float dbacc = 0.f;
    // Register double buffer: the loads of sub-tile s + 1 (8 x 16 B of e6 + up to 16 plane elements per thread) are issued before sub-tile
    // s is multiplied and land under its MFMAs and stores.  (The first version loaded, stored to LDS, synchronised and multiplied each
    // sub-tile in turn with four waves on the CU: four exposed round trips per sub-tile, 38 us per launch for 67 MB.)
    f32x4 tx[8];
    float td[HB_MAXOUT / 2];
    const int sp = tid & (HB_PX - 1), oh = tid >> 7;                       // staging role for the planes: pixel sp, outputs oh, oh + 2, ...
    auto load_sub = [&](int sub) {
        const int px0 = (blockIdx.x * HB_SUB + sub) * HB_PX;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = tid + 256 * u, p = f >> 4, cv = (f & 15) * 4;
            tx[u] = px0 + p < total_px ? *reinterpret_cast<const f32x4*>(e6 + (size_t)(px0 + p) * 64 + cv) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int px = px0 + sp;
        const bool pok = px < total_px;
        const int bb = pok ? px / HW : 0, q = px - bb * HW;
#pragma unroll
        for (int u = 0; u < HB_MAXOUT / 2; ++u) {
            const int o = oh + 2 * u;
            td[u] = (pok && o < NO) ? (o < NP ? dpm[((size_t)bb * NP + o) * HW + q] : dpe[((size_t)bb * NE + (o - NP)) * HW + q]) : 0.f;
        }
    };
    load_sub(0);
    for (int sub = 0; sub < HB_SUB; ++sub) {
        const int px0 = (blockIdx.x * HB_SUB + sub) * HB_PX;
        if (px0 >= total_px) break;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = tid + 256 * u, pp = f >> 4, cv = (f & 15) * 4;
            *reinterpret_cast<f32x4*>(xt + pp * 68 + cv) = tx[u];
        }
#pragma unroll
        for (int u = 0; u < HB_MAXOUT / 2; ++u) { const int o = oh + 2 * u; if (o < NO) dpt[o * HB_DP + sp] = td[u]; }
        __syncthreads();
        if (sub + 1 < HB_SUB) load_sub(sub + 1);                           // (past the end of the image: every load is predicated off)
        {   // d e6 of this wave's 32 pixels: row i of the MFMA tile = pixel, column j = channel
            f32x16 c0, c1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
            for (int ks = 0; ks < KO; ++ks) {
                const int o = 2 * ks + half;
                const float a = dpt[o * HB_DP + 32 * wave + l31];
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wl[o * 64 + l31], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wl[o * 64 + 32 + l31], c1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = px0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (px < total_px) {                       // a half-wave writes 128 contiguous bytes per tile
                    de6[(size_t)px * 64 + l31] = c0[r];
                    de6[(size_t)px * 64 + 32 + l31] = c1[r];
                }
            }
        }
        // dW partial over this wave's 32 pixels: row i = input channel, column j = output (pixels past total_px were staged as zeros)
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
            const int p = 32 * wave + 2 * ks + half;
            const float b = dpt[l31 * HB_DP + p];
            dwacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xt[p * 68 + l31], b, dwacc[0], 0, 0, 0);
            dwacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xt[p * 68 + 32 + l31], b, dwacc[1], 0, 0, 0);
        }
        {   // column sums of dp (bias gradients): eight threads per output take 16 pixels each and meet through DPP (before: one thread per output
            // walked its 128 pixels, a serial chain of 128 LDS reads in one wave while the other three waited at the next barrier)
            const int o = tid >> 3, q8 = tid & 7;
            float a = 0.f;
#pragma unroll
            for (int p = 0; p < HB_PX / 8; ++p) a += dpt[o * HB_DP + q8 * (HB_PX / 8) + p];      // rows >= NO are zero
            a += wave_xor4(a);
            a += wave_dpp<0x4e>(a, a);
            a += wave_dpp<0xb1>(a, a);
            dbacc += a;          // every thread of the eight holds output o's sum
        }
    }
    // block reduction of the four waves' dW tiles through LDS ([wave][k tile][k][33]), then one atomic per (k, output)
    __syncthreads();
    float* red = xt;                                   // 4 * 2 * 32 * 33 floats = 33.8 KB <= 34.8 KB
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * 2 + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 33 + l31] = dwacc[t][r];
    __syncthreads();
    for (int e = tid; e < 64 * NO; e += 256) {
        const int k = e / NO, o = e - k * NO;
        const int idx = ((k >> 5) * 32 + (k & 31)) * 33 + o;
        const float v = (red[idx] + red[2 * 32 * 33 + idx]) + (red[4 * 32 * 33 + idx] + red[6 * 32 * 33 + idx]);
        if (o < NP) atomicAdd(dwm + k * NP + o, v); else atomicAdd(dwe + k * NE + (o - NP), v);
    }
    if ((tid & 7) == 0 && (tid >> 3) < NO) { const int o = tid >> 3; if (o < NP) atomicAdd(dbm + o, dbacc); else atomicAdd(dbe + (o - NP), dbacc); }
}
int heads_bwd(const float* e6, const float* wm, const float* we, const float* dpm, const float* dpe, float* de6,
              float* dwm, float* dbm, float* dwe, float* dbe, int B, int HW, int NP, int NE, hipStream_t s) {
    PIVP_CHECK_ARG(e6 && wm && we && dpm && dpe && de6 && dwm && dbm && dwe && dbe && B > 0 && HW > 0 && NP + NE <= HB_MAXOUT);
    const int total = B * HW;
    hipLaunchKernelGGL(heads_bwd_kernel, dim3((total + HB_PX * HB_SUB - 1) / (HB_PX * HB_SUB)), dim3(256), 0, s, e6, wm, we, dpm, dpe, de6, dwm, dbm, dwe, dbe,
                       total, HW, NP, NE);
    return PIVP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
// CDNA kernel generator backward (forward TM:321-329): v = W x + b, u = relu(v - 1e-12) + 1e-12, k = u / sum_25 u.
//   dk from the composite partials (the last generated kernel never reaches the output: zero gradient, TM:726)
//   du = (dk - sum_j dk_j k_j) / S,  dv = du [v - 1e-12 > 0]
//   d x[b][kk] = sum_o Wt[kk][o] dv[b][o];  dWt[kk][o] += sum_b x[b][kk] dv[b][o];  db[o] += sum_b dv[b][o]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cdna_kernels_bwd_dv_kernel(const float* __restrict__ vpre, const float* __restrict__ dkpart, int ntiles,
                                                                  float* __restrict__ dv, float* __restrict__ db, int NM) {
    PIVP_SET_MAIN_PRIO();
    __shared__ float u[256], dk[256];
    const int b = blockIdx.x, o = threadIdx.x, nout = NM * 25, nused = (NM - 1) * 25;
    float uu = 0.f, d = 0.f;
    if (o < nout) {
        uu = fmaxf(vpre[(size_t)b * 256 + o] - 1e-12f, 0.f) + 1e-12f;
        if (o < nused) for (int t = 0; t < ntiles; ++t) d += dkpart[((size_t)b * ntiles + t) * 256 + o];
    }
    u[o] = uu; dk[o] = d;
    __syncthreads();
    float out = 0.f;
    if (o < nout) {
        const int g = (o / 25) * 25;
        float S = 0.f, dot = 0.f;
#pragma unroll
        for (int i = 0; i < 25; ++i) S += u[g + i];
#pragma unroll
        for (int i = 0; i < 25; ++i) dot = fmaf(dk[g + i], u[g + i], dot);
        const float du = (d - dot / S) / S;
        out = (vpre[(size_t)b * 256 + o] - 1e-12f > 0.f) ? du : 0.f;
        atomicAdd(db + o, out);
    }
    dv[(size_t)b * 256 + o] = out;
}

// d x[b][kk] = sum_o Wt[kk][o] dv[b][o]; block = 32 rows kk, all b (<= 32 per pass)
__global__ __launch_bounds__(256) void skinny_linear_bwd_x_kernel(const float* __restrict__ wt, const float* __restrict__ dv,
                                                                  float* __restrict__ dx, int B, int K, int accum) {
    PIVP_SET_MAIN_PRIO();
    __shared__ float wl[32 * 257];
    __shared__ float dl[32 * 257];
    const int k0 = blockIdx.x * 32, b0 = blockIdx.y * 32, tid = threadIdx.x;
    {   // 2 x 8 independent 16-B loads per thread, all issued before the first LDS store (the element loop was one L2 round trip per element)
        f32x4 tw[8], td[8];
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = (tid + 256 * u) * 4, r = f >> 8, o = f & 255;
            tw[u] = (k0 + r < K) ? *reinterpret_cast<const f32x4*>(wt + (size_t)(k0 + r) * 256 + o) : z4;
            td[u] = (b0 + r < B) ? *reinterpret_cast<const f32x4*>(dv + (size_t)(b0 + r) * 256 + o) : z4;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = (tid + 256 * u) * 4, r = f >> 8, o = f & 255;
#pragma unroll
            for (int e = 0; e < 4; ++e) { wl[r * 257 + o + e] = tw[u][e]; dl[r * 257 + o + e] = td[u][e]; }
        }
    }
    __syncthreads();
    const int kk = tid & 31, bg = tid >> 5;       // 8 groups of 4 samples
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int o = 0; o < 256; ++o) {
        const float w = wl[kk * 257 + o];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(w, dl[(bg * 4 + j) * 257 + o], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = b0 + bg * 4 + j;
        if (b < B && k0 + kk < K) {
            float* p = dx + (size_t)b * K + k0 + kk;
            *p = accum ? *p + acc[j] : acc[j];
        }
    }
}

// dWt[kk][o] += sum_b x[b][kk] dv[b][o]; block = 8 rows kk x 256 columns
__global__ __launch_bounds__(256) void skinny_linear_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ dv,
                                                                  float* __restrict__ dwt, int B, int K) {
    __shared__ float xl[8 * 33];
    const int k0 = blockIdx.x * 8, o = threadIdx.x;
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = 0.f;
    for (int b0 = 0; b0 < B; b0 += 32) {
        __syncthreads();
        {
            const int r = threadIdx.x >> 5, bb = threadIdx.x & 31;
            xl[r * 33 + bb] = (b0 + bb < B && k0 + r < K) ? x[(size_t)(b0 + bb) * K + k0 + r] : 0.f;
        }
        __syncthreads();
        float dvr[32];      // the 32 samples' dv of this column: requested together (inside the loop below each was a round trip of its own)
#pragma unroll
        for (int bb = 0; bb < 32; ++bb) dvr[bb] = dv[(size_t)min(b0 + bb, B - 1) * 256 + o];
#pragma unroll
        for (int bb = 0; bb < 32; ++bb) {
            const float d = b0 + bb < B ? dvr[bb] : 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = fmaf(xl[r * 33 + bb], d, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if (k0 + r < K) dwt[(size_t)(k0 + r) * 256 + o] += acc[r];
}

int cdna_kernels_bwd(const float* hidden5, const float* wt, const float* vpre, const float* dkpart, int ntiles, float* dv,
                     float* dhidden5, int accum_dx, float* dwt, float* db, int B, int K, int NM, hipStream_t s, const SideFork* fork) {
    PIVP_CHECK_ARG(hidden5 && wt && vpre && dkpart && dv && dhidden5 && dwt && db && B > 0 && K > 0 && NM >= 1 && NM * 25 <= 256);
    hipLaunchKernelGGL(cdna_kernels_bwd_dv_kernel, dim3(B), dim3(256), 0, s, vpre, dkpart, ntiles, dv, db, NM);
    hipLaunchKernelGGL(skinny_linear_bwd_x_kernel, dim3((K + 31) / 32, (B + 31) / 32), dim3(256), 0, s, wt, dv, dhidden5, B, K, accum_dx);
    hipStream_t sw = s;      // the weight gradient needs dv only: on the fork's stream it runs beside the rest of the sweep
    if (fork && fork->side) {
        // `ready` sits behind skinny_linear_bwd_x too: one kernel later than strictly needed, and the side stream never reads dv early
        if (hipEventRecord(fork->ready, s) != hipSuccess || hipStreamWaitEvent(fork->side, fork->ready, 0) != hipSuccess) return PIVP_ERR_LAUNCH;
        sw = fork->side;
    }
    hipLaunchKernelGGL(skinny_linear_bwd_w_kernel, dim3((K + 7) / 8), dim3(256), 0, sw, hidden5, dv, dwt, B, K);
    if (fork && fork->side && hipEventRecord(fork->done, fork->side) != hipSuccess) return PIVP_ERR_LAUNCH;
    return PIVP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
// composite backward, STP (forward: composite_kernel<1>, TM:465-471 + TM:720-728):
//   out = mk0*prev + mk1*L0 + (sum_{q>=2} mk_q) * warp,  warp = bilinear(prev; theta),  L0 = sigmoid(z)  (no ReLU, TM:454-455)
//   d mk0 = sum_c go*prev, d mk1 = sum_c go*L0, d mk_{q>=2} = sum_c go*warp;  d z = go*mk1*L0(1-L0)
//   d theta: through the sampling coordinates (u, v) = ((g+1)(W-1)/2), g = theta . (xs, ys, 1); zero where the coordinate was
//            clamped (stp 'clamp' mode) -> per-tile partial sums dthpart[b][tile][6]
//   d prev (feed-self): mk0*go plus the bilinear weights scattered to the 4 neighbours (atomics into a buffer the caller has
//            initialised with the loss term)
// ------------------------------------------------------------------------------------------
// threads per block: one pixel per thread on 8 x 64 tiles, two waves per SIMD (256 threads: 224 us per launch at B = 32)
constexpr int CBS_NT = 512;
constexpr int CBS_R = 12;        // rows above / below the tile held in the LDS window of d prev
// `whole` (feed-self sweeps on frames whose three planes fit in LDS: 48 KB at 64 x 64): a block's window of d prev is the WHOLE frame, so
// every scattered bilinear weight is an LDS atomic whatever theta is (with the +-12-row window a random-init theta, far from the identity,
// sent most of them to global atomics: 194 us per launch).  A sample's tiles are shared by gridDim.x blocks (tile, tile + gridDim.x, ...),
// each of which adds its frame to d prev once at the end: one coalesced atomic per touched element (plain adds when gridDim.x == 1).

template <int CB_TR>
__global__ __launch_bounds__(CBS_NT) void composite_bwd_stp_kernel(const float* __restrict__ prev, const float* __restrict__ logits,
                                                                const float* __restrict__ layer0, const float* __restrict__ theta,
                                                                const float* __restrict__ go, float* __restrict__ dmk, float* __restrict__ dz,
                                                                float* __restrict__ dthpart, float* __restrict__ dprev,
                                                                int H, int W, int NM, int stp_zero, int whole) {
    PIVP_SET_MAIN_PRIO();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[CBS_NT / 64][6];
    const int NP = NM + 1, HW = H * W;
    const int b = blockIdx.y;
    const int ntiles = (H + CB_TR - 1) / CB_TR;
    const int tid = threadIdx.x;
    const int WR = whole ? H : CB_TR + 2 * CBS_R;
    const int np_max = CB_TR * W, win_max = np_max + 2 * (NP - 1), G_max = np_max / NP + 2;
    float* lg = sm;                              // [NP][win]
    float* gmx = lg + NP * win_max;              // [NP][G]
    float* ginv = gmx + NP * G_max;              // [NP][G]
    float* dwin = ginv + NP * G_max;             // [3][WR][W]  this block's window of d prev (feed-self only)
    if (dprev)
        for (int i = tid; i < 3 * WR * W; i += CBS_NT) dwin[i] = 0.f;
    const float* lgb = logits + (size_t)b * NP * HW;
    const unsigned magic = 0xFFFFFFFFu / (unsigned)NP + 1u;        // exact x / NP for x * NP < 2^32 (as in composite_bwd_cdna_kernel)
    auto div_np = [&](int x) { return (int)__umulhi((unsigned)x, magic); };
  for (int tile = blockIdx.x; tile < (whole ? ntiles : (int)blockIdx.x + 1); tile += gridDim.x) {
    const int y0 = tile * CB_TR;
    const int rows = min(CB_TR, H - y0);
    const int p0 = y0 * W, np = rows * W;
    const int win = np + 2 * (NP - 1), G = np / NP + 2;
    const int wy0 = whole ? 0 : y0 - CBS_R;
    if (tile != (int)blockIdx.x) __syncthreads();      // the previous tile's readers of lg / gmx / red are done
    for (int i = tid; i < NP * win; i += CBS_NT) {
        const int m = i / win, j = i - m * win;
        const int F = m * HW + p0 - (NP - 1) + j;
        lg[i] = (F >= 0 && F < NP * HW) ? lgb[F] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < NP * G; i += CBS_NT) {
        const int m = i / G, gi = i - m * G;
        const int gfirst = div_np(m * HW + p0), glast = div_np(m * HW + p0 + np - 1);
        if (gfirst + gi <= glast) {
            const float* e = lg + m * win + (gfirst + gi) * NP - (m * HW + p0 - (NP - 1));
            float mx = e[0];
            for (int u = 1; u < NP; ++u) mx = fmaxf(mx, e[u]);
            float sum = 0.f;
            for (int u = 0; u < NP; ++u) sum += expf(e[u] - mx);
            gmx[i] = mx; ginv[i] = 1.0f / sum;
        }
    }
    __syncthreads();
    const float* th = theta + (size_t)b * 6;
    const float* pb = prev + (size_t)b * 3 * HW;
    float dth[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int pp = tid; pp < np; pp += CBS_NT) {
        const int p = p0 + pp, y = p / W, x = p - y * W;
        float mk0 = 0.f, mk1 = 0.f, msum = 0.f;
        for (int m = 0; m < NP; ++m) {
            const int gi = div_np(m * HW + p) - div_np(m * HW + p0);
            const float v = expf(lg[m * win + pp + (NP - 1)] - gmx[m * G + gi]) * ginv[m * G + gi];
            if (m == 0) mk0 = v; else if (m == 1) mk1 = v; else msum += v;
        }
        const double xs = -1.0 + 2.0 * (double)x / (double)(W - 1), ys = -1.0 + 2.0 * (double)y / (double)(H - 1);
        double gu = (double)th[0] * xs + (double)th[1] * ys + (double)th[2];
        double gv = (double)th[3] * xs + (double)th[4] * ys + (double)th[5];
        bool live_u = true, live_v = true;
        if (!stp_zero) {
            live_u = gu > -1.0 && gu < 1.0; live_v = gv > -1.0 && gv < 1.0;
            gu = fmin(fmax(gu, -1.0), 1.0); gv = fmin(fmax(gv, -1.0), 1.0);
        }
        const double u = (gu + 1.0) * (double)(W - 1) * 0.5, v = (gv + 1.0) * (double)(H - 1) * 0.5;
        double u0 = floor(u), v0 = floor(v);
        if (!stp_zero) { u0 = fmin(fmax(u0, 0.0), (double)(W - 2)); v0 = fmin(fmax(v0, 0.0), (double)(H - 2)); }
        const float wu1 = (float)(u - u0), wv1 = (float)(v - v0);
        const int iu = (int)u0, iv = (int)v0;
        float dmq = 0.f, dmk0 = 0.f, dmk1 = 0.f, du = 0.f, dvv = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = go[((size_t)b * 3 + c) * HW + p];
            float nb[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int uu = iu + e, vv = iv + a;
                    nb[a][e] = ((unsigned)uu < (unsigned)W && (unsigned)vv < (unsigned)H) ? pb[(size_t)c * HW + vv * W + uu] : 0.f;
                }
            const float warp = (1.f - wv1) * ((1.f - wu1) * nb[0][0] + wu1 * nb[0][1]) + wv1 * ((1.f - wu1) * nb[1][0] + wu1 * nb[1][1]);
            const float l0 = layer0[((size_t)b * 3 + c) * HW + p];
            const float pc = pb[(size_t)c * HW + p];
            dmk0 = fmaf(g, pc, dmk0); dmk1 = fmaf(g, l0, dmk1); dmq = fmaf(g, warp, dmq);
            dz[((size_t)b * 3 + c) * HW + p] = g * mk1 * l0 * (1.f - l0);
            const float dw = g * msum;                                   // d loss / d warp[c](p)
            du = fmaf(dw, (1.f - wv1) * (nb[0][1] - nb[0][0]) + wv1 * (nb[1][1] - nb[1][0]), du);
            dvv = fmaf(dw, (1.f - wu1) * (nb[1][0] - nb[0][0]) + wu1 * (nb[1][1] - nb[0][1]), dvv);
            if (dprev) {
                // scatter into the block's LDS window of d prev (rows wy0 .. wy0 + WR - 1: the tile's own rows +- CBS_R, where a near-identity
                // warp lands); targets outside it go straight to memory.  Straight global atomics for everything cost 215 of this kernel's
                // 231 us: neighbouring pixels' bilinear footprints overlap, so the lanes of one instruction hit the same addresses.
                float* dp = dprev + ((size_t)b * 3 + c) * HW;
                float* dwc = dwin + c * WR * W;
                atomicAdd(dwc + (y - wy0) * W + x, mk0 * g);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int uu = iu + e, vv = iv + a;
                        if ((unsigned)uu < (unsigned)W && (unsigned)vv < (unsigned)H) {
                            const float val = dw * (a ? wv1 : 1.f - wv1) * (e ? wu1 : 1.f - wu1);
                            if ((unsigned)(vv - wy0) < (unsigned)WR) atomicAdd(dwc + (vv - wy0) * W + uu, val);
                            else atomicAdd(dp + vv * W + uu, val);
                        }
                    }
            }
        }
        float* dm = dmk + (size_t)b * NP * HW + p;
        dm[0] = dmk0; dm[(size_t)HW] = dmk1;
        for (int q = 2; q < NP; ++q) dm[(size_t)q * HW] = dmq;
        const float dgu = live_u ? du * (float)(W - 1) * 0.5f : 0.f, dgv = live_v ? dvv * (float)(H - 1) * 0.5f : 0.f;
        dth[0] += dgu * (float)xs; dth[1] += dgu * (float)ys; dth[2] += dgu;
        dth[3] += dgv * (float)xs; dth[4] += dgv * (float)ys; dth[5] += dgv;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) dth[j] = wave_sum(dth[j]);
    if ((tid & 63) == 0) for (int j = 0; j < 6; ++j) red[tid >> 6][j] = dth[j];
    __syncthreads();
    if (dprev && !whole)   // the window's touched elements, one atomic each (other tiles' windows overlap this one)
        for (int i = tid; i < 3 * WR * W; i += CBS_NT) {
            const float v = dwin[i];
            const int c = i / (WR * W), rem = i - c * (WR * W), r = rem / W, xx = rem - r * W, yy = wy0 + r;
            if (v != 0.f && (unsigned)yy < (unsigned)H) atomicAdd(dprev + ((size_t)b * 3 + c) * HW + yy * W + xx, v);
        }
    if (tid < 6) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < CBS_NT / 64; ++w) v += red[w][tid];
        dthpart[((size_t)b * ntiles + tile) * 8 + tid] = v;
    }
  }   // tiles
    if (dprev && whole) {   // the whole frame: plain adds when this block is the sample's only writer in the launch, else one atomic per element
        __syncthreads();
        float* dp = dprev + (size_t)b * 3 * HW;
        if (gridDim.x == 1) { for (int i = tid; i < 3 * HW; i += CBS_NT) dp[i] += dwin[i]; }
        else { for (int i = tid; i < 3 * HW; i += CBS_NT) { const float v = dwin[i]; if (v != 0.f) atomicAdd(dp + i, v); } }
    }
}

int composite_bwd_stp(const float* prev, const float* logits, const float* layer0, const float* theta, const float* go,
                      float* dmk, float* dz, float* dthpart, float* dprev, int B, int H, int W, int NM, int stp_zero, hipStream_t s) {
    PIVP_CHECK_ARG(prev && logits && layer0 && theta && go && dmk && dz && dthpart && B > 0 && H > 1 && W > 1 && NM >= 2 && NM <= 10);
    const int CB_TR = composite_bwd_rows(W);
    const int NP = NM + 1, np = CB_TR * W, win = np + 2 * (NP - 1), G = np / NP + 2;
    constexpr int whole_on = 8;      // at most this many blocks per sample, each with the whole frame's d prev in LDS (B = 32: 29.3 / 29.3 / 28.6 / 28.2 / 28.0 ms per STP
                                     // train step for one block per tile with a +-12-row window / 1 / 2 / 4 / 8)
    const size_t lds_head = sizeof(float) * ((size_t)NP * win + 2 * NP * G);
    const int whole = (whole_on > 0 && dprev && lds_head + sizeof(float) * 3 * (size_t)H * W <= 96 * 1024) ? 1 : 0;
    const size_t lds = lds_head + sizeof(float) * 3 * (size_t)(whole ? H : CB_TR + 2 * CBS_R) * W;
    PIVP_CHECK_ARG(lds <= 150 * 1024);
    const int ntiles = composite_bwd_tiles(H, W);
    int per_sample = ntiles;
    if (whole) {           // blocks per sample: enough to cover the chip about once (every block flushes a whole frame of atomics)
        per_sample = whole_on < ntiles ? whole_on : ntiles;
        while (per_sample > 1 && (long)per_sample * B > 1024) per_sample >>= 1;
    }
    const dim3 grid(per_sample, B);
    if (CB_TR == 8) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&composite_bwd_stp_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(composite_bwd_stp_kernel<8>, grid, dim3(CBS_NT), lds, s, prev, logits, layer0, theta, go, dmk,
                           dz, dthpart, dprev, H, W, NM, stp_zero, whole);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&composite_bwd_stp_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(composite_bwd_stp_kernel<4>, grid, dim3(CBS_NT), lds, s, prev, logits, layer0, theta, go, dmk,
                           dz, dthpart, dprev, H, W, NM, stp_zero, whole);
    }
    return PIVP_LAUNCH_STATUS();
}

// STP regressor backward (forward TM:457-468): theta = W2 s1 + b2 + ident, s1 = relu(W1 x + b1).
//   d theta = sum of the tile partials; dW2 += d theta (x) s1; db2 += d theta; dv = (W2^T d theta) [s1 > 0]; db1 += dv
__global__ __launch_bounds__(128) void stp_params_bwd_kernel(const float* __restrict__ dthpart, int ntiles, const float* __restrict__ s1,
                                                             const float* __restrict__ w2, float* __restrict__ dw2, float* __restrict__ db2,
                                                             float* __restrict__ db1, float* __restrict__ dv) {
    PIVP_SET_MAIN_PRIO();
    __shared__ float dth[6];
    const int b = blockIdx.x, o = threadIdx.x;
    if (o < 6) {
        float a = 0.f;
        for (int t = 0; t < ntiles; ++t) a += dthpart[((size_t)b * ntiles + t) * 8 + o];
        dth[o] = a;
        atomicAdd(db2 + o, a);
    }
    __syncthreads();
    float out = 0.f;
    if (o < 100) {
        const float sv = s1[(size_t)b * 256 + o];
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) { d = fmaf(w2[j * 100 + o], dth[j], d); atomicAdd(dw2 + j * 100 + o, dth[j] * sv); }
        out = sv > 0.f ? d : 0.f;
        atomicAdd(db1 + o, out);
    }
    dv[(size_t)b * 256 + o] = out;
    dv[(size_t)b * 256 + 128 + o] = 0.f;
}

int stp_params_bwd(const float* hidden5, const float* wt1, const float* s1, const float* w2, const float* dthpart, int ntiles, float* dv,
                   float* dhidden5, float* dwt1, float* db1, float* dw2, float* db2, int B, int K, hipStream_t s) {
    PIVP_CHECK_ARG(hidden5 && wt1 && s1 && w2 && dthpart && dv && dhidden5 && dwt1 && db1 && dw2 && db2 && B > 0 && K > 0);
    hipLaunchKernelGGL(stp_params_bwd_kernel, dim3(B), dim3(128), 0, s, dthpart, ntiles, s1, w2, dw2, db2, db1, dv);
    hipLaunchKernelGGL(skinny_linear_bwd_x_kernel, dim3((K + 31) / 32, (B + 31) / 32), dim3(256), 0, s, wt1, dv, dhidden5, B, K, 0);
    hipLaunchKernelGGL(skinny_linear_bwd_w_kernel, dim3((K + 7) / 8), dim3(256), 0, s, hidden5, dv, dwt1, B, K);
    return PIVP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
// enc3 (smear + 1x1 + ReLU) and state predictor backward (forward enc3_state_kernel; TM:556-567, 503, 730).
//   dpre = d e3 [e3 > 0];  d e2 = dpre W3x^T;  dW3x += e2^T dpre;  db3 += colsum(dpre);
//   dW3s[j] += sa[j] colsum;  d sa[j] = W3s[j] . colsum + sum_o Wcs[o][j] d snew[o];  dWcs, dbcs from d snew.
//   d state_prev = d sa[5:10]  (the predicted state is fed back, TM:676)
// One block per (64-pixel tile, sample); small sums through atomics.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void enc3_state_bwd_kernel(const float* __restrict__ e2, const float* __restrict__ e3, const float* __restrict__ de3, int ldd3,
                                                             const float* __restrict__ action, const float* __restrict__ state,
                                                             const float* __restrict__ w3, const float* __restrict__ wcs,
                                                             const float* __restrict__ dsnew, float* __restrict__ de2,
                                                             float* __restrict__ dw3, float* __restrict__ db3, float* __restrict__ dwcs,
                                                             float* __restrict__ dbcs, float* __restrict__ dstate_prev,
                                                             int HW8, int use_state, int mask_e2) {
    PIVP_SET_MAIN_PRIO();
    __shared__ float xt[64 * 65];
    __shared__ float dt[64 * 65];
    __shared__ __attribute__((aligned(16))) float wl[64 * 64];
    __shared__ float colsum[64], sa[10], dsa[10];
    const int b = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    if (tid < 5) sa[tid] = action[b * 5 + tid]; else if (tid < 10) sa[tid] = state[b * 5 + tid - 5];
    if (tid < 10) dsa[tid] = 0.f;
    const int npx = min(64, HW8 - tile * 64);
    const size_t base = ((size_t)b * HW8 + tile * 64) * 64;
    {   // sixteen independent 16-B loads per thread, then the LDS stores (the element loops made ~48 serial round trips)
        f32x4 tw[4], tx[4], ty[4], td[4];
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = (tid + 256 * j) * 4, pp = f >> 6, k = f & 63;
            const bool ok = pp < npx;
            tw[j] = *reinterpret_cast<const f32x4*>(w3 + f);
            tx[j] = ok ? *reinterpret_cast<const f32x4*>(e2 + base + f) : z4;
            ty[j] = ok ? *reinterpret_cast<const f32x4*>(e3 + base + f) : z4;
            td[j] = ok ? *reinterpret_cast<const f32x4*>(de3 + ((size_t)b * HW8 + tile * 64 + pp) * ldd3 + k) : z4;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = (tid + 256 * j) * 4, pp = f >> 6, k = f & 63;
            *reinterpret_cast<f32x4*>(wl + f) = tw[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xt[pp * 65 + k + e] = tx[j][e];
                dt[pp * 65 + k + e] = ty[j][e] > 0.f ? td[j][e] : 0.f;
            }
        }
    }
    __syncthreads();
    // blockIdx.z = quarter q: 16 of the tile's 64 pixels for d e2 and 16 of the 64 input channels for dW (every quarter stages the whole
    // tile: 48 KB from L2).  One block per (tile, sample) ran 2 x 64 iterations of 16 FMAs on 32 of 256 CUs: 41 us for 17 MFLOP.
    const int q = blockIdx.z;
    if (tid < 64) {
        float c = 0.f;
        for (int p = 0; p < 64; ++p) c += dt[p * 65 + tid];
        colsum[tid] = c;
        if (q == 0) atomicAdd(db3 + tid, c);
    }
    {   // d e2[p][ci] = sum_co w3[ci][co] dpre[p][co]: thread (p = 16 q + tid/16, 4 ci)
        const int p = q * 16 + (tid >> 4), cg = (tid & 15) * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int co = 0; co < 64; ++co) {
            const float d = dt[p * 65 + co];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(wl[(cg + i) * 64 + co], d, acc[i]);
        }
        if (mask_e2) {   // e2 = relu(enc2's conv): hand enc2's backward the gradient of the PRE-activation (no relu_mask launch there)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = xt[p * 65 + cg + i] > 0.f ? acc[i] : 0.f;
        }
        if (p < npx) *reinterpret_cast<f32x4*>(de2 + base + (size_t)p * 64 + cg) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    }
    {   // dW3x[ci][co] += sum_p e2[p][ci] dpre[p][co]: thread (ci = 16 q + tid/16, 4 co)
        const int ci = q * 16 + (tid >> 4), cg = (tid & 15) * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < 64; ++p) {
            const float xv = xt[p * 65 + ci];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(xv, dt[p * 65 + cg + i], acc[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(dw3 + ci * 64 + cg + i, acc[i]);
    }
    if (q != 0) return;      // the state predictor and the smeared action/state rows: once per (tile, sample)
    __syncthreads();
    if (use_state && tid < 64) {
#pragma unroll
        for (int j = 0; j < 10; ++j) atomicAdd(dw3 + (64 + j) * 64 + tid, sa[j] * colsum[tid]);
    }
    if (tid < 10) {
        float v = 0.f;
        if (use_state) for (int co = 0; co < 64; ++co) v = fmaf(w3[(64 + tid) * 64 + co], colsum[co], v);
        if (tile == 0) for (int o = 0; o < 5; ++o) v = fmaf(wcs[o * 10 + tid], dsnew[b * 5 + o], v);
        if (tid >= 5) atomicAdd(dstate_prev + b * 5 + tid - 5, v);
    }
    if (tile == 0 && tid >= 64 && tid < 64 + 50) {
        const int o = (tid - 64) / 10, j = (tid - 64) % 10;
        atomicAdd(dwcs + o * 10 + j, dsnew[b * 5 + o] * sa[j]);
        if (j == 0) atomicAdd(dbcs + o, dsnew[b * 5 + o]);
    }
}
int enc3_state_bwd(const float* e2, const float* e3, const float* de3, int ldd3, const float* action, const float* state, const float* w3,
                   const float* wcs, const float* dsnew, float* de2, float* dw3, float* db3, float* dwcs, float* dbcs,
                   float* dstate_prev, int B, int HW8, int use_state, hipStream_t s, int mask_e2) {
    PIVP_CHECK_ARG(e2 && e3 && de3 && action && state && w3 && wcs && dsnew && de2 && dw3 && db3 && dwcs && dbcs && dstate_prev && B > 0 && HW8 > 0);
    hipLaunchKernelGGL(enc3_state_bwd_kernel, dim3((HW8 + 63) / 64, B, 4), dim3(256), 0, s, e2, e3, de3, ldd3, action, state, w3, wcs, dsnew, de2,
                       dw3, db3, dwcs, dbcs, dstate_prev, HW8, use_state, mask_e2);
    return PIVP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------
// enc0 backward (forward conv_enc0_kernel, TM:500): dW[k][co] += sum_pix patch[pix][k] d[pix][co], db, and
// (feed-self) d img[c](y,x) = sum over taps with matching parity.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void enc0_wgrad_kernel(const float* __restrict__ img, const float* __restrict__ d, float* __restrict__ dw,
                                                         float* __restrict__ db, int B, int H, int W) {
    __shared__ float pt[64 * 76];
    __shared__ float dtl[64 * 33];
    const int H2 = H >> 1, W2 = W >> 1, total = B * H2 * W2, tid = threadIdx.x;
    const int co = tid & 31, kg = tid >> 5;          // k = kg, kg+8, ...
    float acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.f;
    float bacc = 0.f;
    for (int t0 = blockIdx.x * 64; t0 < total; t0 += gridDim.x * 64) {
        __syncthreads();
        for (int i = tid; i < 64 * 75; i += 256) {
            const int p = i / 75, k = i - p * 75;
            const int pix = t0 + p;
            float v = 0.f;
            if (pix < total) {
                const int b = pix / (H2 * W2), rem = pix - b * H2 * W2, oy = rem / W2, ox = rem - oy * W2;
                const int tap = k / 3, ci = k - tap * 3, ky = tap / 5, kx = tap - ky * 5;
                const int iy = 2 * oy - 2 + ky, ix = 2 * ox - 2 + kx;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = img[((size_t)b * 3 + ci) * H * W + iy * W + ix];
            }
            pt[p * 76 + k] = v;
        }
        for (int i = tid; i < 64 * 32; i += 256) {
            const int p = i >> 5, c = i & 31;
            dtl[p * 33 + c] = t0 + p < total ? d[(size_t)(t0 + p) * 32 + c] : 0.f;
        }
        __syncthreads();
        for (int p = 0; p < 64; ++p) {
            const float dv = dtl[p * 33 + co];
#pragma unroll
            for (int i = 0; i < 10; ++i) { const int k = kg + 8 * i; if (k < 75) acc[i] = fmaf(pt[p * 76 + k], dv, acc[i]); }
            if (kg == 0) bacc += dv;
        }
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) { const int k = kg + 8 * i; if (k < 75) atomicAdd(dw + k * 32 + co, acc[i]); }
    if (kg == 0) atomicAdd(db + co, bacc);
}

// d img[c](y, x) = sum over the taps (ky, kx) with y + 2 - ky and x + 2 - kx even of sum_co w[tap][c][co] d[(y + 2 - ky) / 2, (x + 2 - kx) / 2][co].
// Which taps contribute depends on the PARITY of (y, x) only: 9 / 6 / 6 / 4 of the 25.  The first version gave a thread one pixel in raster
// order, so the four parity classes of a wave each walked all 25 taps under their own exec mask and every lane fetched its own 128-B row of d
// per tap (51 us per launch at B = 32, on the sweep's critical path).  Here blockIdx.y is the parity class (the tap set is block-uniform:
// no divergence), and a pixel is shared by FOUR lanes that take 8 of the 32 channels each (a wave reads 16 whole 128-B rows per tap) and
// meet in two xor-shuffles.
__global__ __launch_bounds__(256) void enc0_dgrad_kernel(const float* __restrict__ d, const float* __restrict__ w, float* __restrict__ dimg,
                                                         int accum, int B, int H, int W) {
    PIVP_SET_MAIN_PRIO();
    __shared__ __attribute__((aligned(16))) float wl[75 * 32];
    {   // 2,400 weights: ten loads per thread requested together, then the LDS stores (the loop lds[i] = g[i] was ten serial L2 round trips
        // in front of a 14-us kernel on the sweep's critical path)
        float wv[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) wv[u] = w[min((int)threadIdx.x + 256 * u, 75 * 32 - 1)];
#pragma unroll
        for (int u = 0; u < 10; ++u) { const int i = threadIdx.x + 256 * u; if (i < 75 * 32) wl[i] = wv[u]; }
    }
    __syncthreads();
    const int H2 = H >> 1, W2 = W >> 1;
    const int py = blockIdx.y >> 1, px = blockIdx.y & 1;          // parity of (y, x): taps ky = py, py + 2, (py + 4), likewise kx
    const int cg = threadIdx.x & 3;                               // channels 8 cg .. 8 cg + 7
    const long total = (long)B * H2 * W2;                         // pixels of one parity class
    const long q = (long)blockIdx.x * 64 + (threadIdx.x >> 2);
    const bool live = q < total;
    const long qq = live ? q : total - 1;
    const int b = (int)(qq / (H2 * W2)), rem = (int)(qq - (long)b * H2 * W2), yy = rem / W2, xx = rem - yy * W2;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const int nky = py ? 2 : 3, nkx = px ? 2 : 3;
    for (int iy = 0; iy < nky; ++iy) {
        const int ky = py + 2 * iy, oy = yy + 1 - iy;             // (y + 2 - ky) / 2 with y = 2 yy + py
        const bool rowok = (unsigned)oy < (unsigned)H2;
        for (int ix = 0; ix < nkx; ++ix) {
            const int kx = px + 2 * ix, ox = xx + 1 - ix;
            const bool ok = rowok && (unsigned)ox < (unsigned)W2;
            const float* dp = d + ((size_t)(b * H2 + (rowok ? oy : 0)) * W2 + (ok ? ox : 0)) * 32 + cg * 8;   // clamped: always in range
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(dp), d1 = *reinterpret_cast<const f32x4*>(dp + 4);
            const float m = ok ? 1.f : 0.f;
            const float* wr = wl + (ky * 5 + kx) * 3 * 32 + cg * 8;
            float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 dv = h ? d1 : d0;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + 4 * h), w1 = *reinterpret_cast<const f32x4*>(wr + 32 + 4 * h);
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(wr + 64 + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) { t0 = fmaf(w0[e], dv[e], t0); t1 = fmaf(w1[e], dv[e], t1); t2 = fmaf(w2[e], dv[e], t2); }
            }
            a0 = fmaf(m, t0, a0); a1 = fmaf(m, t1, a1); a2 = fmaf(m, t2, a2);
        }
    }
    a0 += __shfl_xor(a0, 1, 64); a1 += __shfl_xor(a1, 1, 64); a2 += __shfl_xor(a2, 1, 64);
    a0 += __shfl_xor(a0, 2, 64); a1 += __shfl_xor(a1, 2, 64); a2 += __shfl_xor(a2, 2, 64);
    if (live && cg == 0) {
        float* o = dimg + (size_t)b * 3 * H * W + (size_t)(2 * yy + py) * W + 2 * xx + px;
        if (accum) { o[0] += a0; o[(size_t)H * W] += a1; o[2 * (size_t)H * W] += a2; }
        else { o[0] = a0; o[(size_t)H * W] = a1; o[2 * (size_t)H * W] = a2; }
    }
}

int enc0_bwd(const float* img, const float* w, const float* d, float* dw, float* db, float* dimg, int dimg_accum, int B, int H, int W,
             hipStream_t s, const SideFork* fork) {
    PIVP_CHECK_ARG(img && w && d && dw && db && B > 0 && H > 0 && W > 0);
    PIVP_CHECK_ARG(!dimg || (H % 2 == 0 && W % 2 == 0));       // the parity-class data gradient: checked before anything is enqueued
    hipStream_t sw = s;
    if (fork && fork->side) {
        if (hipEventRecord(fork->ready, s) != hipSuccess || hipStreamWaitEvent(fork->side, fork->ready, 0) != hipSuccess) return PIVP_ERR_LAUNCH;
        sw = fork->side;
    }
    const int total = B * (H / 2) * (W / 2);
    int blocks = (total + 63) / 64; if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(enc0_wgrad_kernel, dim3(blocks), dim3(256), 0, sw, img, d, dw, db, B, H, W);
    if (fork && fork->side && hipEventRecord(fork->done, fork->side) != hipSuccess) return PIVP_ERR_LAUNCH;
    if (dimg) {
        const long tp = (long)B * (H / 2) * (W / 2);           // pixels per parity class, 64 per block
        hipLaunchKernelGGL(enc0_dgrad_kernel, dim3((unsigned)((tp + 63) / 64), 4), dim3(256), 0, s, d, w, dimg, dimg_accum, B, H, W);
    }
    return PIVP_LAUNCH_STATUS();
}

// out[i] = a[i] + b[i] over n floats with row strides (sum of two gradient contributions into one buffer)
__global__ __launch_bounds__(256) void add_strided_kernel(float* __restrict__ dst, int ldd, const float* __restrict__ src, int lds_, int C, long npix) {
    PIVP_SET_MAIN_PRIO();
    const long total = npix * (C / 4);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / (C / 4); const int c = (int)(i - p * (C / 4)) * 4;
        f32x4 a = *reinterpret_cast<f32x4*>(dst + p * ldd + c);
        a += *reinterpret_cast<const f32x4*>(src + p * lds_ + c);
        *reinterpret_cast<f32x4*>(dst + p * ldd + c) = a;
    }
}
int add_strided(float* dst, int ldd, const float* src, int lds_, int C, long npix, hipStream_t s) {
    PIVP_CHECK_ARG(dst && src && C > 0 && C % 4 == 0 && npix > 0 && ldd % 4 == 0 && lds_ % 4 == 0);
    const long total = npix * (C / 4);
    hipLaunchKernelGGL(add_strided_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, s, dst, ldd, src, lds_, C, npix);
    return PIVP_LAUNCH_STATUS();
}

}  // namespace pivp

PIVP_DEFINE_MAIN_PRIO_SETTER(backward_heads)
