// Implicit-GEMM convolution on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One kernel template serves every dense contraction of the per-timestep program
// (reference call sites: src/models/train_model.py, "TM"):
//   * the seven ConvLSTM 5x5 gate convolutions with the gate math fused into the epilogue
//     (TM:262-272: concat(x,h) -> conv5x5 -> split j,i,f,o -> c,h update)      [95% of FLOPs]
//   * the 3x3 stride-2 convolutions enc1/enc2 (TM:501-502)
//   * the 3x3 stride-2 transposed convolutions enc4/enc5/enc6 (TM:505-507), run as their four
//     sub-pixel phases (1/2/2/4 taps) so no zero-stuffed input is ever touched.
//
// GEMM view: M = anchors (b, ay, ax) of an anchor grid, N = output channels, K = taps x Cin.
// A[m][k] is gathered on the fly from NHWC activations (zero outside the image), B[k][n] is the
// weight matrix stored [tap][Cin][N].  fp32 in / fp32 accumulate: the MFMA result is bit-for-bit a
// k-ordered fmaf chain, so parity with the fp32/fp64 oracle is an ordering question only.
//
// Tiling: 256 threads = 4 waves; wave w owns anchors [32w, 32w+32) x all BN = 32*NT columns
// (NT 32x32 accumulators).  K is consumed in chunks of one tap x 32 input channels, staged
// global -> registers -> LDS with the loads for chunk i+1 issued before the MFMAs of chunk i
// and written after them (one barrier per chunk, two LDS buffers).
//   A tile  [128][36] floats: 144-B rows make the per-lane ds_read_b128 conflict-free
//   B tile  [32][BN]  floats: ds_read_b32, 32 consecutive columns per half-wave
// For the ConvLSTM the four accumulators of a wave are the four gates (j,i,f,o) of the same
// 32 channels, so the epilogue holds j,i,f,o of one (pixel, channel) in one lane and the 4C-wide
// gate tensor is never written.
#include "pivp_kernels.h"

namespace pivp {

constexpr int IG_BM = 128;
constexpr int IG_KC = 32;
constexpr int IG_AP = 36;  // A row pitch in floats

template <int NT>
constexpr int ig_lds_bytes() { return 2 * (IG_BM * IG_AP + IG_KC * 32 * NT) * 4; }

template <int NT, bool LSTM>
__global__ __launch_bounds__(256, 2) void igemm_f32_kernel(const IgemmDesc d) {
    constexpr int BN = 32 * NT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    auto lds_a = [&](int buf) { return lds + buf * (IG_BM * IG_AP); };
    auto lds_b = [&](int buf) { return lds + 2 * IG_BM * IG_AP + buf * (IG_KC * BN); };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int phase = blockIdx.y;
    const int n_nblk = LSTM ? (d.C >> 5) : (d.N / BN);
    const int nblk = blockIdx.x % n_nblk;
    const int mblk = blockIdx.x / n_nblk;
    const int m0 = mblk * IG_BM;
    const int cin = d.c0 + d.c1;
    const int ncc = cin >> 5;
    const int tap0 = d.tap_start[phase];
    const int nchunks = d.tap_count[phase] * ncc;
    const int HWg = d.Hg * d.Wg;

    // ---- staging roles -------------------------------------------------------------------
    const int cvec = tid & 7;   // float4 within the 32-channel chunk
    const int prow = tid >> 3;  // 0..31
    int a_boff[4], a_iy0[4], a_ix0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + prow + 32 * j;
        if (m < d.M) {
            const int b = m / HWg;
            const int rem = m - b * HWg;
            const int ay = rem / d.Wg;
            const int ax = rem - ay * d.Wg;
            a_boff[j] = b * d.Hin * d.Win;
            a_iy0[j] = ay * d.in_step;
            a_ix0[j] = ax * d.in_step;
        } else {
            a_boff[j] = 0;
            a_iy0[j] = -(1 << 20);  // never in range
            a_ix0[j] = 0;
        }
    }
    // B staging: f = tid + 256*j -> row k = f / (BN/4), float4 column cv = f % (BN/4)
    int b_k[NT], b_col[NT];  // global column of the float4
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int f = tid + 256 * j;
        const int k = f / (BN / 4);
        const int cv = (f - k * (BN / 4)) * 4;
        b_k[j] = k;
        if (LSTM) b_col[j] = (cv >> 5) * d.C + nblk * 32 + (cv & 31);
        else      b_col[j] = nblk * BN + cv;
    }

    f32x4 ra[4];
    f32x4 rb[NT];

    auto load_chunk = [&](int it) {
        const int t = tap0 + it / ncc;
        const int cc = it - (it / ncc) * ncc;
        const int dy = d.dy[t], dx = d.dx[t];
        const int ch = cc << 5;
        const float* src;
        int ld, cbase;
        if (ch < d.c0) { src = d.x0; ld = d.ld0; cbase = ch; }
        else           { src = d.x1; ld = d.ld1; cbase = ch - d.c0; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iy = a_iy0[j] + dy;
            const int ix = a_ix0[j] + dx;
            const bool ok = (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const size_t off = (size_t)(a_boff[j] + iy * d.Win + ix) * ld + cbase + cvec * 4;
                v = *reinterpret_cast<const f32x4*>(src + off);
            }
            ra[j] = v;
        }
        const size_t wrow = (size_t)d.wi[t] * cin + ch;
#pragma unroll
        for (int j = 0; j < NT; ++j)
            rb[j] = *reinterpret_cast<const f32x4*>(d.w + (wrow + b_k[j]) * d.N + b_col[j]);
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(lds_a(buf) + (prow + 32 * j) * IG_AP + cvec * 4) = ra[j];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int f = tid + 256 * j;
            *reinterpret_cast<f32x4*>(lds_b(buf) + f * 4) = rb[j];
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int a_off = (wave * 32 + l31) * IG_AP + 4 * half;
    const int b_off = (4 * half) * BN + l31;

    if (nchunks > 0) {
        load_chunk(0);
        store_chunk(0);
    }
    __syncthreads();
    for (int it = 0; it < nchunks; ++it) {
        const int buf = it & 1;
        if (it + 1 < nchunks) load_chunk(it + 1);
        const float* As = lds_a(buf) + a_off;
        const float* Bs = lds_b(buf) + b_off;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(As + 8 * q);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float b = Bs[(8 * q + s) * BN + n * 32];
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b, acc[n], 0, 0, 0);
                }
            }
        }
        if (it + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue --------------------------------------------------------------------------
    if (LSTM) {
        const int C = d.C;
        const int ch = nblk * 32 + l31;
        const float bj = d.bias[ch], bi = d.bias[C + ch], bf = d.bias[2 * C + ch] + 1.0f, bo = d.bias[3 * C + ch];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < d.M) {
                const size_t o = (size_t)m * C + ch;
                const float gj = acc[0][r] + bj;
                const float gi = acc[1 % NT][r] + bi;
                const float gf = acc[2 % NT][r] + bf;
                const float go = acc[3 % NT][r] + bo;
                const float cn = d.cstate_in[o] * sigmoidf_(gf) + sigmoidf_(gi) * tanhf(gj);
                d.cstate_out[o] = cn;
                d.hout[o] = tanhf(cn) * sigmoidf_(go);
            }
        }
    } else {
        const int oy0 = d.oy0[phase], ox0 = d.ox0[phase];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < d.M) {
                const int b = m / HWg;
                const int rem = m - b * HWg;
                const int ay = rem / d.Wg;
                const int ax = rem - ay * d.Wg;
                const size_t o = ((size_t)(b * d.Hout + ay * d.out_step + oy0) * d.Wout + ax * d.out_step + ox0) * d.ldo;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int col = nblk * BN + n * 32 + l31;
                    float v = acc[n][r] + (d.bias ? d.bias[col] : 0.f);
                    if (d.relu) v = fmaxf(v, 0.f);
                    d.out[o + col] = v;
                }
            }
        }
    }
}

template <int NT, bool LSTM>
static int launch_igemm(const IgemmDesc& d, hipStream_t stream) {
    constexpr int BN = 32 * NT;
    const int n_nblk = LSTM ? (d.C >> 5) : (d.N / BN);
    const int mblk = (d.M + IG_BM - 1) / IG_BM;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_f32_kernel<NT, LSTM>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, ig_lds_bytes<NT>());
        attr_set = true;
    }
    dim3 grid(mblk * n_nblk, d.nphase);
    hipLaunchKernelGGL((igemm_f32_kernel<NT, LSTM>), grid, dim3(256), ig_lds_bytes<NT>(), stream, d);
    return PIVP_LAUNCH_STATUS();
}

int igemm_validate(const IgemmDesc& d, bool lstm) {
    PIVP_CHECK_ARG(d.x0 && d.w && d.c0 > 0 && d.c0 % 32 == 0 && d.c1 % 32 == 0 && d.c1 >= 0);
    PIVP_CHECK_ARG(d.c1 == 0 || d.x1);
    PIVP_CHECK_ARG(d.ld0 >= d.c0 && d.ld0 % 4 == 0 && (d.c1 == 0 || (d.ld1 >= d.c1 && d.ld1 % 4 == 0)));
    PIVP_CHECK_ARG(d.B > 0 && d.Hin > 0 && d.Win > 0 && d.Hg > 0 && d.Wg > 0 && d.in_step >= 1);
    PIVP_CHECK_ARG(d.M == d.B * d.Hg * d.Wg);
    PIVP_CHECK_ARG(d.nphase >= 1 && d.nphase <= 4);
    for (int p = 0; p < d.nphase; ++p) {
        PIVP_CHECK_ARG(d.tap_start[p] >= 0 && d.tap_count[p] >= 1 && d.tap_start[p] + d.tap_count[p] <= IG_MAX_TAPS);
    }
    if (lstm) {
        PIVP_CHECK_ARG(d.C > 0 && d.C % 32 == 0 && d.N == 4 * d.C && d.bias && d.cstate_in && d.cstate_out && d.hout);
        PIVP_CHECK_ARG(d.nphase == 1 && d.in_step == 1 && d.Hg == d.Hin && d.Wg == d.Win);
    } else {
        PIVP_CHECK_ARG(d.out && d.N % 32 == 0 && d.N >= 32 && d.N <= 128 && d.ldo >= d.N);
        PIVP_CHECK_ARG(d.out_step >= 1 && d.Hout > 0 && d.Wout > 0);
        for (int p = 0; p < d.nphase; ++p) {
            PIVP_CHECK_ARG((d.Hg - 1) * d.out_step + d.oy0[p] < d.Hout && (d.Wg - 1) * d.out_step + d.ox0[p] < d.Wout);
            PIVP_CHECK_ARG(d.oy0[p] >= 0 && d.ox0[p] >= 0);
        }
    }
    return PIVP_OK;
}

int igemm_lstm(const IgemmDesc& d, hipStream_t stream) {
    int rc = igemm_validate(d, true);
    if (rc != PIVP_OK) return rc;
    return launch_igemm<4, true>(d, stream);
}

int igemm_conv(const IgemmDesc& d, hipStream_t stream) {
    int rc = igemm_validate(d, false);
    if (rc != PIVP_OK) return rc;
    switch (d.N / 32) {
        case 1: return launch_igemm<1, false>(d, stream);
        case 2: return launch_igemm<2, false>(d, stream);
        case 3: return launch_igemm<3, false>(d, stream);
        case 4: return launch_igemm<4, false>(d, stream);
    }
    return PIVP_ERR_BADARG;
}

}  // namespace pivp
