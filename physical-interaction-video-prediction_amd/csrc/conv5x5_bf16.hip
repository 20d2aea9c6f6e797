// The plain 5x5 stride-1 convolution on the bf16 / split-precision kernels: the ConvLSTM DATA gradient of the precision modes (x = dG, w = the
// flipped transposed weights).  LSTM = false instantiations of convlstm_bf16_kernel (csrc/convlstm_ring.h) and convlstm_x6g_kernel (csrc/convlstm_l2direct.h).
#undef PIVP_BF16_STAMPS      // (the phase-stamp array lives in convlstm_bf16.hip: the cell forms only)
#include "convlstm_ring.h"
#include "convlstm_l2direct.h"

namespace pivp {

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
    const int target = pivp_cu_count();
    int ks = 1;
    if (d.ksplit_ok && !d.accum)
        while (ks * 2 <= ncg && (long)tiles * nb * ks * 2 <= target) ks *= 2;
    return ks;
}

int conv5x5_bf16(const IgemmDesc& d, const unsigned short* wb, hipStream_t stream, int planes) {
    PIVP_CHECK_ARG(wb && bf16_geometry_ok(d) && d.out && d.N > 0 && d.ldo >= d.N && d.x0 && d.c0 > 0 && ((planes >= 1 && planes <= 3) || planes == -2) &&
                   (planes != -2 || (d.wscale_part && d.c1 == 0)));
    const int Np = conv5x5_bf16_rows(d.N);
    IgemmDesc dd = d;
    dd.N = Np;                                         // the kernel's weight-row count
    const bool wide = Np % 128 == 0;
    const int nb = Np / (wide ? 128 : 64);
    const int ks = conv5x5_bf16_ksplit(d, planes);
    PIVP_CHECK_ARG(!dd.ep_mode || (ks == 1 && dd.ep_src && dd.ep_ld >= dd.ep_cols && (dd.ep_mode == 1 || dd.ep_mode == 2)));      // the caller asks conv5x5_bf16_ksplit first
    if (planes == 3 && d.Win % 16)     // ... on an 8-wide map (an even batch): tiles of two images
        return launch_x6g_plain<3, true>(dd, wb, stream, Np / 64, ks, d.N);
    if (planes == 3)     // three pieces (wb packed with planes = 3, plain = 1): 64-column blocks, weights from L2 into the operand registers, eight
        return launch_x6g_plain<3>(dd, wb, stream, Np / 64, ks, d.N);      // waves (the k-step-ring form of it measured 118 us per launch in the sweep against 100)
    if (planes == -2 && d.Win % 16)     // ... on an 8-wide map (an even batch): tiles of two images
        return launch_x6g_plain<2, true>(dd, wb, stream, Np / 64, ks, d.N);
    if (planes == -2)    // two fp16 pieces (wb packed with planes = -2, plain = 1; d.wscale_part = absmax_partials(d.x0): the activations' scale)
        return launch_x6g_plain<2>(dd, wb, stream, Np / 64, ks, d.N);
    if (planes == 2)     // split mode (wb packed with planes = 2): 128-column blocks run the two-slot schedule, 64-column ones the four-slot one
        return wide ? launch_bf16<32, false, 2>(dd, wb, stream, nullptr, nb, ks, d.N)
                    : launch_bf16<16, false, 2>(dd, wb, stream, nullptr, nb, ks, d.N);
    return wide ? launch_bf16<32, false>(dd, wb, stream, nullptr, nb, ks, d.N)
                : launch_bf16<16, false>(dd, wb, stream, nullptr, nb, ks, d.N);
}

}  // namespace pivp

PIVP_DEFINE_MAIN_PRIO_SETTER(conv5x5_bf16)
