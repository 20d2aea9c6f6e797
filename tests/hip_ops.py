"""NumPy-in / NumPy-out wrappers around the per-op C-ABI entry points, for the GPU parity tests.
Inputs and outputs use the reference's NCHW layouts; the wrappers do the NHWC permutations."""
import numpy as np
import torch

import pivp_amd
from pivp_amd import _lib

DEV = 'cuda:0'


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).to(DEV)


def nhwc(a):            # (B,C,H,W) -> device NHWC
    return _t(np.asarray(a).transpose(0, 2, 3, 1))


def nchw(t, B, H, W, C):
    return t.cpu().numpy().reshape(B, H, W, C).transpose(0, 3, 1, 2)


def stream():
    return torch.cuda.current_stream().cuda_stream


def convlstm(x, h, c, W, b, variant=0, h_is_zero=False):
    lib = _lib.load()
    B, cx, H, Wd = x.shape
    C = h.shape[1]
    xd, hd, cd = nhwc(x), nhwc(h), nhwc(c)
    wd, bd = _t(pivp_amd.to_internal('lstm1/conv/W', W)), _t(b)
    c_out = torch.empty_like(cd); h_out = torch.empty_like(hd)
    _lib.check(lib.pivp_convlstm_v(xd.data_ptr(), cx, cx, None if h_is_zero else hd.data_ptr(), C, wd.data_ptr(), bd.data_ptr(), cd.data_ptr(),
                                   c_out.data_ptr(), h_out.data_ptr(), B, H, Wd, variant, stream()), 'convlstm')
    torch.cuda.synchronize()
    return nchw(h_out, B, H, Wd, C), nchw(c_out, B, H, Wd, C)


def convlstm_bf16(x, h, c, W, b, nch=0, h_is_zero=False, want_gates=False, want_ln=False):
    """bf16-operand ConvLSTM; returns (h, c[, gates NHWC-flat][, (ln partials [B][np][4], np)])."""
    import ctypes
    lib = _lib.load()
    B, cx, H, Wd = x.shape
    C = h.shape[1]
    xd, hd, cd = nhwc(x), nhwc(h), nhwc(c)
    wd, bd = _t(pivp_amd.to_internal('lstm1/conv/W', W)), _t(b)
    wb = torch.empty(lib.pivp_lstm_bf16_weight_elems(cx + C, C), dtype=torch.int16, device=DEV)
    _lib.check(lib.pivp_pack_lstm_bf16(wd.data_ptr(), wb.data_ptr(), cx + C, C, stream()), 'pack_lstm_bf16')
    c_out = torch.empty_like(cd); h_out = torch.empty_like(hd)
    gates = torch.empty((B * H * Wd, 4 * C), dtype=torch.float32, device=DEV) if want_gates else None
    cap = (H * Wd // 128 + 1) * (C // 16)
    part = torch.zeros((B, cap, 4), dtype=torch.float32, device=DEV) if want_ln else None
    npart = ctypes.c_int(-1)
    _lib.check(lib.pivp_convlstm_bf16(xd.data_ptr(), cx, cx, None if h_is_zero else hd.data_ptr(), C, wb.data_ptr(), bd.data_ptr(),
                                      cd.data_ptr(), c_out.data_ptr(), h_out.data_ptr(), gates.data_ptr() if want_gates else None,
                                      part.data_ptr() if want_ln else None, cap, ctypes.addressof(npart) if want_ln else None,
                                      B, H, Wd, nch, stream()), 'convlstm_bf16')
    torch.cuda.synchronize()
    out = [nchw(h_out, B, H, Wd, C), nchw(c_out, B, H, Wd, C)]
    if want_gates:
        out.append(gates.cpu().numpy())
    if want_ln:
        n = npart.value
        out.append((part.cpu().numpy().reshape(-1)[:B * n * 4].reshape(B, n, 4) if n > 0 else None, n))
    return tuple(out)


def convlstm_bf16x3(x, h, c, W, b, h_is_zero=False, nch=0):
    """Split-bf16 ConvLSTM (three bf16 MFMAs per product); returns (h, c)."""
    lib = _lib.load()
    B, cx, H, Wd = x.shape
    C = h.shape[1]
    xd, hd, cd = nhwc(x), nhwc(h), nhwc(c)
    wd, bd = _t(pivp_amd.to_internal('lstm1/conv/W', W)), _t(b)
    wb = torch.empty(2 * lib.pivp_lstm_bf16_weight_elems(cx + C, C), dtype=torch.int16, device=DEV)
    _lib.check(lib.pivp_pack_lstm_bf16x3(wd.data_ptr(), wb.data_ptr(), cx + C, C, stream()), 'pack_lstm_bf16x3')
    c_out = torch.empty_like(cd); h_out = torch.empty_like(hd)
    _lib.check(lib.pivp_convlstm_bf16x3(xd.data_ptr(), cx, cx, None if h_is_zero else hd.data_ptr(), C, wb.data_ptr(), bd.data_ptr(),
                                        cd.data_ptr(), c_out.data_ptr(), h_out.data_ptr(), None, None, 0, None, B, H, Wd, nch, stream()),
               'convlstm_bf16x3')
    torch.cuda.synchronize()
    return nchw(h_out, B, H, Wd, C), nchw(c_out, B, H, Wd, C)


def convlstm_bf16x6(x, h, c, W, b, h_is_zero=False, nch=0, want_ln=False):
    """Three-piece ConvLSTM (six bf16 MFMAs per product, fp32-grade); returns (h, c)."""
    lib = _lib.load()
    B, cx, H, Wd = x.shape
    C = h.shape[1]
    xd, hd, cd = nhwc(x), nhwc(h), nhwc(c)
    wd, bd = _t(pivp_amd.to_internal('lstm1/conv/W', W)), _t(b)
    wb = torch.empty(3 * lib.pivp_lstm_bf16_weight_elems(cx + C, C), dtype=torch.int16, device=DEV)
    _lib.check(lib.pivp_pack_lstm_bf16x6(wd.data_ptr(), wb.data_ptr(), cx + C, C, stream()), 'pack_lstm_bf16x6')
    c_out = torch.empty_like(cd); h_out = torch.empty_like(hd)
    import ctypes
    cap = 4096
    part = torch.zeros(B * cap * 4, dtype=torch.float32, device=DEV)
    npart = ctypes.c_int(-1)
    _lib.check(lib.pivp_convlstm_bf16x6(xd.data_ptr(), cx, cx, None if h_is_zero else hd.data_ptr(), C, wb.data_ptr(), bd.data_ptr(),
                                        cd.data_ptr(), c_out.data_ptr(), h_out.data_ptr(), None, part.data_ptr() if want_ln else None, cap,
                                        ctypes.addressof(npart) if want_ln else None, B, H, Wd, nch, stream()),
               'convlstm_bf16x6')
    torch.cuda.synchronize()
    if want_ln:
        n = npart.value
        return nchw(h_out, B, H, Wd, C), nchw(c_out, B, H, Wd, C), (part.cpu().numpy().reshape(-1)[:B * n * 4].reshape(B, n, 4) if n > 0 else None, n)
    return nchw(h_out, B, H, Wd, C), nchw(c_out, B, H, Wd, C)


def convlstm_fp16x3(x, h, c, W, b, h_is_zero=False, nch=0, want_ln=False):
    """Two-fp16-piece ConvLSTM (three fp16 MFMAs per product, weights packed times a power of two); returns (h, c)."""
    import ctypes
    lib = _lib.load()
    B, cx, H, Wd = x.shape
    C = h.shape[1]
    xd, hd, cd = nhwc(x), nhwc(h), nhwc(c)
    wd, bd = _t(pivp_amd.to_internal('lstm1/conv/W', W)), _t(b)
    wb = torch.empty(2 * lib.pivp_lstm_bf16_weight_elems(cx + C, C) + 256, dtype=torch.int16, device=DEV)     # + the scale's tail
    _lib.check(lib.pivp_pack_lstm_fp16x3(wd.data_ptr(), wb.data_ptr(), cx + C, C, Wd, stream()), 'pack_lstm_fp16x3')
    c_out = torch.empty_like(cd); h_out = torch.empty_like(hd)
    cap = 4096
    part = torch.zeros(B * cap * 4, dtype=torch.float32, device=DEV)
    npart = ctypes.c_int(-1)
    _lib.check(lib.pivp_convlstm_fp16x3(xd.data_ptr(), cx, cx, None if h_is_zero else hd.data_ptr(), C, wb.data_ptr(), bd.data_ptr(),
                                        cd.data_ptr(), c_out.data_ptr(), h_out.data_ptr(), None, part.data_ptr() if want_ln else None, cap,
                                        ctypes.addressof(npart) if want_ln else None, B, H, Wd, nch, stream()),
               'convlstm_fp16x3')
    torch.cuda.synchronize()
    if want_ln:
        n = npart.value
        return nchw(h_out, B, H, Wd, C), nchw(c_out, B, H, Wd, C), (part.cpu().numpy().reshape(-1)[:B * n * 4].reshape(B, n, 4) if n > 0 else None, n)
    return nchw(h_out, B, H, Wd, C), nchw(c_out, B, H, Wd, C)


def convlstm_ln(x, h, c, W, b, gamma, beta, eps, variant=0):
    """hidden = norm(lstm(x)) with the LayerNorm statistics from the ConvLSTM epilogue; returns (ln(h), h, c, fused)."""
    import ctypes
    lib = _lib.load()
    B, cx, H, Wd = x.shape
    C = h.shape[1]
    xd, hd, cd = nhwc(x), nhwc(h), nhwc(c)
    wd, bd = _t(pivp_amd.to_internal('lstm1/conv/W', W)), _t(b)
    perm = lambda v: _t(np.asarray(v).reshape(C, H * Wd).T)
    gd, betad = perm(gamma), perm(beta)
    c_out = torch.empty_like(cd); h_out = torch.empty_like(hd); ln_out = torch.empty_like(hd)
    scratch = torch.empty(lib.pivp_convlstm_ln_scratch_floats(B, H, Wd, C), dtype=torch.float32, device=DEV)
    fused = ctypes.c_int(-1)
    _lib.check(lib.pivp_convlstm_ln(xd.data_ptr(), cx, cx, hd.data_ptr(), C, wd.data_ptr(), bd.data_ptr(), cd.data_ptr(),
                                    c_out.data_ptr(), h_out.data_ptr(), gd.data_ptr(), betad.data_ptr(), ln_out.data_ptr(), C,
                                    scratch.data_ptr(), eps, B, H, Wd, variant, ctypes.addressof(fused), stream()), 'convlstm_ln')
    torch.cuda.synchronize()
    return nchw(ln_out, B, H, Wd, C), nchw(h_out, B, H, Wd, C), nchw(c_out, B, H, Wd, C), fused.value


def conv3x3s2(x, W, b, relu):
    lib = _lib.load()
    B, cin, H, Wd = x.shape
    cout = W.shape[0]
    xd, wd, bd = nhwc(x), _t(pivp_amd.to_internal('enc1/W', W)), _t(b)
    out = torch.empty((B, H // 2, Wd // 2, cout), dtype=torch.float32, device=DEV)
    _lib.check(lib.pivp_conv3x3s2(xd.data_ptr(), cin, cin, wd.data_ptr(), bd.data_ptr(), out.data_ptr(), cout, cout,
                                  int(relu), B, H, Wd, stream()), 'conv3x3s2')
    torch.cuda.synchronize()
    return nchw(out, B, H // 2, Wd // 2, cout)


def deconv3x3s2(x, W, b, relu, bf16=False):
    lib = _lib.load()
    B, cin, H, Wd = x.shape
    cout = W.shape[1]
    xd, wd, bd = nhwc(x), _t(pivp_amd.to_internal('enc4/W', W)), _t(b)
    out = torch.empty((B, 2 * H, 2 * Wd, cout), dtype=torch.float32, device=DEV)
    if bf16 == 'fp16x3':            # two fp16 pieces per operand (fp32-grade)
        scratch = torch.zeros(128, dtype=torch.float32, device=DEV)
        _lib.check(lib.pivp_deconv3x3s2_fp16x3(xd.data_ptr(), cin, cin, wd.data_ptr(), bd.data_ptr(), out.data_ptr(), cout, cout,
                                               int(relu), B, H, Wd, scratch.data_ptr(), stream()), 'deconv3x3s2_fp16x3')
        torch.cuda.synchronize()
        return nchw(out, B, 2 * H, 2 * Wd, cout)
    fn = lib.pivp_deconv3x3s2_bf16x3 if bf16 == 3 else lib.pivp_deconv3x3s2_bf16 if bf16 else lib.pivp_deconv3x3s2
    _lib.check(fn(xd.data_ptr(), cin, cin, wd.data_ptr(), bd.data_ptr(), out.data_ptr(), cout, cout,
                  int(relu), B, H, Wd, stream()), 'deconv3x3s2')
    torch.cuda.synchronize()
    return nchw(out, B, 2 * H, 2 * Wd, cout)


def conv_enc0(img, W, b):
    lib = _lib.load()
    B, _, H, Wd = img.shape
    out = torch.empty((B, H // 2, Wd // 2, 32), dtype=torch.float32, device=DEV)
    imgd, wd, bd = _t(img), _t(pivp_amd.to_internal('enc0/W', W)), _t(b)
    _lib.check(lib.pivp_conv_enc0(imgd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, H, Wd, stream()), 'enc0')
    torch.cuda.synchronize()
    return nchw(out, B, H // 2, Wd // 2, 32)


def layernorm(x, gamma, beta, eps, relu, name='hidden1'):
    lib = _lib.load()
    B, C, H, Wd = x.shape
    n = C * H * Wd
    xd = nhwc(x)
    perm = lambda v: _t(np.asarray(v).reshape(C, H * Wd).T)
    gd, bd = perm(gamma), perm(beta)
    out = torch.empty_like(xd)
    scratch = torch.empty(lib.pivp_layernorm_scratch_floats(B, n), dtype=torch.float32, device=DEV)
    _lib.check(lib.pivp_layernorm(xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), out.data_ptr(), scratch.data_ptr(),
                                  B, n, C, C, eps, int(relu), stream()), 'layernorm')
    torch.cuda.synchronize()
    return nchw(out, B, H, Wd, C)


def enc3_state(e2, action, state, W3, b3, Wcs, bcs, use_state=True):
    lib = _lib.load()
    B, _, H, Wd = e2.shape
    e2d = nhwc(e2)
    out = torch.empty_like(e2d)
    st = torch.empty((B, 5), dtype=torch.float32, device=DEV)
    args = [_t(action), _t(state), _t(pivp_amd.to_internal('enc3/W', W3)), _t(b3), _t(Wcs), _t(bcs)]
    _lib.check(lib.pivp_enc3_state(e2d.data_ptr(), *[a.data_ptr() for a in args], out.data_ptr(), st.data_ptr(),
                                   B, H * Wd, int(use_state), stream()), 'enc3_state')
    torch.cuda.synchronize()
    return nchw(out, B, H, Wd, 64), st.cpu().numpy()


def heads(e6, Wm, bm, We, be, num_masks, model_type):
    lib = _lib.load()
    B, _, H, Wd = e6.shape
    NP, NE = num_masks + 1, We.shape[1]
    e6d = nhwc(e6)
    logits = torch.empty((B, NP, H, Wd), dtype=torch.float32, device=DEV)
    enc7 = torch.empty((B, NE, H, Wd), dtype=torch.float32, device=DEV)
    layer0 = torch.empty((B, 3, H, Wd), dtype=torch.float32, device=DEV)
    args = [_t(pivp_amd.to_internal('masks/W', Wm)), _t(bm), _t(pivp_amd.to_internal('model/enc7/W', We)), _t(be)]
    _lib.check(lib.pivp_heads(e6d.data_ptr(), *[a.data_ptr() for a in args], logits.data_ptr(), enc7.data_ptr(),
                              layer0.data_ptr(), B, H * Wd, num_masks, model_type, stream()), 'heads')
    torch.cuda.synchronize()
    return logits.cpu().numpy(), enc7.cpu().numpy(), layer0.cpu().numpy()


def cdna_kernels(hidden5, W, b, num_masks):
    lib = _lib.load()
    B, C, H, Wd = hidden5.shape
    K = C * H * Wd
    hd = nhwc(hidden5)
    wd, bd = _t(pivp_amd.to_internal('model/cdna_kerns/W', W)), _t(b)
    scratch = torch.empty(lib.pivp_linear_scratch_floats(B, K), dtype=torch.float32, device=DEV)
    kern = torch.empty((B, num_masks, 5, 5), dtype=torch.float32, device=DEV)
    _lib.check(lib.pivp_cdna_kernels(hd.data_ptr(), wd.data_ptr(), bd.data_ptr(), scratch.data_ptr(), kern.data_ptr(),
                                     B, K, num_masks, stream()), 'cdna_kernels')
    torch.cuda.synchronize()
    return kern.cpu().numpy()


def stp_params(hidden5, W1, b1, W2, b2):
    lib = _lib.load()
    B, C, H, Wd = hidden5.shape
    K = C * H * Wd
    hd = nhwc(hidden5)
    args = [_t(pivp_amd.to_internal('model/stp_input/W', W1)), _t(b1), _t(W2), _t(b2)]
    scratch = torch.empty(lib.pivp_linear_scratch_floats(B, K), dtype=torch.float32, device=DEV)
    theta = torch.empty((B, 6), dtype=torch.float32, device=DEV)
    _lib.check(lib.pivp_stp_params(hd.data_ptr(), *[a.data_ptr() for a in args], scratch.data_ptr(), theta.data_ptr(),
                                   B, K, stream()), 'stp_params')
    torch.cuda.synchronize()
    return theta.cpu().numpy()


def composite(prev, logits, layer0, aux, num_masks, model_type, stp_zero=0):
    lib = _lib.load()
    B, _, H, Wd = prev.shape
    pd, ld, ad = _t(prev), _t(logits), _t(aux)
    l0 = _t(layer0) if layer0 is not None else None
    out = torch.empty((B, 3, H, Wd), dtype=torch.float32, device=DEV)
    masks = torch.empty((B, num_masks + 1, H, Wd), dtype=torch.float32, device=DEV)
    _lib.check(lib.pivp_composite(pd.data_ptr(), ld.data_ptr(), l0.data_ptr() if l0 is not None else None, ad.data_ptr(),
                                  out.data_ptr(), masks.data_ptr(), B, H, Wd, num_masks, model_type, stp_zero, stream()),
               'composite')
    torch.cuda.synchronize()
    return out.cpu().numpy(), masks.cpu().numpy()


def frame_head_pair(e6raw, gamma, beta, eps, Wm, bm, We, be, prev, hidden5, Wh, bh, W2, b2, num_masks, model_type, stp_zero=0, finisher=True,
                    train=False):
    """The output side of a timestep twice from the same device buffers: (1) pivp_layernorm(+relu) -> pivp_heads -> pivp_cdna_kernels /
    pivp_stp_params -> pivp_composite, (2) pivp_motion_partials -> pivp_frame_head (finisher=False: the separate finisher's kernels / theta
    through `aux`).  Returns two dicts of numpy arrays (out, masks, enc7, and for train=True logits, layer0, enc6, kerns, vpre)."""
    import ctypes
    lib = _lib.load()
    B, _, H, Wd = prev.shape
    HW = H * Wd
    NP, NE = num_masks + 1, We.shape[1]
    n = 64 * HW
    e6d = nhwc(e6raw)
    perm = lambda v: _t(np.asarray(v).reshape(64, HW).T)
    gd, bd = perm(gamma), perm(beta)
    wmd, bmd = _t(pivp_amd.to_internal('masks/W', Wm)), _t(bm)
    wed, bed = _t(pivp_amd.to_internal('model/enc7/W', We)), _t(be)
    pd = _t(prev)
    new = lambda *shape: torch.full(shape, 7.0, dtype=torch.float32, device=DEV)
    # ---- (1) the separate kernels -------------------------------------------------------------------------------------------------
    e6n = torch.empty_like(e6d)
    lnscr = torch.empty(lib.pivp_layernorm_scratch_floats(B, n), dtype=torch.float32, device=DEV)
    _lib.check(lib.pivp_layernorm(e6d.data_ptr(), gd.data_ptr(), bd.data_ptr(), e6n.data_ptr(), lnscr.data_ptr(), B, n, 64, 64, eps, 1, stream()), 'layernorm')
    nparts = lib.pivp_layernorm_scratch_floats(1, n) // 4          # (count, mean, M2, -) per ln_stats slice, left in the scratch
    logits, enc7, layer0 = new(B, NP, H, Wd), new(B, NE, H, Wd), new(B, 3, H, Wd)
    _lib.check(lib.pivp_heads(e6n.data_ptr(), wmd.data_ptr(), bmd.data_ptr(), wed.data_ptr(), bed.data_ptr(), logits.data_ptr(), enc7.data_ptr(),
                              layer0.data_ptr(), B, HW, num_masks, model_type, stream()), 'heads')
    aux = None
    if model_type != 2:
        K = int(np.prod(hidden5.shape[1:]))
        hd = nhwc(hidden5)
        lin = torch.empty(lib.pivp_linear_scratch_floats(B, K), dtype=torch.float32, device=DEV)
        if model_type == 0:
            whd, bhd = _t(pivp_amd.to_internal('model/cdna_kerns/W', Wh)), _t(bh)
            aux = new(B, num_masks, 5, 5)
            _lib.check(lib.pivp_cdna_kernels(hd.data_ptr(), whd.data_ptr(), bhd.data_ptr(), lin.data_ptr(), aux.data_ptr(), B, K, num_masks, stream()), 'cdna_kernels')
        else:
            whd, bhd, w2d, b2d = _t(pivp_amd.to_internal('model/stp_input/W', Wh)), _t(bh), _t(W2), _t(b2)
            aux = new(B, 6)
            _lib.check(lib.pivp_stp_params(hd.data_ptr(), whd.data_ptr(), bhd.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), lin.data_ptr(), aux.data_ptr(), B, K, stream()), 'stp_params')
    out, masks = new(B, 3, H, Wd), new(B, NP, H, Wd)
    _lib.check(lib.pivp_composite(pd.data_ptr(), logits.data_ptr(), layer0.data_ptr() if model_type != 2 else None, (aux if aux is not None else enc7).data_ptr(),
                                  out.data_ptr(), masks.data_ptr(), B, H, Wd, num_masks, model_type, stp_zero, stream()), 'composite')
    torch.cuda.synchronize()
    sep = dict(out=out.cpu().numpy(), masks=masks.cpu().numpy(), enc7=enc7.cpu().numpy(), logits=logits.cpu().numpy(), layer0=layer0.cpu().numpy(),
               enc6=nchw(e6n, B, H, Wd, 64), kerns=None if aux is None else aux.cpu().numpy())
    # ---- (2) the fused launch ------------------------------------------------------------------------------------------------------------
    fits = lib.pivp_frame_head_fits(model_type, B, H, Wd, num_masks, 0 if model_type == 2 else int(np.prod(hidden5.shape[1:])))
    assert fits in (1, 2), 'pivp_frame_head_fits says no'
    a = _lib.PivpFrameHeadArgs()
    a.e6raw = e6d.data_ptr(); a.ln_part = lnscr.data_ptr(); a.ln_nparts = nparts; a.gamma = gd.data_ptr(); a.beta = bd.data_ptr(); a.ln_eps = eps
    a.masks_w = wmd.data_ptr(); a.masks_b = bmd.data_ptr(); a.enc7_w = wed.data_ptr(); a.enc7_b = bed.data_ptr(); a.prev = pd.data_ptr()
    out2, masks2, enc72 = new(B, 3, H, Wd), new(B, NP, H, Wd), new(B, NE, H, Wd)
    logits2, layer02, e6n2, stat2 = new(B, NP, H, Wd), new(B, 3, H, Wd), torch.full_like(e6d, 7.0), new(B, 2)
    kerns2 = new(B, num_masks, 5, 5) if model_type == 0 else new(B, 6)
    vpre2 = new(B, 256)
    a.out = out2.data_ptr(); a.masks_out = masks2.data_ptr(); a.enc7 = enc72.data_ptr()
    if train:
        a.logits_out = logits2.data_ptr(); a.layer0_out = layer02.data_ptr(); a.enc6_out = e6n2.data_ptr(); a.stat_out = stat2.data_ptr()
    if model_type != 2:
        if finisher and fits == 1:
            lin.fill_(float('nan'))                                   # the tail padding is read but never summed: NaNs there must not matter
            _lib.check(lib.pivp_motion_partials(hd.data_ptr(), whd.data_ptr(), lin.data_ptr(), B, K, 1 if model_type == 1 else 0, stream()), 'motion_partials')
            a.partials = lin.data_ptr(); a.kslices = (K + 63) // 64; a.head_bias = bhd.data_ptr()
            a.kerns_out = kerns2.data_ptr(); a.vpre_out = vpre2.data_ptr()
            if model_type == 1:
                a.w2 = w2d.data_ptr(); a.b2 = b2d.data_ptr()
        else:
            a.aux = aux.data_ptr()
    a.B, a.H, a.W, a.num_masks, a.model_type, a.stp_zero_border = B, H, Wd, num_masks, model_type, stp_zero
    _lib.check(lib.pivp_frame_head(ctypes.byref(a), stream()), 'frame_head')
    torch.cuda.synchronize()
    fused = dict(out=out2.cpu().numpy(), masks=masks2.cpu().numpy(), enc7=enc72.cpu().numpy(), logits=logits2.cpu().numpy(), layer0=layer02.cpu().numpy(),
                 enc6=nchw(e6n2, B, H, Wd, 64), kerns=kerns2.cpu().numpy() if (model_type != 2 and finisher and fits == 1) else None,
                 stat=stat2.cpu().numpy(), vpre=vpre2.cpu().numpy())
    return sep, fused


def select_frames(gt, gen, take):
    lib = _lib.load()
    B = gt.shape[0]
    g, p = _t(gt), _t(gen)
    tk = torch.from_numpy(np.asarray(take, dtype=np.uint8)).to(DEV)
    out = torch.empty_like(g)
    _lib.check(lib.pivp_select_frames(g.data_ptr(), p.data_ptr(), tk.data_ptr(), out.data_ptr(), B, g[0].numel(), stream()),
               'select_frames')
    torch.cuda.synchronize()
    return out.cpu().numpy()


def conv5x5_bf16(x, W, accum_into=None, split=False, pieces=0):
    """Plain 5x5 stride-1 'same' convolution with bf16 operands: x NCHW, W (Cout, Cin, 5, 5) -> NCHW."""
    lib = _lib.load()
    B, cin, H, Wd = x.shape
    cout = W.shape[0]
    xd = nhwc(x)
    wd = _t(pivp_amd.to_internal('lstm1/conv/W', W))          # [25][cin/32][cout][32]
    wb = torch.empty((3 if pieces == 3 else 2 if (split or pieces == 'fp16x3') else 1) * lib.pivp_conv5x5_bf16_weight_elems(cin, cout) + 256, dtype=torch.int16, device=DEV)
    out = nhwc(accum_into) if accum_into is not None else torch.full((B, H, Wd, cout), 7.0, dtype=torch.float32, device=DEV)
    if pieces == 'fp16x3':       # two fp16 pieces per operand, the activations' scale from their largest |value|
        scratch = torch.zeros(128, dtype=torch.float32, device=DEV)
        _lib.check(lib.pivp_conv5x5_fp16x3(xd.data_ptr(), cin, cin, wd.data_ptr(), wb.data_ptr(), out.data_ptr(), cout, cout,
                                           1 if accum_into is not None else 0, B, H, Wd, scratch.data_ptr(), stream()), 'conv5x5_fp16x3')
        torch.cuda.synchronize()
        return nchw(out, B, H, Wd, cout)
    _lib.check((lib.pivp_conv5x5_bf16x6 if pieces == 3 else lib.pivp_conv5x5_bf16x3 if split else lib.pivp_conv5x5_bf16)(xd.data_ptr(), cin, cin, wd.data_ptr(), wb.data_ptr(), out.data_ptr(), cout, cout,
                                     1 if accum_into is not None else 0, B, H, Wd, stream()), 'conv5x5_bf16')
    torch.cuda.synchronize()
    return nchw(out, B, H, Wd, cout)


def wgrad5x5_bf16(x, h, dG, h_is_zero=False):
    """ConvLSTM weight gradient with bf16 operands: x (B,cx,H,W), h (B,C,H,W), dG (B,4C,H,W) -> dW in reference layout (4C, cx+C, 5, 5)."""
    lib = _lib.load()
    B, cx, H, Wd = x.shape
    C = h.shape[1]
    xd, hd, gd = nhwc(x), nhwc(h), nhwc(dG)
    dW = torch.zeros(25 * (cx + C) * 4 * C, dtype=torch.float32, device=DEV)
    db = torch.zeros(4 * C, dtype=torch.float32, device=DEV)
    _lib.check(lib.pivp_wgrad5x5_bf16(xd.data_ptr(), cx, cx, None if h_is_zero else hd.data_ptr(), C, gd.data_ptr(), dW.data_ptr(),
                                      db.data_ptr(), B, H, Wd, stream()), 'wgrad5x5_bf16')
    torch.cuda.synchronize()
    return pivp_amd.from_internal('lstm1/conv/W', dW.cpu().numpy(), (4 * C, cx + C, 5, 5)), db.cpu().numpy()


def wgrad5x5_bf16_batch(xs, hs, dGs, fp16x3=False, h_is_zero=False, bf16x6=False, form=0):
    """The batched form: lists of per-timestep x (B,cx,H,W), h (B,C,H,W), dG (B,4C,H,W); operands are laid out LAST timestep first with
    negative strides for x / h (as the backward sweep's slabs are) and positive for dG (as its ring is)."""
    lib = _lib.load()
    T = len(xs)
    B, cx, H, Wd = xs[0].shape
    C = hs[0].shape[1]
    xd = torch.stack([nhwc(x) for x in xs]); hd = torch.stack([nhwc(h) for h in hs])       # [t] ascending in memory
    gd = torch.stack([nhwc(g) for g in dGs[::-1]])                                          # ring slot j = timestep T-1-j
    dW = torch.zeros(25 * (cx + C) * 4 * C, dtype=torch.float32, device=DEV)
    db = torch.zeros(4 * C, dtype=torch.float32, device=DEV)
    sx, sh, sg = xd[0].numel() * 4, hd[0].numel() * 4, gd[0].numel() * 4
    if bf16x6:       # three bf16 pieces per operand, six MFMAs per product
        _lib.check(lib.pivp_wgrad5x5_bf16x6_batch(xd[T - 1].data_ptr(), cx, cx, None if h_is_zero else hd[T - 1].data_ptr(), C, gd.data_ptr(), dW.data_ptr(),
                                                  db.data_ptr(), B, H, Wd, T, -sx, -sh, sg, stream()), 'wgrad5x5_bf16x6_batch')
        torch.cuda.synchronize()
        return pivp_amd.from_internal('lstm1/conv/W', dW.cpu().numpy(), (4 * C, cx + C, 5, 5)), db.cpu().numpy()
    if fp16x3:       # two fp16 pieces per operand, dG scaled by a power of two from the batch's largest value
        scratch = torch.zeros(72 * T, dtype=torch.float32, device=DEV)
        _lib.check(lib.pivp_wgrad5x5_fp16x3_batch(xd[T - 1].data_ptr(), cx, cx, None if h_is_zero else hd[T - 1].data_ptr(), C, gd.data_ptr(), dW.data_ptr(),
                                                  db.data_ptr(), B, H, Wd, T, -sx, -sh, sg, scratch.data_ptr(), stream()), 'wgrad5x5_fp16x3_batch')
        torch.cuda.synchronize()
        return pivp_amd.from_internal('lstm1/conv/W', dW.cpu().numpy(), (4 * C, cx + C, 5, 5)), db.cpu().numpy()
    if form:         # 1: four-wave blocks, 2: eight-wave blocks (pivp_wgrad5x5_bf16_batch picks by size)
        _lib.check(lib.pivp_wgrad5x5_bf16_batch_form(xd[T - 1].data_ptr(), cx, cx, hd[T - 1].data_ptr(), C, gd.data_ptr(), dW.data_ptr(), db.data_ptr(),
                                                     B, H, Wd, T, -sx, -sh, sg, form, stream()), 'wgrad5x5_bf16_batch_form')
    else:
        _lib.check(lib.pivp_wgrad5x5_bf16_batch(xd[T - 1].data_ptr(), cx, cx, hd[T - 1].data_ptr(), C, gd.data_ptr(), dW.data_ptr(), db.data_ptr(),
                                                B, H, Wd, T, -sx, -sh, sg, stream()), 'wgrad5x5_bf16_batch')
    torch.cuda.synchronize()
    return pivp_amd.from_internal('lstm1/conv/W', dW.cpu().numpy(), (4 * C, cx + C, 5, 5)), db.cpu().numpy()


def deconv3x3s2_of_norm_concat(h, gamma, beta, x1, W, b, relu, eps=1e-6, precision=0, fused=True):
    """deconv3x3s2(concat(LayerNorm(h), x1)) in NCHW in / out.  fused: pivp_deconv3x3s2_ln (the norm applied while the conv stages its
    input); otherwise pivp_layernorm into the concat buffer + pivp_deconv3x3s2* on it (what a training plan runs)."""
    lib = _lib.load()
    B, c_ln, H, Wd = h.shape
    c1 = 0 if x1 is None else x1.shape[1]
    cin, cout = c_ln + c1, W.shape[1]
    n = c_ln * H * Wd
    hd = nhwc(h)
    perm = lambda v: _t(np.asarray(v).reshape(c_ln, H * Wd).T)
    gd, bed = perm(gamma), perm(beta)
    wd, bd = _t(pivp_amd.to_internal('enc4/W', W)), _t(b)
    cat = torch.zeros((B, H, Wd, cin), dtype=torch.float32, device=DEV)
    if c1:
        cat[..., c_ln:] = nhwc(x1)
    out = torch.empty((B, 2 * H, 2 * Wd, cout), dtype=torch.float32, device=DEV)
    scratch = torch.empty(lib.pivp_layernorm_scratch_floats(B, n), dtype=torch.float32, device=DEV)
    if fused:
        x1p = cat.data_ptr() + c_ln * 4 if c1 else None
        _lib.check(lib.pivp_deconv3x3s2_ln(hd.data_ptr(), c_ln, x1p, c1, cin, wd.data_ptr(), bd.data_ptr(), gd.data_ptr(), bed.data_ptr(), eps,
                                           scratch.data_ptr(), out.data_ptr(), cout, cout, int(relu), B, H, Wd, precision, stream()), 'deconv3x3s2_ln')
    else:
        _lib.check(lib.pivp_layernorm(hd.data_ptr(), gd.data_ptr(), bed.data_ptr(), cat.data_ptr(), scratch.data_ptr(),
                                      B, n, c_ln, cin, eps, 0, stream()), 'layernorm')
        fn = [lib.pivp_deconv3x3s2, lib.pivp_deconv3x3s2_bf16, lib.pivp_deconv3x3s2_bf16x3][precision]
        _lib.check(fn(cat.data_ptr(), cin, cin, wd.data_ptr(), bd.data_ptr(), out.data_ptr(), cout, cout, int(relu), B, H, Wd, stream()), 'deconv3x3s2')
    torch.cuda.synchronize()
    return nchw(out, B, 2 * H, 2 * Wd, cout)


def wgrad5x5_f32_batch(launches, slots=True, h_is_zero=False, form=0):
    """The fp32 ConvLSTM weight gradient on its own.  launches: a list of (xs, hs, dGs) batches (lists of per-timestep x (B,cx,H,W), h (B,C,H,W),
    dG (B,4C,H,W)), one launch each, laid out as the sweep's slabs and rings are (see wgrad5x5_bf16_batch).  slots: the round-6 kernel (the first launch
    stores into the NaN-initialised partial slots, the others add; one reduction at the end) -- else the round-2 kernel (atomics).  -> (dW, db, raw dW)."""
    lib = _lib.load()
    B, cx, H, Wd = launches[0][0][0].shape
    C = launches[0][1][0].shape[1]
    dW = torch.zeros(25 * (cx + C) * 4 * C, dtype=torch.float32, device=DEV)
    db = torch.zeros(4 * C, dtype=torch.float32, device=DEV)
    part = None
    if slots:
        n = lib.pivp_wgrad5x5_f32_part_floats(cx, C, B, H, Wd, form)
        assert n > 0
        part = torch.full((n,), float('nan'), dtype=torch.float32, device=DEV)
    keep = []
    for li, (xs, hs, dGs) in enumerate(launches):
        T = len(xs)
        xd = torch.stack([nhwc(x) for x in xs]); hd = torch.stack([nhwc(h) for h in hs])
        gd = torch.stack([nhwc(g) for g in dGs[::-1]])
        keep.append((xd, hd, gd))
        sx, sh, sg = xd[0].numel() * 4, hd[0].numel() * 4, gd[0].numel() * 4
        _lib.check(lib.pivp_wgrad5x5_f32_batch(xd[T - 1].data_ptr(), cx, cx, None if h_is_zero else hd[T - 1].data_ptr(), C, gd.data_ptr(),
                                               part.data_ptr() if slots else None, 1 if li == 0 else 0, dW.data_ptr(), db.data_ptr(), B, H, Wd, T,
                                               -sx, -sh, sg, form, stream()), 'wgrad5x5_f32_batch')
    if slots:
        _lib.check(lib.pivp_wgrad5x5_f32_reduce(cx, C, 0 if h_is_zero else 1, part.data_ptr(), dW.data_ptr(), db.data_ptr(), B, H, Wd, form, stream()),
                   'wgrad5x5_f32_reduce')
    torch.cuda.synchronize()
    raw = dW.cpu().numpy().copy()
    return pivp_amd.from_internal('lstm1/conv/W', dW.cpu().numpy(), (4 * C, cx + C, 5, 5)), db.cpu().numpy(), raw
