"""Known-answer tests that pin the CPU oracle to the reference's code.

Every expectation here is derived analytically from src/models/train_model.py (TM) of the
reference, not from running the oracle."""
import math

import numpy as np
import pytest

from oracle import restatement as R


def test_param_count_and_key_layout():
    # (10) 9,212,159 parameters; Chainer save_npz path keys
    shapes = R.param_shapes(num_masks=10, model_type='CDNA')
    assert sum(int(np.prod(s)) for s in shapes.values()) == 9212159
    assert shapes['lstm5/conv/W'] == (512, 192, 5, 5)
    assert shapes['enc4/W'] == (128, 128, 3, 3)
    assert shapes['masks/W'] == (64, 11, 1, 1)            # deconv layout (Cin, Cout, 1, 1)
    assert shapes['model/cdna_kerns/W'] == (250, 8192)
    assert shapes['norm_enc6/norm/gamma'] == (262144,)
    assert shapes['hidden5/norm/beta'] == (8192,)
    assert shapes['enc3/W'] == (64, 74, 1, 1)
    assert R.param_shapes(model_type='STP')['model/identity_params/W'] == (6, 100)
    assert R.param_shapes(num_masks=1, model_type='DNA')['model/enc7/W'] == (64, 25, 1, 1)
    with pytest.raises(ValueError):
        R.Model(10, is_cdna=False, is_dna=False, is_stp=False)   # TM:540


def test_conv_is_cross_correlation_with_zero_pad():
    x = np.zeros((1, 1, 5, 5)); x[0, 0, 2, 2] = 1.0
    W = np.arange(9, dtype=np.float64).reshape(1, 1, 3, 3)
    y = R.conv2d(x, W, None, 1, 1)
    # cross-correlation of a delta gives the FLIPPED kernel around the delta
    assert np.array_equal(y[0, 0, 1:4, 1:4], W[0, 0, ::-1, ::-1])
    assert R.conv2d(np.ones((1, 3, 64, 64)), np.ones((32, 3, 5, 5)), None, 2, 2).shape == (1, 32, 32, 32)


def test_deconv_sizes_and_adjointness():
    # (7) 8 -> 16 -> 32 -> 64 with outsize = 2*in (TM:505-507)
    rs = np.random.RandomState(0)
    for n in (8, 16, 32):
        x = rs.randn(1, 4, n, n)
        W = rs.randn(4, 6, 3, 3)
        y = R.deconv2d(x, W, None, 2, 1, (2 * n, 2 * n))
        assert y.shape == (1, 6, 2 * n, 2 * n)
    # deconvolution_2d is the adjoint of convolution_2d with the same W viewed as (Cout=Cin_d, Cin=Cout_d)
    x = rs.randn(2, 4, 8, 8); W = rs.randn(4, 6, 3, 3); z = rs.randn(2, 6, 16, 16)
    lhs = (R.deconv2d(x, W, None, 2, 1, (16, 16)) * z).sum()
    rhs = (x * R.conv2d(z, W, None, 2, 1)).sum()
    assert abs(lhs - rhs) < 1e-9 * max(1.0, abs(lhs))


def test_layernorm_flat_chw():
    # (6) gamma=1, beta=0: per-sample mean 0 and variance s2/(s2+eps) over flattened C*H*W
    rs = np.random.RandomState(1)
    x = rs.randn(3, 4, 5, 6) * 3 + 2
    n = 4 * 5 * 6
    y = R.layer_norm_conv2d(x, np.ones(n), np.zeros(n))
    flat = y.reshape(3, -1)
    assert np.allclose(flat.mean(1), 0, atol=1e-12)
    s2 = x.reshape(3, -1).var(1)
    assert np.allclose(flat.var(1), s2 / (s2 + R.LN_EPS), rtol=1e-10)
    # gamma/beta are indexed by flat NCHW position c*H*W + y*W + x
    g = np.arange(n, dtype=np.float64); b = -np.arange(n, dtype=np.float64)
    y2 = R.layer_norm_conv2d(x, g, b)
    assert np.allclose(y2[1, 2, 3, 4], y[1, 2, 3, 4] * g[2 * 30 + 3 * 6 + 4] + b[2 * 30 + 3 * 6 + 4])


def _model(model_type='CDNA', num_masks=10, scale=0.0, **kw):
    P = R.init_params(seed=1, dtype=np.float64, scale=scale, num_masks=num_masks, model_type=model_type)
    m = R.Model(num_masks, is_cdna=model_type == 'CDNA', is_stp=model_type == 'STP',
                is_dna=model_type == 'DNA', params=P, dtype=np.float64, prefix='kat', **kw)
    return m, P


def test_convlstm_zero_weights():
    # (5) zero weights/bias: c_t = c_{t-1} * sigmoid(1), h = tanh(c) * 1/2; zero state stays zero
    m, P = _model()
    P['lstm1/conv/W'][...] = 0; P['lstm1/conv/b'][...] = 0
    m.load_params(P)
    x = np.random.RandomState(0).randn(2, 32, 8, 8)
    h = m._lstm('lstm1', x)
    assert np.all(h == 0) and np.all(m.lstm_c['lstm1'] == 0)
    m.lstm_c['lstm1'] = np.full((2, 32, 8, 8), 0.7)
    h = m._lstm('lstm1', x)
    c_expect = 0.7 * (1 / (1 + math.exp(-1.0)))
    assert np.allclose(m.lstm_c['lstm1'], c_expect)
    assert np.allclose(h, math.tanh(c_expect) * 0.5)


def test_convlstm_gate_order_j_i_f_o():
    # quirk 7 / TM:269: out-channel blocks are j, i, f, o
    m, P = _model()
    C = 32
    W = np.zeros_like(P['lstm1/conv/W']); b = np.zeros_like(P['lstm1/conv/b'])
    b[0 * C:1 * C] = 0.3     # j -> tanh(0.3)
    b[1 * C:2 * C] = -0.2    # i -> sigmoid(-0.2)
    b[2 * C:3 * C] = 0.5     # f -> sigmoid(0.5 + 1)
    b[3 * C:4 * C] = 1.1     # o -> sigmoid(1.1)
    P['lstm1/conv/W'], P['lstm1/conv/b'] = W, b
    m.load_params(P)
    m.lstm_c['lstm1'] = np.full((1, C, 4, 4), 2.0); m.lstm_h['lstm1'] = np.zeros((1, C, 4, 4))
    h = m._lstm('lstm1', np.zeros((1, 32, 4, 4)))
    sg = lambda v: 1 / (1 + math.exp(-v))
    c = 2.0 * sg(1.5) + sg(-0.2) * math.tanh(0.3)
    assert np.allclose(m.lstm_c['lstm1'], c) and np.allclose(h, math.tanh(c) * sg(1.1))


def test_cdna_kernels_normalised_and_orientation():
    m, P = _model()
    rs = np.random.RandomState(3)
    B = 2
    prev = rs.rand(B, 3, 64, 64)
    enc6 = rs.randn(B, 64, 64, 64)
    # (1) all-equal logits -> uniform 1/25; every (b, m) kernel sums to 1
    P['model/cdna_kerns/W'][...] = 0; P['model/cdna_kerns/b'][...] = 0.37
    m.load_params(P)
    layers, enc7 = m._cdna(enc6, rs.randn(B, 128, 8, 8), prev)
    assert len(layers) == 11                                  # quirk 2: sigmoid(enc7) + 10 transformed
    assert np.allclose(m.last_cdna_kerns, 1 / 25.0)
    P['model/cdna_kerns/W'] = rs.randn(250, 8192) * 0.01
    m.load_params(P)
    m._cdna(enc6, rs.randn(B, 128, 8, 8), prev)
    assert np.allclose(m.last_cdna_kerns.sum(axis=(2, 3)), 1.0)
    assert np.all(m.last_cdna_kerns > 0)                      # relu(k - 1e-12) + 1e-12 keeps every tap positive
    # (2) delta kernel at (i, j) shifts by (i-2, j-2) with zero fill: cross-correlation orientation
    P['model/cdna_kerns/W'][...] = 0
    bias = np.full(250, -1.0)                                 # relu(-1 - 1e-12) + 1e-12 ~ 0
    bias[0 * 25 + 2 * 5 + 2] = 1.0                            # mask 0: centre
    bias[1 * 25 + 0 * 5 + 4] = 1.0                            # mask 1: (i, j) = (0, 4)
    P['model/cdna_kerns/b'] = bias
    m.load_params(P)
    layers, _ = m._cdna(enc6, np.zeros((B, 128, 8, 8)), prev)
    assert np.allclose(layers[1], prev, atol=1e-10)
    shifted = np.zeros_like(prev)
    shifted[:, :, 2:, :-2] = prev[:, :, :-2, 2:]              # out[y,x] = prev[y+0-2, x+4-2]
    assert np.allclose(layers[2], shifted, atol=1e-10)
    # generated-pixel layer is sigmoid(relu(enc7)) >= 0.5 (quirk 4)
    assert layers[0].min() >= 0.5


def test_flat11_softmax_quirk():
    # (3) softmax over 11 consecutive elements of the NCHW-flat buffer (TM:720-722)
    m, P = _model()
    B, H, W = 1, 64, 64
    P['masks/W'][...] = 0; P['masks/b'][...] = 0.25
    # make enc6 irrelevant, masks constant -> every element 1/11
    m.load_params(P)
    prev = np.random.RandomState(0).rand(B, 3, H, W)
    sa = np.zeros((B, 10))
    m.reset_state()
    taps = {}
    m._step(prev, sa, taps)
    assert np.allclose(taps['masks'], 1 / 11.0)
    # ramp: direct check of the grouping on a hand-built tensor
    r = np.arange(11 * H * W, dtype=np.float64).reshape(1, 11, H, W) * 1e-3
    sm = R.softmax_axis1(r.reshape(-1, 11)).reshape(1, 11, H, W)
    f = 5 * H * W + 17 * W + 3                                 # element (m=5, y=17, x=3)
    g0 = (f // 11) * 11
    grp = r.ravel()[g0:g0 + 11]
    expect = math.exp(r.ravel()[f] - grp.max()) / np.exp(grp - grp.max()).sum()
    assert abs(sm[0, 5, 17, 3] - expect) < 1e-15
    # groups never straddle samples: 11*H*W is divisible by 11
    assert (11 * H * W) % 11 == 0


def test_tenth_cdna_layer_has_no_influence():
    # (4) TM:726 zip: the 10th transformed layer is dropped
    m, P = _model(scale=1.0)
    imgs, acts, stas = R.synthetic_batch(1, 3)
    m.train = False
    m([imgs, acts, stas], 0)
    base = np.stack(m.gen_images)
    # perturb only the 10th kernel's logits (rows 225..249 of cdna_kerns)
    P2 = dict(P)
    P2['model/cdna_kerns/b'] = P['model/cdna_kerns/b'].copy()
    P2['model/cdna_kerns/b'][225:250] += np.linspace(-3, 3, 25)
    m2 = R.Model(10, params=P2, dtype=np.float64, prefix='kat'); m2.train = False
    m2([imgs, acts, stas], 0)
    assert np.array_equal(base, np.stack(m2.gen_images))


def test_loss_psnr_and_divisor():
    # (8) T=10, ctx=2: 8 frame terms + 8 state terms, divided by 8 (TM:739-758); psnr_all is a SUM
    m, _ = _model(scale=1.0)
    imgs, acts, stas = R.synthetic_batch(1, 10)
    m.train = False
    loss = m([imgs, acts, stas], 0)
    assert len(m.gen_images) == 9 and len(m.gen_states) == 9
    fr = [np.mean((imgs[t + 2].astype(np.float64) - m.gen_images[t + 1]) ** 2) for t in range(8)]
    st = [np.mean((stas[t + 2].astype(np.float64) - m.gen_states[t + 1]) ** 2) * 1e-4 for t in range(8)]
    assert abs(loss - (sum(fr) + sum(st)) / 8.0) < 1e-12
    assert abs(m.psnr_all - sum(10 * math.log10(1 / f) for f in fr)) < 1e-9
    assert len(m.summaries) == 8 * 2 + 8 + 2
    assert m.summaries[0].startswith('kat_recon_cost0: ')
    assert len(m.conv_res) == 8                                 # enc0..enc6 + enc7 (TM:715, TM:734)


def test_scheduled_sampling_schedule_and_select():
    # (9) k=900, iter=0, B=32 -> 32 ground-truth frames (TM:654-656)
    assert R.num_ground_truth_schedule(32, 900.0, 0) == 32
    assert R.num_ground_truth_schedule(32, 900.0, 1e9) == 0
    k = 900.0
    assert R.num_ground_truth_schedule(32, k, 6000) == int(np.round(32 * (k / (k + np.exp(6000 / k)))))
    gt = np.zeros((8, 3, 4, 4), np.float32); gen = np.ones((8, 3, 4, 4), np.float32)
    rs = np.random.RandomState(5)
    out = R.scheduled_sample(gt, gen, 8, 3, rng=rs)
    rs2 = np.random.RandomState(5); idx = np.arange(8); rs2.shuffle(idx)
    picked = np.zeros(8, bool); picked[idx[:3]] = True
    assert np.array_equal(out[:, 0, 0, 0] == 0, picked)


def test_feedself_uses_generated_frames_after_context():
    # TM:663-673: frames 0..ctx-1 are ground truth, afterwards the model's own prediction
    m, P = _model(scale=1.0)
    imgs, acts, stas = R.synthetic_batch(1, 5)
    m.train = False
    m([imgs, acts, stas], 0)
    a = np.stack(m.gen_images)
    imgs2 = imgs.copy(); imgs2[2:4] = 0.5                      # frames 2,3 are never inputs in feed-self mode
    m2 = R.Model(10, params=P, dtype=np.float64, prefix='kat'); m2.train = False
    m2([imgs2, acts, stas], 0)
    assert np.array_equal(a, np.stack(m2.gen_images))
    imgs3 = imgs.copy(); imgs3[1] = 0.5                        # frame 1 IS an input (warm start)
    m3 = R.Model(10, params=P, dtype=np.float64, prefix='kat'); m3.train = False
    m3([imgs3, acts, stas], 0)
    assert not np.array_equal(a[1:], np.stack(m3.gen_images)[1:])


def test_concat_examples_layout():
    # TM:51-71: NHWC per-sequence arrays -> time-major NCHW
    rs = np.random.RandomState(0)
    batch = [(rs.rand(4, 8, 8, 3), rs.rand(4, 5), rs.rand(4, 5)) for _ in range(3)]
    img, act, sta = R.concat_examples(batch)
    assert img.shape == (4, 3, 3, 8, 8) and act.shape == (4, 3, 5) and sta.shape == (4, 3, 5)
    assert img[2, 1, 0, 5, 6] == batch[1][0][2, 5, 6, 0]
    assert act[3, 2, 4] == batch[2][1][3, 4]


def test_stp_shares_one_transform_and_identity():
    # quirk 6: 9 identical warps; zero Linear -> identity warp reproduces prev
    m, P = _model('STP', 10)
    P['model/identity_params/W'][...] = 0; P['model/identity_params/b'][...] = 0
    m.load_params(P)
    rs = np.random.RandomState(2)
    prev = rs.rand(2, 3, 64, 64)
    layers, _ = m._stp(rs.randn(2, 64, 64, 64), rs.randn(2, 128, 8, 8), prev)
    assert len(layers) == 10
    for t in layers[1:]:
        assert np.allclose(t, prev, atol=1e-12)


def test_dna_requires_single_mask():
    m, _ = _model('DNA', 1)
    m.num_masks = 2
    with pytest.raises(ValueError):
        m._dna(np.zeros((1, 64, 64, 64)), None, np.zeros((1, 3, 64, 64)))


def test_dna_shifted_stack_is_detached():
    # TM:404 `kernel_inputs.append(tmp.data)`: the 25 shifted copies of the previous frame leave the graph, so through the
    # DNA head d out / d prev is the mask-0 term alone: d out[b,c,p] / d prev[b,c,p] = m0[b,p] (TM:725), nothing via the kernels.
    import torch
    from oracle.torch_restatement import TorchModel
    P = R.init_params(seed=3, dtype=np.float64, scale=1.0, model_type='DNA', num_masks=1)
    tm = TorchModel(1, is_cdna=False, is_dna=True, params=P)
    rs = np.random.RandomState(5)
    prev = torch.tensor(rs.rand(2, 3, 64, 64))
    sa = torch.tensor(rs.randn(2, 10) * 0.1)
    head = prev.clone().requires_grad_(True)
    out, _ = tm._step(prev, sa, prev_head=head)          # trunk reads the constant `prev`, the head reads the leaf
    go = torch.tensor(rs.randn(2, 3, 64, 64))
    (out * go).sum().backward()
    m0 = tm.last['masks'][:, 0:1].detach()
    assert torch.allclose(head.grad, m0 * go, rtol=0, atol=1e-14)
    # and the term the reference drops is not small: with the stack attached the gradient would differ visibly
    kn = torch.relu(tm.last['enc7'].detach() - 1e-12) + 1e-12
    kn = kn / kn.sum(1, keepdim=True)
    assert float((kn[:, 12:13] * tm.last['masks'][:, 1:2].detach() * go).abs().max()) > 1e-3


def test_dna_one_hot_kernel_is_a_shift_with_the_slice_quirk():
    # TM:395-402, derived by hand: prev is padded by 2 on every side; tap (xk, yk) takes rows xk..H-1 and columns yk..W-1 of the PADDED
    # image (the slice ends at the unpadded size: the quirk), i.e. prev shifted by (xk - 2, yk - 2) and cut to H - xk rows / W - yk
    # columns, then zero-filled at the bottom / right.  With enc7 one-hot at k = xk * 5 + yk the DNA output is exactly that image.
    m, P = _model('DNA', 1)
    rs = np.random.RandomState(4)
    B, H, W = 2, 64, 64
    prev = rs.rand(B, 3, H, W)
    enc6 = np.ones((B, 64, H, W))
    for xk, yk in ((0, 0), (2, 2), (4, 1), (1, 3), (4, 4)):
        P['model/enc7/W'][...] = 0; P['model/enc7/b'][...] = 0
        P['model/enc7/b'][xk * 5 + yk] = 1.0
        m.load_params(P)
        (out,), enc7 = m._dna(enc6, None, prev)
        exp = np.zeros_like(prev)
        for y in range(H - xk):
            sy = y + xk - 2
            if not 0 <= sy < H:
                continue
            for x in range(W - yk):
                sx = x + yk - 2
                if 0 <= sx < W:
                    exp[:, :, y, x] = prev[:, :, sy, sx]
        assert np.abs(out - exp).max() < 1e-9, (xk, yk)     # the 24 other taps weigh RELU_SHIFT = 1e-12 each


def test_smear_and_state_predictor():
    # TM:556-567: state_action is tiled over the 8x8 map and concatenated BEHIND enc2's 64 channels in front of the 1x1 enc3;
    # TM:676: state_action = concat(action, current_state) in that order; TM:730-731: the PREDICTED state is fed back.
    P = R.init_params(seed=2, dtype=np.float64, scale=1.0)
    P['enc3/W'][:, :64] = 0.0                                     # enc3 sees the smear only
    m = R.Model(10, params=P, dtype=np.float64, prefix='kat'); m.train = False
    imgs, acts, stas = R.synthetic_batch(2, 4)
    m([imgs, acts, stas], 0, tap_steps=(0, 1))
    W3, b3 = P['enc3/W'][:, 64:, 0, 0], P['enc3/b']              # (64, 10)
    Wc, bc = P['current_state/W'], P['current_state/b']          # (5, 10)
    cur = stas[0].astype(np.float64)                              # TM:646
    for t in (0, 1):
        sa = np.concatenate((acts[t].astype(np.float64), cur), axis=1)
        e3 = np.maximum(sa @ W3.T + b3, 0.0)                      # constant over the map
        assert np.abs(m.taps[t]['enc3'] - e3[:, :, None, None]).max() < 1e-12
        cur = sa @ Wc.T + bc
        assert np.abs(m.gen_states[t] - cur).max() < 1e-12
    # the ground-truth state of step 1 is never read (TM:675 comment "Predicted state is always fed back in")
    stas2 = stas.copy(); stas2[1:] += 7.0
    m2 = R.Model(10, params=P, dtype=np.float64, prefix='kat'); m2.train = False
    m2([imgs, acts, stas2], 0)
    assert np.array_equal(np.stack(m2.gen_images), np.stack(m.gen_images))
