"""GPU parity tests, per op: HIP kernel (through the C ABI) vs the float64 CPU oracle on the same
seeded inputs.  Tolerances are absolute on O(1) activations; fp32 MFMA accumulation over K <= 4800
gives ~1e-6 error, gates at 2e-5."""
import numpy as np
import pytest

from oracle import restatement as R

pytestmark = pytest.mark.gpu

TOL = 2e-5


@pytest.fixture(scope='module')
def ops():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    import hip_ops
    return hip_ops


def _lstm_ref(x, h, c, W, b):
    g = R.conv2d(np.concatenate((x, h), 1), W, b, 1, 2)
    j, i, f, o = np.split(g, 4, axis=1)
    cn = c * R.sigmoid(f + 1.0) + R.sigmoid(i) * np.tanh(j)
    return np.tanh(cn) * R.sigmoid(o), cn


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
@pytest.mark.parametrize('B,cx,C,H', [(2, 32, 32, 32), (2, 32, 64, 16), (3, 64, 128, 8), (2, 128, 64, 16), (1, 96, 32, 32), (5, 64, 64, 16)])
def test_convlstm_parity(ops, B, cx, C, H, variant):
    rs = np.random.RandomState(B * 100 + C)
    x = rs.randn(B, cx, H, H); h = rs.randn(B, C, H, H) * 0.5; c = rs.randn(B, C, H, H)
    W = rs.randn(4 * C, cx + C, 5, 5) / np.sqrt(25 * (cx + C)); b = rs.randn(4 * C) * 0.1
    hr, cr = _lstm_ref(x, h, c, W, b)
    hg, cg = ops.convlstm(x, h, c, W, b, variant)
    assert np.abs(hg - hr).max() < TOL and np.abs(cg - cr).max() < TOL


@pytest.mark.parametrize("variant", [0, 2, 3])
@pytest.mark.parametrize('B,cx,C,H,Wd', [(2, 32, 32, 12, 20), (3, 64, 32, 10, 6), (1, 32, 64, 7, 24)])
def test_convlstm_on_maps_that_are_not_powers_of_two(ops, B, cx, C, H, Wd, variant):
    # the kernels divide by Hg * Wg, Wg and the tile counts through multiply-high + shift (pivp_fastdiv, round 6): divisors that are not
    # powers of two, tiles that straddle rows and samples
    rs = np.random.RandomState(B * 100 + C + Wd)
    x = rs.randn(B, cx, H, Wd); h = rs.randn(B, C, H, Wd) * 0.5; c = rs.randn(B, C, H, Wd)
    W = rs.randn(4 * C, cx + C, 5, 5) / np.sqrt(25 * (cx + C)); b = rs.randn(4 * C) * 0.1
    hr, cr = _lstm_ref(x, h, c, W, b)
    hg, cg = ops.convlstm(x, h, c, W, b, variant)
    assert np.abs(hg - hr).max() < TOL and np.abs(cg - cr).max() < TOL


@pytest.mark.parametrize('variant', [1, 2, 3, 4])
def test_convlstm_first_step_skips_zero_h(ops, variant):
    # h_prev = NULL (all zeros after reset_state, TM:254-257): the h half of K is skipped, result identical
    rs = np.random.RandomState(77)
    B, cx, C, H = 2, 64, 128, 8
    x = rs.randn(B, cx, H, H); h = np.zeros((B, C, H, H)); c = np.zeros((B, C, H, H))
    W = rs.randn(4 * C, cx + C, 5, 5) / np.sqrt(25 * (cx + C)); b = rs.randn(4 * C) * 0.1
    hr, cr = _lstm_ref(x, h, c, W, b)
    hg, cg = ops.convlstm(x, h, c, W, b, variant, h_is_zero=True)
    hz, cz = ops.convlstm(x, h, c, W, b, variant, h_is_zero=False)
    assert np.abs(hg - hr).max() < TOL and np.abs(cg - cr).max() < TOL
    if variant in (0, 3, 4):   # tiles that split K over two wave groups: 25 + 25 chunks without h, 75 + 75 with it -- other partial sums
        assert np.abs(hg - hz).max() < 1e-6 and np.abs(cg - cz).max() < 1e-6
    else:
        assert np.array_equal(hg, hz) and np.array_equal(cg, cz)      # skipping adds exact zeros: bit-identical


@pytest.mark.parametrize('variant', [0, 1, 2, 3, 4])
@pytest.mark.parametrize('B,cx,C,H,expect_fused', [(2, 32, 32, 32, True), (3, 64, 128, 8, None), (2, 32, 64, 16, True),
                                                   (2, 32, 32, 6, False)])
def test_convlstm_layernorm_fused_stats(ops, B, cx, C, H, variant, expect_fused):
    # hidden = norm(lstm(x)) (TM:596-601): LayerNorm statistics come from the ConvLSTM epilogue's per-tile partials;
    # H = 6 (36 pixels per sample) has tiles straddling samples, so that case must fall back to the statistics pass
    rs = np.random.RandomState(B * 10 + H + variant)
    x = rs.randn(B, cx, H, H); h = rs.randn(B, C, H, H) * 0.5; c = rs.randn(B, C, H, H)
    W = rs.randn(4 * C, cx + C, 5, 5) / np.sqrt(25 * (cx + C)); b = rs.randn(4 * C) * 0.1
    gamma = 1.0 + 0.1 * rs.randn(C, H, H); beta = 0.1 * rs.randn(C, H, H)
    hr, cr = _lstm_ref(x, h, c, W, b)
    lr = R.layer_norm_conv2d(hr, gamma.reshape(-1), beta.reshape(-1), 1e-6)
    lg, hg, cg, fused = ops.convlstm_ln(x, h, c, W, b, gamma, beta, 1e-6, variant)
    assert np.abs(hg - hr).max() < TOL and np.abs(cg - cr).max() < TOL
    assert np.abs(lg - lr).max() < 1e-4      # LayerNorm divides by std(h) ~ 0.3: 3x the error of h
    bm = {1: 128, 2: 64, 3: 32, 4: 64}.get(variant)
    if bm is not None:
        assert fused == int((H * H) % bm == 0)
    elif expect_fused is not None:
        assert fused == int(expect_fused)
    # and identical to LayerNorm run on the kernel's own h with the separate statistics pass, up to rounding
    l2 = ops.layernorm(hg, gamma, beta, 1e-6, False)
    assert np.abs(lg - l2).max() < 2e-6


def test_convlstm_zero_weights_kat(ops):
    # zero weights -> c = c * sigmoid(1), h = tanh(c)/2
    x = np.random.RandomState(0).randn(2, 32, 8, 8)
    c = np.full((2, 32, 8, 8), 0.7); h = np.zeros((2, 32, 8, 8))
    hg, cg = ops.convlstm(x, h, c, np.zeros((128, 64, 5, 5)), np.zeros(128))
    ce = 0.7 / (1 + np.exp(-1.0))
    assert np.abs(cg - ce).max() < 1e-6 and np.abs(hg - np.tanh(ce) * 0.5).max() < 1e-6


@pytest.mark.parametrize('B,cin,cout,H', [(2, 32, 32, 32), (3, 64, 64, 16)])
def test_conv3x3s2_parity(ops, B, cin, cout, H):
    rs = np.random.RandomState(1)
    x = rs.randn(B, cin, H, H); W = rs.randn(cout, cin, 3, 3) / np.sqrt(9 * cin); b = rs.randn(cout) * 0.1
    ref = R.relu(R.conv2d(x, W, b, 2, 1))
    assert np.abs(ops.conv3x3s2(x, W, b, True) - ref).max() < TOL


@pytest.mark.parametrize('B,cin,cout,H,relu', [(2, 128, 128, 8, True), (2, 96, 96, 16, True), (3, 64, 64, 32, False)])
def test_deconv3x3s2_parity(ops, B, cin, cout, H, relu):
    rs = np.random.RandomState(2)
    x = rs.randn(B, cin, H, H); W = rs.randn(cin, cout, 3, 3) / np.sqrt(9 * cin); b = rs.randn(cout) * 0.1
    ref = R.deconv2d(x, W, b, 2, 1, (2 * H, 2 * H))
    if relu:
        ref = R.relu(ref)
    assert np.abs(ops.deconv3x3s2(x, W, b, relu) - ref).max() < TOL


@pytest.mark.parametrize('B,c_ln,c1,cout,H,relu', [(4, 64, 32, 96, 16, True), (2, 32, 32, 64, 32, False), (5, 64, 0, 64, 16, True)])
def test_deconv3x3s2_of_norm_concat_in_one_launch(ops, B, c_ln, c1, cout, H, relu):
    """enc5 / enc6 of an inference rollout (TM:565-566, 574-575): deconv(concat(LayerNorm(hidden), skip)) with the norm applied while the
    conv stages its input: within TOL of the float64 oracle and BIT-identical to the two launches a training plan runs, in every precision."""
    rs = np.random.RandomState(7 + c_ln)
    h = rs.randn(B, c_ln, H, H) * 0.3 + 0.1
    x1 = rs.randn(B, c1, H, H) if c1 else None
    n = c_ln * H * H
    g = 1 + 0.1 * rs.randn(n); be = 0.1 * rs.randn(n)
    W = rs.randn(c_ln + c1, cout, 3, 3) / np.sqrt(9 * (c_ln + c1)); b = rs.randn(cout) * 0.1
    hn = R.layer_norm_conv2d(h, g, be, 1e-6)
    ref = R.deconv2d(np.concatenate((hn, x1), 1) if c1 else hn, W, b, 2, 1, (2 * H, 2 * H))
    if relu:
        ref = R.relu(ref)
    one = ops.deconv3x3s2_of_norm_concat(h, g, be, x1, W, b, relu)
    assert np.abs(one - ref).max() < TOL
    for precision in (0, 1, 2):
        a = one if precision == 0 else ops.deconv3x3s2_of_norm_concat(h, g, be, x1, W, b, relu, precision=precision)
        two = ops.deconv3x3s2_of_norm_concat(h, g, be, x1, W, b, relu, precision=precision, fused=False)
        assert np.array_equal(a, two), precision


def test_enc0_parity(ops):
    rs = np.random.RandomState(3)
    img = rs.rand(3, 3, 64, 64); W = rs.randn(32, 3, 5, 5) / np.sqrt(75); b = rs.randn(32) * 0.1
    assert np.abs(ops.conv_enc0(img, W, b) - R.conv2d(img, W, b, 2, 2)).max() < TOL


@pytest.mark.parametrize('B,C,H,relu', [(2, 32, 32, True), (3, 64, 16, False), (2, 128, 8, False), (2, 64, 64, True)])
def test_layernorm_parity(ops, B, C, H, relu):
    rs = np.random.RandomState(4)
    x = rs.randn(B, C, H, H) * 2 + 0.7
    n = C * H * H
    g = 1 + 0.1 * rs.randn(n); be = 0.1 * rs.randn(n)
    ref = R.layer_norm_conv2d(x, g, be, 1e-6)
    if relu:
        ref = R.relu(ref)
    assert np.abs(ops.layernorm(x, g, be, 1e-6, relu) - ref).max() < TOL


def test_layernorm_large_offset_is_stable(ops):
    # mean >> std: a sum / sum-of-squares formulation would lose the variance here
    rs = np.random.RandomState(5)
    x = rs.randn(2, 32, 32, 32) * 0.01 + 100.0
    n = 32 * 32 * 32
    ref = R.layer_norm_conv2d(x, np.ones(n), np.zeros(n), 1e-6)
    got = ops.layernorm(x, np.ones(n), np.zeros(n), 1e-6, False)
    assert np.abs(got - ref).max() < 2e-2 * 0.05 + 1e-3   # fp32 input quantisation of x dominates (ulp(100)/0.01)


def test_enc3_state_parity(ops):
    rs = np.random.RandomState(6)
    B = 3
    e2 = rs.randn(B, 64, 8, 8); act = rs.randn(B, 5) * 0.1; st = rs.randn(B, 5) * 0.1
    W3 = rs.randn(64, 74, 1, 1) / np.sqrt(74); b3 = rs.randn(64) * 0.1
    Wcs = rs.randn(5, 10); bcs = rs.randn(5)
    sa = np.concatenate((act, st), 1)
    xin = np.concatenate((e2, np.tile(sa[:, :, None, None], (1, 1, 8, 8))), 1)
    ref = R.relu(R.conv2d(xin, W3, b3, 1, 0))
    e3, s = ops.enc3_state(e2, act, st, W3, b3, Wcs, bcs)
    assert np.abs(e3 - ref).max() < TOL and np.abs(s - R.linear(sa, Wcs, bcs)).max() < TOL
    ref2 = R.relu(R.conv2d(e2, W3[:, :64], b3, 1, 0))
    e3b, _ = ops.enc3_state(e2, act, st, W3[:, :64], b3, Wcs, bcs, use_state=False)
    assert np.abs(e3b - ref2).max() < TOL


@pytest.mark.parametrize('mt,nm,ne', [(0, 10, 3), (1, 10, 3), (2, 1, 25)])
def test_heads_parity(ops, mt, nm, ne):
    rs = np.random.RandomState(7)
    B = 2
    e6 = R.relu(rs.randn(B, 64, 64, 64))
    Wm = rs.randn(64, nm + 1, 1, 1) / 8; bm = rs.randn(nm + 1) * 0.1
    We = rs.randn(64, ne, 1, 1) / 8; be = rs.randn(ne) * 0.1
    logits, enc7, layer0 = ops.heads(e6, Wm, bm, We, be, nm, mt)
    rl = R.relu(R.deconv2d(e6, Wm, bm))
    r7 = R.deconv2d(e6, We, be)
    if mt != 1:
        r7 = R.relu(r7)
    assert np.abs(logits - rl).max() < TOL and np.abs(enc7 - r7).max() < TOL
    if mt != 2:
        assert np.abs(layer0 - R.sigmoid(r7)).max() < TOL


@pytest.mark.parametrize('mt,nm,ne,H,W,finisher,train', [
    (0, 10, 3, 64, 64, True, False), (0, 10, 3, 64, 64, True, True), (0, 10, 3, 64, 64, False, False), (0, 4, 3, 64, 64, True, True),
    (1, 10, 3, 64, 64, True, True), (1, 10, 3, 64, 64, False, False), (2, 1, 25, 64, 64, True, True),
    (0, 10, 3, 128, 128, True, False), (0, 10, 3, 32, 48, True, True), (1, 3, 3, 16, 64, True, False)])
def test_frame_head_matches_the_separate_kernels(ops, mt, nm, ne, H, W, finisher, train):
    """pivp_frame_head (norm_enc6 + relu + 1x1 heads + the motion head's finisher + flat softmax + transform + compositing in one launch)
    against pivp_layernorm -> pivp_heads -> pivp_cdna_kernels / pivp_stp_params -> pivp_composite on the same device buffers: every
    output BIT-identical.  The halo logic is what this pins: the flat-(num_masks+1) softmax groups of a 4-row band reach into the
    neighbouring rows and, at the top / bottom band, into the neighbouring mask PLANE (TM:720-722); random logits would make any
    misfiled halo element visible in the band's first / last groups.  128 x 128: the Linear has 512 K slices, so the finisher stays a
    launch of its own (fits == 2) and the kernels arrive through aux."""
    rs = np.random.RandomState(100 * mt + nm + H)
    B = 3
    e6raw = rs.randn(B, 64, H, W) * 1.5 + 0.3
    gamma = 1.0 + 0.1 * rs.randn(64, H, W); beta = 0.1 * rs.randn(64, H, W)
    Wm = rs.randn(64, nm + 1, 1, 1) / 4; bm = rs.randn(nm + 1) * 0.1
    We = rs.randn(64, ne, 1, 1) / 8; be = rs.randn(ne) * 0.1
    prev = rs.rand(B, 3, H, W)
    h5 = rs.randn(B, 128, H // 8, W // 8)
    K = 128 * (H // 8) * (W // 8)
    nout = 25 * nm if mt == 0 else 100
    Wh = rs.randn(nout, K) / np.sqrt(K); bh = rs.randn(nout) * 0.1
    W2 = rs.randn(6, 100) * 0.02; b2 = rs.randn(6) * 0.01
    sep, fused = ops.frame_head_pair(e6raw, gamma, beta, 1e-6, Wm, bm, We, be, prev, h5, Wh, bh, W2, b2, nm, mt, 0, finisher, train)
    assert np.array_equal(sep['out'], fused['out']), np.abs(sep['out'] - fused['out']).max()
    assert np.array_equal(sep['masks'], fused['masks'])
    assert np.array_equal(sep['enc7'], fused['enc7'])
    if train:
        assert np.array_equal(sep['logits'], fused['logits']) and np.array_equal(sep['enc6'], fused['enc6'])
        if mt != 2:
            assert np.array_equal(sep['layer0'], fused['layer0'])
        assert np.all(np.isfinite(fused['stat'])) and np.all(fused['stat'][:, 1] > 0)
    else:
        assert np.all(fused['logits'] == 7.0) and np.all(fused['enc6'] == 7.0)       # optional outputs: untouched when not asked for
    if fused['kerns'] is not None:
        assert np.array_equal(sep['kerns'].reshape(B, -1), fused['kerns'].reshape(B, -1))
    # and against the float64 oracle, end to end
    e6 = R.relu(R.layer_norm_conv2d(e6raw, gamma.reshape(-1), beta.reshape(-1), 1e-6))
    rl = R.relu(R.deconv2d(e6, Wm, bm))
    masks = _masks_ref(rl)
    assert np.abs(fused['masks'] - masks).max() < 2e-5


def test_cdna_kernels_parity_and_kat(ops):
    rs = np.random.RandomState(8)
    B = 4
    h5 = rs.randn(B, 128, 8, 8)
    W = rs.randn(250, 8192) / np.sqrt(8192); b = rs.randn(250) * 0.1
    k = R.linear(h5.reshape(B, -1), W, b).reshape(B, 10, 25)
    k = R.relu(k - 1e-12) + 1e-12
    k = (k / k.sum(2, keepdims=True)).reshape(B, 10, 5, 5)
    got = ops.cdna_kernels(h5, W, b, 10)
    assert np.abs(got - k).max() < 1e-6
    assert np.abs(got.sum(axis=(2, 3)) - 1).max() < 1e-6         # KAT (1): each kernel sums to 1
    uni = ops.cdna_kernels(h5, np.zeros((250, 8192)), np.full(250, 0.37), 10)
    assert np.abs(uni - 1 / 25.0).max() < 1e-7                   # KAT (1): equal logits -> 1/25


def _masks_ref(logits):
    B, NP, H, W = logits.shape
    return R.softmax_axis1(logits.reshape(-1, NP)).reshape(B, NP, H, W)


def test_composite_cdna_parity_and_kats(ops):
    rs = np.random.RandomState(9)
    B, NM = 3, 10
    prev = rs.rand(B, 3, 64, 64); logits = R.relu(rs.randn(B, NM + 1, 64, 64) * 2); l0 = rs.rand(B, 3, 64, 64)
    k = rs.rand(B, NM, 5, 5); k /= k.sum(axis=(2, 3), keepdims=True)
    masks = _masks_ref(logits)
    t = R.depthwise_conv2d(prev.transpose(1, 0, 2, 3), k.transpose(1, 0, 2, 3), 2).reshape(3, B, NM, 64, 64).transpose(2, 1, 0, 3, 4)
    ref = prev * masks[:, 0:1] + l0 * masks[:, 1:2]
    for q in range(NM - 1):
        ref = ref + t[q] * masks[:, q + 2:q + 3]
    out, mg = ops.composite(prev, logits, l0, k, NM, 0)
    assert np.abs(mg - masks).max() < 1e-6                       # flat-11 softmax quirk reproduced
    assert np.abs(out - ref).max() < 1e-5
    # KAT (3): constant logits -> every mask 1/11
    _, mc = ops.composite(prev, np.full_like(logits, 0.25), l0, k, NM, 0)
    assert np.abs(mc - 1 / 11.0).max() < 1e-7
    # KAT (4): the 10th kernel has no influence
    k2 = k.copy(); k2[:, 9] = rs.rand(B, 5, 5)
    out2, _ = ops.composite(prev, logits, l0, k2, NM, 0)
    assert np.array_equal(out, out2)
    # KAT (2): delta kernels -> identity / shift with zero fill; one-hot masks isolate a layer
    kd = np.zeros((B, NM, 5, 5)); kd[:, :, 2, 2] = 1.0; kd[:, 0, 2, 2] = 0.0; kd[:, 0, 0, 4] = 1.0
    big = np.zeros((B, NM + 1, 64, 64)); big[:, 2] = 60.0        # mask 2 (layer T_0) ~ 1 wherever its group is pure
    out3, m3 = ops.composite(prev, big, l0, kd, NM, 0)
    shifted = np.zeros_like(prev); shifted[:, :, 2:, :-2] = prev[:, :, :-2, 2:]
    ref3 = prev * m3[:, 0:1] + l0 * m3[:, 1:2] + shifted * m3[:, 2:3]
    for q in range(1, NM - 1):
        ref3 = ref3 + prev * m3[:, q + 2:q + 3]
    assert np.abs(out3 - ref3).max() < 1e-6


@pytest.mark.parametrize('zero', [0, 1])
def test_composite_stp_parity(ops, zero):
    rs = np.random.RandomState(10)
    B, NM = 2, 10
    prev = rs.rand(B, 3, 64, 64); logits = R.relu(rs.randn(B, NM + 1, 64, 64)); l0 = rs.rand(B, 3, 64, 64)
    theta = np.tile(np.array([[1.0, 0, 0, 0, 1.0, 0]]), (B, 1)) + rs.randn(B, 6) * 0.15
    grid = R.spatial_transformer_grid(theta.reshape(B, 2, 3), (64, 64))
    warp = R.spatial_transformer_sampler(prev, grid, 'zeros' if zero else 'clamp')
    masks = _masks_ref(logits)
    ref = prev * masks[:, 0:1] + l0 * masks[:, 1:2] + warp * masks[:, 2:].sum(1, keepdims=True)
    out, _ = ops.composite(prev, logits, l0, theta, NM, 1, zero)
    assert np.abs(out - ref).max() < 3e-5


def test_composite_dna_parity(ops):
    rs = np.random.RandomState(11)
    B = 2
    prev = rs.rand(B, 3, 64, 64); logits = R.relu(rs.randn(B, 2, 64, 64)); e7 = R.relu(rs.randn(B, 25, 64, 64))
    m = R.Model(1, is_cdna=False, is_dna=True, dtype=np.float64)
    m.p = {'model/enc7/W': np.zeros((64, 25, 1, 1)), 'model/enc7/b': np.zeros(25)}
    # drive _dna's arithmetic with our enc7 by bypassing the 1x1: reuse its code path on a crafted enc6
    pad = np.pad(prev, ((0, 0), (0, 0), (2, 2), (2, 2)))
    ins = []
    for xk in range(5):
        for yk in range(5):
            tmp = pad[:, :, xk:64, yk:64]
            ins.append(np.pad(tmp, ((0, 0), (0, 0), (0, xk), (0, yk)))[:, None])
    kin = np.concatenate(ins, 1)
    kn = R.relu(e7 - 1e-12) + 1e-12
    kn = kn / kn.sum(1, keepdims=True)
    dna = (kin * kn[:, :, None]).sum(1)
    masks = _masks_ref(logits)
    ref = prev * masks[:, 0:1] + dna * masks[:, 1:2]
    out, _ = ops.composite(prev, logits, None, e7, 1, 2)
    assert np.abs(out - ref).max() < 1e-5


def test_stp_params_parity(ops):
    rs = np.random.RandomState(12)
    B = 3
    h5 = rs.randn(B, 128, 8, 8)
    W1 = rs.randn(100, 8192) / 90; b1 = rs.randn(100) * 0.1; W2 = rs.randn(6, 100) / 10; b2 = rs.randn(6) * 0.1
    s1 = R.relu(R.linear(h5.reshape(B, -1), W1, b1))
    ref = R.linear(s1, W2, b2) + np.array([1.0, 0, 0, 0, 1.0, 0])
    assert np.abs(ops.stp_params(h5, W1, b1, W2, b2) - ref).max() < 1e-5


def test_select_frames(ops):
    gt = np.zeros((5, 3, 64, 64)); gen = np.ones((5, 3, 64, 64))
    out = ops.select_frames(gt, gen, [1, 0, 0, 1, 0])
    assert out[:, 0, 0, 0].tolist() == [0, 1, 1, 0, 1] and np.all(out[1] == 1) and np.all(out[3] == 0)
