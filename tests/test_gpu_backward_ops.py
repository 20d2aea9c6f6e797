"""GPU parity tests of the backward kernels, per op: HIP (through the C ABI) vs PyTorch autograd in float64 on the CPU,
on the same seeded inputs.  Gradients are O(1)-scaled test cotangents; tolerances are relative to the gradient scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def env():
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    import pivp_amd
    from pivp_amd import _lib
    return pivp_amd, _lib, _lib.load()


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).to(DEV)


def _nhwc(a):
    return _t(np.asarray(a).transpose(0, 2, 3, 1))


def _st():
    return torch.cuda.current_stream().cuda_stream


def _rel(a, b):
    return np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max() / (np.abs(b).max() + 1e-30)


@pytest.mark.parametrize('B,cx,C,H,first', [(2, 32, 32, 64, False), (2, 32, 32, 32, False), (2, 32, 64, 16, False), (3, 64, 128, 8, False),
                                            (2, 96, 32, 32, False), (2, 128, 64, 16, True)])
def test_convlstm_backward(env, B, cx, C, H, first):
    pivp, _lib, lib = env
    rs = np.random.RandomState(C + cx)
    x = rs.randn(B, cx, H, H); h = np.zeros((B, C, H, H)) if first else rs.randn(B, C, H, H) * 0.5
    c = np.zeros((B, C, H, H)) if first else rs.randn(B, C, H, H)
    W = rs.randn(4 * C, cx + C, 5, 5) / np.sqrt(25 * (cx + C)); b = rs.randn(4 * C) * 0.1
    dh = rs.randn(B, C, H, H); dh2 = rs.randn(B, C, H, H) * 0.5; dcn = rs.randn(B, C, H, H)
    # reference: autograd
    tx, th, tc = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in (x, h, c)]
    tW = torch.tensor(W, dtype=torch.float64, requires_grad=True); tb = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    g = F.conv2d(torch.cat((tx, th), 1), tW, tb, padding=2)
    j, i, f, o = torch.split(g, C, dim=1)
    cn = tc * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
    hn = torch.tanh(cn) * torch.sigmoid(o)
    (hn * torch.tensor(dh + dh2) + cn * torch.tensor(dcn)).sum().backward()
    # HIP
    xd, hd, cd = _nhwc(x), _nhwc(h), _nhwc(c)
    wd, bd = _t(pivp.to_internal('lstm1/conv/W', W)), _t(b)
    c_out = torch.empty_like(cd); h_out = torch.empty_like(hd)
    M = B * H * H
    gates = torch.empty((M, 4 * C), dtype=torch.float32, device=DEV)
    _lib.check(lib.pivp_convlstm_train(xd.data_ptr(), cx, cx, None if first else hd.data_ptr(), C, wd.data_ptr(), bd.data_ptr(),
                                       cd.data_ptr(), c_out.data_ptr(), h_out.data_ptr(), gates.data_ptr(), B, H, H, _st()), 'fwd')
    # dh_b arrives as the last C channels of a wider buffer (the next step's d_in), exercise that stride
    wide = torch.zeros((M, cx + C), dtype=torch.float32, device=DEV)
    wide[:, cx:] = _nhwc(dh2).reshape(M, C)
    dha = _nhwc(dh)
    dc = _nhwc(dcn).clone()
    dG = torch.empty((M, 4 * C), dtype=torch.float32, device=DEV)
    wt = torch.empty_like(wd)
    d_in = torch.empty((M, cx + C), dtype=torch.float32, device=DEV)
    dW = torch.zeros_like(wd); db = torch.zeros_like(bd)
    _lib.check(lib.pivp_convlstm_backward(xd.data_ptr(), cx, cx, None if first else hd.data_ptr(), C, wd.data_ptr(), gates.data_ptr(),
                                          cd.data_ptr(), c_out.data_ptr(), dha.data_ptr(), C, (wide.data_ptr() + cx * 4), cx + C,
                                          dc.data_ptr(), 1, dG.data_ptr(), wt.data_ptr(), d_in.data_ptr(), dW.data_ptr(), db.data_ptr(),
                                          B, H, H, _st()), 'bwd')
    torch.cuda.synchronize()
    din = d_in.cpu().numpy().reshape(B, H, H, cx + C).transpose(0, 3, 1, 2)
    assert _rel(din[:, :cx], tx.grad.numpy()) < 2e-5
    if not first:
        assert _rel(din[:, cx:], th.grad.numpy()) < 2e-5
    assert _rel(dc.cpu().numpy().reshape(B, H, H, C).transpose(0, 3, 1, 2), tc.grad.numpy()) < 2e-5
    dW_ref = tW.grad.numpy().copy()
    if first:
        dW_ref[:, cx:] = 0          # the skipped zero-h K range gets no gradient (h == 0 => it is exactly 0 anyway)
    got_dW = pivp.from_internal('lstm1/conv/W', dW.cpu().numpy(), W.shape)
    assert _rel(got_dW, dW_ref) < 2e-5
    assert _rel(db.cpu().numpy(), tb.grad.numpy()) < 2e-5


@pytest.mark.parametrize('mode,B,cin,cout,H', [(0, 2, 32, 32, 32), (0, 3, 64, 64, 16), (1, 2, 128, 128, 8), (1, 2, 96, 96, 16), (1, 2, 64, 64, 32), (1, 2, 64, 64, 64),
                                               (0, 2, 32, 32, 64)])
def test_conv_deconv_backward(env, mode, B, cin, cout, H):
    pivp, _lib, lib = env
    rs = np.random.RandomState(cin + mode)
    x = rs.randn(B, cin, H, H)
    Ho = 2 * H if mode else H // 2
    dy = rs.randn(B, cout, Ho, Ho)
    if mode:
        W = rs.randn(cin, cout, 3, 3) / np.sqrt(9 * cin); key = 'enc4/W'
    else:
        W = rs.randn(cout, cin, 3, 3) / np.sqrt(9 * cin); key = 'enc1/W'
    tx = torch.tensor(x, dtype=torch.float64, requires_grad=True); tW = torch.tensor(W, dtype=torch.float64, requires_grad=True)
    tb = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    y = F.conv_transpose2d(tx, tW, tb, stride=2, padding=1, output_padding=1) if mode else F.conv2d(tx, tW, tb, stride=2, padding=1)
    (y * torch.tensor(dy)).sum().backward()
    xd, dyd = _nhwc(x), _nhwc(dy)
    wd = _t(pivp.to_internal(key, W))
    wt = torch.empty_like(wd); dW = torch.zeros_like(wd); db = torch.zeros(cout, dtype=torch.float32, device=DEV)
    prior = rs.randn(B, H, H, cin).astype(np.float32)
    dx = _t(prior)
    _lib.check(lib.pivp_conv_backward(mode, xd.data_ptr(), cin, cin, wd.data_ptr(), dyd.data_ptr(), cout, cout, wt.data_ptr(),
                                      dx.data_ptr(), cin, 1, dW.data_ptr(), db.data_ptr(), B, H, H, _st()), 'conv_backward')
    torch.cuda.synchronize()
    got_dx = (dx.cpu().numpy() - prior).transpose(0, 3, 1, 2)      # accum_dx = 1 adds into the existing buffer
    assert _rel(got_dx, tx.grad.numpy()) < 2e-5
    assert _rel(pivp.from_internal(key, dW.cpu().numpy(), W.shape), tW.grad.numpy()) < 2e-5
    assert _rel(db.cpu().numpy(), tb.grad.numpy()) < 2e-5


@pytest.mark.parametrize('B,C,H,relu', [(2, 32, 32, True), (3, 64, 16, False), (2, 128, 8, False), (2, 64, 64, True), (2, 64, 128, True),
                                       (2, 32, 64, False)])
def test_layernorm_backward(env, B, C, H, relu):
    pivp, _lib, lib = env
    rs = np.random.RandomState(C)
    n = C * H * H
    x = rs.randn(B, C, H, H) * 2 + 0.5; g = 1 + 0.1 * rs.randn(n); be = 0.1 * rs.randn(n); dy = rs.randn(B, C, H, H)
    tx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    tg = torch.tensor(g, dtype=torch.float64, requires_grad=True); tb = torch.tensor(be, dtype=torch.float64, requires_grad=True)
    y = F.layer_norm(tx.reshape(B, -1), (n,), tg, tb, 1e-6).reshape(x.shape)
    if relu:
        y = F.relu(y)
    (y * torch.tensor(dy)).sum().backward()
    perm = lambda v: _t(np.asarray(v).reshape(C, H * H).T)
    xd, gd, bd = _nhwc(x), perm(g), perm(be)
    ldo = C + 32                                               # output / dy live in a wider concat buffer
    out = torch.zeros((B, H, H, ldo), dtype=torch.float32, device=DEV)
    scratch = torch.empty(lib.pivp_layernorm_scratch_floats(B, n), dtype=torch.float32, device=DEV)
    stat = torch.empty((B, 2), dtype=torch.float32, device=DEV)
    _lib.check(lib.pivp_layernorm_train(xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), out.data_ptr() + 32 * 4, scratch.data_ptr(),
                                        stat.data_ptr(), B, n, C, ldo, 1e-6, int(relu), _st()), 'ln fwd')
    dyw = torch.zeros((B, H, H, ldo), dtype=torch.float32, device=DEV)
    dyw[..., 32:] = _nhwc(dy)
    dx = torch.empty_like(xd); dg = torch.zeros_like(gd); dbt = torch.zeros_like(bd)
    sc2 = torch.empty(lib.pivp_layernorm_backward_scratch_floats(B, n), dtype=torch.float32, device=DEV)
    _lib.check(lib.pivp_layernorm_backward(dyw.data_ptr() + 32 * 4, ldo, out.data_ptr() + 32 * 4, ldo, xd.data_ptr(), stat.data_ptr(),
                                           gd.data_ptr(), sc2.data_ptr(), dx.data_ptr(), dg.data_ptr(), dbt.data_ptr(), B, n, C,
                                           int(relu), _st()), 'ln bwd')
    torch.cuda.synchronize()
    assert _rel(dx.cpu().numpy().transpose(0, 3, 1, 2), tx.grad.numpy()) < 3e-5
    unperm = lambda t: t.cpu().numpy().reshape(H * H, C).T.ravel()
    assert _rel(unperm(dg), tg.grad.numpy()) < 3e-5 and _rel(unperm(dbt), tb.grad.numpy()) < 3e-5


def test_adam_step_matches_chainer_rule(env):
    pivp, _lib, lib = env
    from oracle.torch_restatement import chainer_adam_step
    rs = np.random.RandomState(0)
    n = 100003
    p = rs.randn(n); g = rs.randn(n) * 0.01
    P = {'w': p.copy()}; G = {'w': g.copy()}; M = {'w': np.zeros(n)}; V = {'w': np.zeros(n)}
    pd, gd = _t(p), _t(g); md = torch.zeros_like(pd); vd = torch.zeros_like(pd)
    import math
    for t in (1, 2, 3):
        chainer_adam_step(P, G, M, V, t)
        lr_t = 0.001 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        _lib.check(lib.pivp_adam_step(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, lr_t, 0.9, 0.999, 1e-8, 1.0, _st()), 'adam')
    torch.cuda.synchronize()
    assert np.abs(pd.cpu().numpy() - P['w']).max() < 2e-6
    assert _rel(md.cpu().numpy(), M['w']) < 1e-5 and _rel(vd.cpu().numpy(), V['w']) < 1e-5


def test_adam_epsilon_placement_kat(env):
    """Chainer 2's AdamRule adds eps to the UNcorrected sqrt(v): p -= alpha * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps).
    PyTorch / the paper divide the corrected moments: p -= alpha * mhat / (sqrt(vhat) + eps).  They differ where sqrt(v) ~ eps, i.e. for
    |g| ~ eps / sqrt(1 - b2) = 3e-7 at t = 1: there Chainer's first step is alpha / 2, the other placement's 0.97 alpha.  Gradients that
    small pin the placement; the 0.01-sized gradients of test_adam_step_matches_chainer_rule cannot."""
    pivp, _lib, lib = env
    import math
    g = np.array([1e-9, 1e-8, 1e-7, 3.1623e-7, 1e-6, 1e-5, 1e-3, -3.1623e-7, -1e-8, 0.0], dtype=np.float64)
    alpha, b1, b2, eps = 0.001, 0.9, 0.999, 1e-8
    m = (1 - b1) * g; v = (1 - b2) * g * g
    lr_t = alpha * math.sqrt(1 - b2) / (1 - b1)
    chainer = -lr_t * m / (np.sqrt(v) + eps)                       # AdamRule.update_core at t = 1 from zero moments
    paper = -alpha * g / (np.abs(g) + eps)                         # mhat = g, vhat = g^2
    assert abs(chainer[3] / -alpha - 0.5) < 1e-3 and abs(paper[3] / -alpha - 0.97) < 1e-2   # the two rules really differ here
    pd = _t(np.zeros_like(g)); gd = _t(g); md = torch.zeros_like(pd); vd = torch.zeros_like(pd)
    _lib.check(lib.pivp_adam_step(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), g.size, lr_t, b1, b2, eps, 1.0, _st()), 'adam')
    torch.cuda.synchronize()
    got = pd.cpu().numpy().astype(np.float64)
    assert np.abs(got - chainer).max() < 2e-9, (got, chainer)
    assert np.abs(got - paper).max() > 4e-4                        # and is NOT the other placement


@pytest.mark.parametrize('mode,B,cin,cout,H,repeats', [(0, 3, 32, 32, 32, 2), (0, 5, 64, 64, 16, 3), (1, 3, 128, 128, 8, 2), (1, 5, 96, 96, 16, 3),
                                                       (1, 1, 64, 64, 32, 1), (0, 1, 32, 32, 64, 2), (1, 7, 64, 64, 8, 2)])
def test_conv_weight_gradient_through_partial_planes(env, mode, B, cin, cout, H, repeats):
    """The enc convs' weight gradients as the BPTT sweep runs them (round 2): per-block partial planes accumulated over `repeats` launches
    with plain loads and stores, ONE reduction into dW, bias gradient summed by the tap blocks that see every dY element once (conv:
    tap 0; transposed conv: taps (1,1), (1,2), (2,1), (2,2)).  Odd batch sizes put tile tails into the pixel splits; exact to fp32."""
    pivp, _lib, lib = env
    rs = np.random.RandomState(cin + mode + B)
    x = rs.randn(B, cin, H, H)
    Ho = 2 * H if mode else H // 2
    dy = rs.randn(B, cout, Ho, Ho)
    if mode:
        W = rs.randn(cin, cout, 3, 3) / np.sqrt(9 * cin); key = 'enc4/W'
    else:
        W = rs.randn(cout, cin, 3, 3) / np.sqrt(9 * cin); key = 'enc1/W'
    tx = torch.tensor(x, dtype=torch.float64); tW = torch.tensor(W, dtype=torch.float64, requires_grad=True)
    tb = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    y = F.conv_transpose2d(tx, tW, tb, stride=2, padding=1, output_padding=1) if mode else F.conv2d(tx, tW, tb, stride=2, padding=1)
    (y * torch.tensor(dy)).sum().backward()
    xd, dyd = _nhwc(x), _nhwc(dy)
    n = lib.pivp_conv_backward_part_floats(mode, cin, cout, B, H, H)
    assert n > 0
    part = torch.zeros(n, dtype=torch.float32, device=DEV)
    prior = rs.randn(W.size).astype(np.float32) * 0.01                      # dW is accumulated into, not overwritten
    dW = _t(prior); db = torch.zeros(cout, dtype=torch.float32, device=DEV)
    _lib.check(lib.pivp_conv_wgrad_partial(mode, xd.data_ptr(), cin, cin, dyd.data_ptr(), cout, cout, part.data_ptr(), dW.data_ptr(),
                                           db.data_ptr(), B, H, H, repeats, _st()), 'conv_wgrad_partial')
    torch.cuda.synchronize()
    got = pivp.from_internal(key, dW.cpu().numpy() - prior, W.shape)
    assert _rel(got, repeats * tW.grad.numpy()) < 2e-5
    assert _rel(db.cpu().numpy(), repeats * tb.grad.numpy()) < 2e-5


@pytest.mark.parametrize('mode,B,cin,cout,H,T', [(0, 3, 32, 32, 32, 3), (0, 5, 64, 64, 16, 4), (1, 3, 128, 128, 8, 3), (1, 5, 96, 96, 16, 2), (1, 2, 64, 64, 32, 3),
                                                 (0, 1, 32, 32, 64, 2), (1, 7, 64, 64, 8, 5), (1, 2, 64, 64, 64, 2), (1, 2, 32, 32, 16, 2), (0, 2, 32, 32, 24, 2)])
def test_conv_weight_gradient_batched_over_timesteps(env, mode, B, cin, cout, H, T):
    """The sweep's form since round 5: ONE launch takes a batch of timesteps (x walked backwards through the forward slabs: a negative step; dY forwards
    through its ring), the first launch of a sweep stores into the partial planes (no zeroing: they start as garbage here), a second launch adds, one
    reduction.  The model's shapes run all nine taps from one staging (wgrad3x3s2.hip); 32 -> 32 transposed and the 12 x 12 anchor map fall to the
    per-tap kernel.  dW = sum over timesteps, against float64 autograd."""
    pivp, _lib, lib = env
    rs = np.random.RandomState(cin + mode + B + T)
    Ho = 2 * H if mode else H // 2
    x = rs.randn(T, B, cin, H, H); dy = rs.randn(T, B, cout, Ho, Ho)
    if mode:
        W = rs.randn(cin, cout, 3, 3) / np.sqrt(9 * cin); key = 'enc4/W'
    else:
        W = rs.randn(cout, cin, 3, 3) / np.sqrt(9 * cin); key = 'enc1/W'
    tW = torch.tensor(W, dtype=torch.float64, requires_grad=True); tb = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    tot = 0
    for t in range(T):
        tx = torch.tensor(x[t], dtype=torch.float64)
        y = F.conv_transpose2d(tx, tW, tb, stride=2, padding=1, output_padding=1) if mode else F.conv2d(tx, tW, tb, stride=2, padding=1)
        tot = tot + (y * torch.tensor(dy[t])).sum()
    tot.backward()
    # forward slabs hold x[t] at slab t; the sweep runs t = T-1 .. 0 and its dY ring slot j holds timestep T-1-j
    xd = _t(np.ascontiguousarray(x.transpose(0, 1, 3, 4, 2)))
    dyd = _t(np.ascontiguousarray(dy[::-1].transpose(0, 1, 3, 4, 2)))
    n = lib.pivp_conv_backward_part_floats(mode, cin, cout, B, H, H)
    assert n > 0
    part = torch.full((n,), float('nan'), dtype=torch.float32, device=DEV)
    prior = rs.randn(W.size).astype(np.float32) * 0.01
    dW = _t(prior); db = torch.zeros(cout, dtype=torch.float32, device=DEV)
    xs, ys = xd[0].numel() * 4, dyd[0].numel() * 4
    first = max(1, T - 1)          # a batch of T - 1 timesteps (stored), then the last one alone (added)
    _lib.check(lib.pivp_conv_wgrad_partial_batch(mode, xd.data_ptr() + (T - 1) * xs, cin, cin, -xs, dyd.data_ptr(), cout, cout, ys, first, 1,
                                                 part.data_ptr(), dW.data_ptr(), db.data_ptr(), B, H, H, _st()), 'batch 1')
    if first < T:
        _lib.check(lib.pivp_conv_wgrad_partial_batch(mode, xd.data_ptr(), cin, cin, -xs, dyd.data_ptr() + (T - 1) * ys, cout, cout, ys, 1, 0,
                                                     part.data_ptr(), dW.data_ptr(), db.data_ptr(), B, H, H, _st()), 'batch 2')
    _lib.check(lib.pivp_conv_wgrad_partial_reduce(mode, cin, cout, part.data_ptr(), dW.data_ptr(), db.data_ptr(), B, H, H, _st()), 'reduce')
    torch.cuda.synchronize()
    got = pivp.from_internal(key, dW.cpu().numpy() - prior, W.shape)
    assert _rel(got, tW.grad.numpy()) < 2e-5
    assert _rel(db.cpu().numpy(), tb.grad.numpy()) < 2e-5


@pytest.mark.parametrize('B,cx,C,H', [(3, 32, 32, 32), (5, 32, 64, 16), (3, 64, 128, 8), (1, 96, 32, 32), (2, 128, 64, 16)])
def test_convlstm_backward_last_timestep_computes_dx_only(env, B, cx, C, H):
    """t = 0 of the sweep: the data gradient runs on the first cx columns of the transposed weight pack (IgemmDesc::wN); d x, d c, dW, db
    are those of the full backward and the d h_{-1} columns of d_in are NOT computed (they keep their content, or are zero where the gate
    kernel cleared d_in for a K-split data gradient).  Odd batches: M tails of the column-limited tiles."""
    pivp, _lib, lib = env
    rs = np.random.RandomState(C + cx + B)
    x = rs.randn(B, cx, H, H); h = rs.randn(B, C, H, H) * 0.5; c = rs.randn(B, C, H, H)
    W = rs.randn(4 * C, cx + C, 5, 5) / np.sqrt(25 * (cx + C)); b = rs.randn(4 * C) * 0.1
    dh = rs.randn(B, C, H, H); dcn = rs.randn(B, C, H, H)
    tx, th, tc = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in (x, h, c)]
    tW = torch.tensor(W, dtype=torch.float64, requires_grad=True); tb = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    g = F.conv2d(torch.cat((tx, th), 1), tW, tb, padding=2)
    j, i, f, o = torch.split(g, C, dim=1)
    cn = tc * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
    hn = torch.tanh(cn) * torch.sigmoid(o)
    (hn * torch.tensor(dh) + cn * torch.tensor(dcn)).sum().backward()
    xd, hd, cd = _nhwc(x), _nhwc(h), _nhwc(c)
    wd, bd = _t(pivp.to_internal('lstm1/conv/W', W)), _t(b)
    c_out = torch.empty_like(cd); h_out = torch.empty_like(hd)
    M = B * H * H
    gates = torch.empty((M, 4 * C), dtype=torch.float32, device=DEV)
    _lib.check(lib.pivp_convlstm_train(xd.data_ptr(), cx, cx, hd.data_ptr(), C, wd.data_ptr(), bd.data_ptr(), cd.data_ptr(), c_out.data_ptr(),
                                       h_out.data_ptr(), gates.data_ptr(), B, H, H, _st()), 'fwd')
    dha = _nhwc(dh); dc = _nhwc(dcn).clone()
    dG = torch.empty((M, 4 * C), dtype=torch.float32, device=DEV)
    wt = torch.empty_like(wd)
    d_in = torch.full((M, cx + C), 7.0, dtype=torch.float32, device=DEV)
    dW = torch.zeros_like(wd); db = torch.zeros_like(bd)
    _lib.check(lib.pivp_convlstm_backward_dx_only(xd.data_ptr(), cx, cx, hd.data_ptr(), C, wd.data_ptr(), gates.data_ptr(), cd.data_ptr(),
                                                  c_out.data_ptr(), dha.data_ptr(), C, None, 0, dc.data_ptr(), 1, dG.data_ptr(), wt.data_ptr(),
                                                  d_in.data_ptr(), dW.data_ptr(), db.data_ptr(), B, H, H, _st()), 'bwd')
    torch.cuda.synchronize()
    din = d_in.cpu().numpy().reshape(B, H, H, cx + C).transpose(0, 3, 1, 2)
    assert _rel(din[:, :cx], tx.grad.numpy()) < 2e-5
    assert np.all((din[:, cx:] == 7.0) | (din[:, cx:] == 0.0))     # d h_{-1}: not computed (untouched, or cleared with the rest of d_in)
    assert np.abs(din[:, cx:] - th.grad.numpy()).max() > 1e-3
    assert _rel(dc.cpu().numpy().reshape(B, H, H, C).transpose(0, 3, 1, 2), tc.grad.numpy()) < 2e-5
    assert _rel(pivp.from_internal('lstm1/conv/W', dW.cpu().numpy(), W.shape), tW.grad.numpy()) < 2e-5
    assert _rel(db.cpu().numpy(), tb.grad.numpy()) < 2e-5


@pytest.mark.parametrize('B,C,H,lddy,recurrent', [(32, 32, 32, 64, True), (8, 64, 16, 96, True), (32, 128, 8, 128, True), (5, 32, 32, 32, False),
                                                  (2, 64, 16, 64, True)])
def test_layernorm_plus_gate_backward_pair(env, B, C, H, lddy, recurrent):
    """hidden<k>'s LayerNorm backward + lstm<k>'s gate backward as the sweep runs the pair (TM:203-208, TM:269-272: sums and parameter
    planes, then the gate kernel forming the norm's dx from the partials) against float64 autograd."""
    pivp, _lib, lib = env
    rs = np.random.RandomState(100 * C + B)
    npix, n = H * H, H * H * C
    # forward pieces in NHWC: pre-activation gates, c_{t-1}; h_t = tanh(c_t) s(o) goes through the norm, whose output meets dy
    pre = rs.randn(B, npix, 4, C); c_old = rs.randn(B, npix, C)
    gamma = 1.0 + 0.3 * rs.randn(npix, C); beta = 0.1 * rs.randn(npix, C)
    dy = rs.randn(B, npix, C); dh_b = rs.randn(B, npix, C) * 0.5; dc_in = rs.randn(B, npix, C)
    tp, tco = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in (pre, c_old)]
    tg, tb = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in (gamma, beta)]
    aj, ai, af, ao = torch.tanh(tp[:, :, 0]), torch.sigmoid(tp[:, :, 1]), torch.sigmoid(tp[:, :, 2] + 1.0), torch.sigmoid(tp[:, :, 3])
    cn = tco * af + ai * aj
    hn = torch.tanh(cn) * ao
    mean = hn.reshape(B, -1).mean(1)[:, None, None]; var = hn.reshape(B, -1).var(1, unbiased=False)[:, None, None]
    eps = 1e-6
    y = (hn - mean) / torch.sqrt(var + eps) * tg + tb
    loss = (y * torch.tensor(dy)).sum() + (cn * torch.tensor(dc_in)).sum()
    if recurrent:
        loss = loss + (hn * torch.tensor(dh_b)).sum()
    loss.backward()
    # HIP: stored activations j, i, f, o; stat = (mean, rstd)
    gates = _t(torch.stack((aj, ai, af, ao), 2).detach().numpy().reshape(B * npix, 4 * C))
    stat = _t(np.stack((mean.detach().numpy().reshape(B), 1.0 / np.sqrt(var.detach().numpy().reshape(B) + eps)), 1))
    dyw = torch.full((B * npix, lddy), 7.0, dtype=torch.float32, device=DEV)
    off = lddy - C                                                           # the slice sits at the end of a wider concat row
    dyw[:, off:] = _t(dy.reshape(B * npix, C))
    dc = _t(dc_in.reshape(B * npix, C)); dG = torch.empty((B * npix, 4 * C), dtype=torch.float32, device=DEV)
    dgm = torch.full((n,), 0.5, dtype=torch.float32, device=DEV); dbt = torch.full((n,), -0.25, dtype=torch.float32, device=DEV)   # accumulated into
    scratch = torch.empty(lib.pivp_gates_backward_ln_scratch_floats(B, n), dtype=torch.float32, device=DEV)
    hb = _t(dh_b.reshape(B * npix, C)) if recurrent else None
    cod, cnd, gmd, hd = _t(c_old.reshape(B * npix, C)), _t(cn.detach().numpy().reshape(B * npix, C)), _t(gamma.reshape(-1)), _t(hn.detach().numpy().reshape(B * npix, C))
    _lib.check(lib.pivp_gates_backward_ln(gates.data_ptr(), cod.data_ptr(), cnd.data_ptr(),
                                          dyw.data_ptr() + off * 4, lddy, gmd.data_ptr(), stat.data_ptr(),
                                          hd.data_ptr(), hb.data_ptr() if recurrent else None, C,
                                          dc.data_ptr(), 1, dG.data_ptr(), dgm.data_ptr(), dbt.data_ptr(), scratch.data_ptr(), B, npix, C, _st()),
               'pivp_gates_backward_ln')
    torch.cuda.synchronize()
    ref_dG = tp.grad.numpy().reshape(B * npix, 4 * C)
    assert _rel(dG.cpu().numpy(), ref_dG) < 2e-5
    assert _rel(dc.cpu().numpy(), tco.grad.numpy().reshape(B * npix, C)) < 2e-5
    assert _rel(dgm.cpu().numpy() - 0.5, tg.grad.numpy().reshape(-1)) < 2e-5
    assert _rel(dbt.cpu().numpy() + 0.25, tb.grad.numpy().reshape(-1)) < 2e-5
    assert off == 0 or float(dyw[:, :off].min()) == 7.0


def _lstm_wgrad_ref(x, h, dG):
    """dW[n, ci, ky, kx] = sum_{b,y,x} concat(x, h)[b, ci, y+ky-2, x+kx-2] * dG[b, n, y, x] (zero outside the image), float64 (TM:262-266 backward)."""
    xin = np.concatenate([x, h], 1) if h is not None else x
    H, W = xin.shape[2:]
    pad = np.pad(xin, ((0, 0), (0, 0), (2, 2), (2, 2)))
    dW = np.zeros((dG.shape[1], xin.shape[1], 5, 5))
    for ky in range(5):
        for kx in range(5):
            dW[:, :, ky, kx] = np.einsum('bnyx,bcyx->nc', dG, pad[:, :, ky:ky + H, kx:kx + W])
    return dW


# (B, cx, C, H, timesteps per launch): every map width of the model (8 ... 64) and 128 (config 5), odd batches, cx != C, partitions with one and several
# pixel parts, segments that cross tile boundaries, a batch of timesteps per launch and several launches into the same slots
@pytest.mark.parametrize('B,cx,C,H,Ts', [(2, 32, 32, 32, (1,)), (3, 32, 32, 32, (2, 1)), (2, 32, 64, 16, (1, 1, 1)), (5, 64, 64, 16, (3,)), (3, 64, 128, 8, (1, 2)),
                                         (1, 96, 32, 32, (1,)), (2, 128, 64, 16, (2,)), (2, 32, 32, 64, (1,)), (1, 32, 32, 128, (1,)), (32, 64, 128, 8, (1,)),
                                         (8, 96, 32, 32, (1,))])
@pytest.mark.parametrize('form', [1, 2])      # 32 / 64 gate columns per wave
def test_convlstm_weight_gradient_in_partial_slots(env, B, cx, C, H, Ts, form):
    """Round 6's fp32 ConvLSTM weight gradient (csrc/wgrad5x5p.hip): launches add into NaN-initialised partial slots (the first one stores), one reduction
    forms dW and db.  Against float64 sums; against the round-2 kernel; and bit-identical when repeated."""
    import hip_ops as ops
    rs = np.random.RandomState(B * 7 + cx + C + H)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    launches = []
    ref = 0.0; dbr = 0.0
    for T in Ts:
        xs = [f32(rs.randn(B, cx, H, H)) for _ in range(T)]; hs = [f32(rs.randn(B, C, H, H) * 0.5) for _ in range(T)]
        dGs = [f32(rs.randn(B, 4 * C, H, H) * 0.1) for _ in range(T)]
        launches.append((xs, hs, dGs))
        ref = ref + sum(_lstm_wgrad_ref(x, h, g) for x, h, g in zip(xs, hs, dGs))
        dbr = dbr + sum(g.sum(axis=(0, 2, 3)) for g in dGs)
    got, db, raw = ops.wgrad5x5_f32_batch(launches, form=form)
    unit = np.sqrt((ref ** 2).mean())
    err = np.abs(got - ref).max() / unit
    old, dbo, _ = ops.wgrad5x5_f32_batch(launches, slots=False)
    err_old = np.abs(old - ref).max() / unit
    print('lstm wgrad form %d %d+%d @%d B %d launches %s: slots max |err| %.2e of the gradient rms, round-2 kernel %.2e' % (form, cx, C, H, B, Ts, err, err_old))
    assert np.isfinite(got).all() and err < 2e-5 and err <= 3.0 * err_old + 1e-7
    assert np.abs(db - dbr).max() < 2e-5 * np.abs(dbr).max() + 1e-6
    got2, db2, raw2 = ops.wgrad5x5_f32_batch(launches, form=form)
    assert np.array_equal(raw, raw2) and np.array_equal(db, db2)          # no atomics: bit-identical
    if len(Ts) == 1 and Ts[0] == 1:      # the sweep's t = 0: no h operand, only the x rows are differentiated (their own partition of the slots)
        xs, hs, dGs = launches[0]
        got0, db0, _ = ops.wgrad5x5_f32_batch(launches, h_is_zero=True, form=form)
        assert np.abs(got0[:, :cx] - ref[:, :cx]).max() < 2e-5 * unit and np.all(got0[:, cx:] == 0)
        assert np.abs(db0 - dbr).max() < 2e-5 * np.abs(dbr).max() + 1e-6
