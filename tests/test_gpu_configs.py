"""BASELINE.json configs 3 and 5 at their real per-GPU size on one MI355X.

config 3: 8 x MI355X data parallel, global batch 256 = 32 sequences per GPU, bf16 -> one rank's share: B = 32, T = 10, bf16 train step.
config 5: 128 x 128 x 3 frames, 20-step rollout, 8 x MI355X data parallel     -> one rank's share: B = 32, T = 20, 128 x 128.

The float64 oracle needs minutes at these sizes, so the checks are the committed fixture of config 5's geometry at B = 2
(tests/golden/cdna_128_b2_t20.npz) plus size-independent properties: samples are independent (rows of the B = 32 run equal the
same sequences run as B = 2), a duplicated sample is bit-identical, the loss equals its definition recomputed on the host
(TM:739-758), the bf16 gradient stays within its reported distance of the fp32 one, the workspace is what DESIGN.md 4 says."""
import os

import numpy as np
import pytest
import torch

from oracle import restatement as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def pivp():
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    import pivp_amd
    return pivp_amd


def _host_loss(imgs, stas, gen, gen_states, ctx=2):
    """TM:739-758: sum of frame MSEs + 1e-4 * sum of state MSEs over the predicted steps, / (T - ctx)."""
    T = imgs.shape[0]
    fr = [np.mean((imgs[t + ctx].astype(np.float64) - gen[t + ctx - 1]) ** 2) for t in range(T - ctx)]
    st = [np.mean((stas[t + ctx].astype(np.float64) - gen_states[t + ctx - 1]) ** 2) * 1e-4 for t in range(T - ctx)]
    return (sum(fr) + sum(st)) / float(T - ctx)


def _model(pivp, P, precision='fp32', keep=False, **kw):
    m = pivp.Model(10, prefix='cfg', precision=precision, keep_activations=keep, **kw)
    m.load_state_dict_reference(P)
    return m


def test_config3_bf16_train_step_at_rank_size(pivp):
    """One rank of config 3: B = 32, T = 10, 64 x 64, bf16 gate convolutions, forward + BPTT backward + Adam."""
    P = R.init_params(seed=1, dtype=np.float32, scale=1.0)
    imgs, acts, stas = R.synthetic_batch(32, 10)
    imgs[:, 9] = imgs[:, 4]; acts[:, 9] = acts[:, 4]; stas[:, 9] = stas[:, 4]
    grads, losses = {}, {}
    for prec in ('fp32', 'bf16'):
        m = _model(pivp, P, prec, keep=True)
        opt = pivp.Adam(alpha=0.001).setup(m)
        with pivp.using_config('train', True):       # schedsamp_k = -1: feed-self in training mode too (TM:649-657)
            loss = float(m([imgs, acts, stas], 0))
            m.cleargrads(); m.backward()
        gen = torch.stack(m.gen_images).cpu().numpy(); gs = torch.stack(m.gen_states).cpu().numpy()
        assert np.isfinite(gen).all() and np.isfinite(loss)
        if prec == 'fp32':
            assert np.array_equal(gen[:, 9], gen[:, 4])                   # duplicated sample: bit-identical frames
        else:   # the bf16 kernel's summation order depends on the sample's place in its tile (deterministic run to run:
            #     scripts/debug_duplicate_sample.py); a last-bit difference then flips bf16 operand roundings downstream
            assert R.per_pixel_l2(gen[:, 9], gen[:, 4]).max() < 2e-2
        assert abs(loss - _host_loss(imgs, stas, gen, gs)) < 1e-6
        g = m._flat_grads.clone()
        assert bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
        grads[prec], losses[prec] = g, loss
        if prec == 'bf16':
            assert m._active.lib.pivp_plan_get_precision(m._active.h) == 1
            # samples are independent: the same two sequences as a B = 2 batch (other tiles, same bf16 operands)
            m2 = _model(pivp, P, 'bf16')
            with pivp.using_config('train', False):
                m2([imgs[:, 2:4], acts[:, 2:4], stas[:, 2:4]], 0)
            l2 = R.per_pixel_l2(gen[:, 2:4], torch.stack(m2.gen_images).cpu().numpy())
            print('config 3: B=32 rows vs the same sequences as B=2, bf16: max per-pixel L2 %.2e rms %.2e' % (l2.max(), np.sqrt((l2 ** 2).mean())))
            assert l2.max() < 2e-2 and np.sqrt((l2 ** 2).mean()) < 2e-3
            before = m._flat_params.clone()
            opt.step(m)                                                    # Chainer-rule Adam on the flat buffer
            moved = (m._flat_params - before).abs()
            assert bool(torch.isfinite(m._flat_params).all()) and 0 < float(moved.max()) <= 1.001e-3   # |step| <= alpha at t = 1
    rel = float((grads['bf16'] - grads['fp32']).norm() / grads['fp32'].norm())
    print('config 3: bf16 vs fp32 at B=32, T=10: loss %.6f / %.6f, relative gradient difference %.2e' % (losses['bf16'], losses['fp32'], rel))
    assert abs(losses['bf16'] - losses['fp32']) < 1e-3
    assert 1e-6 < rel < 5e-2                                               # reported 2.2e-2


def test_config5_geometry_matches_golden(pivp):
    """128 x 128 frames, 20-frame sequences (19 predicted, 17 fed back) against the float64 oracle's fixture at B = 2."""
    g = np.load(os.path.join(GOLD, 'cdna_128_b2_t20.npz'))
    P = R.init_params(seed=1, dtype=np.float32, scale=1.0, height=128, width=128)
    imgs, acts, stas = R.synthetic_batch(2, 20, 128, 128)
    m = _model(pivp, P)
    with pivp.using_config('train', False):
        loss = float(m([imgs, acts, stas], 0))
    gen = torch.stack(m.gen_images).cpu().numpy()
    pix = np.ascontiguousarray(gen.transpose(0, 1, 3, 4, 2)).reshape(-1, 3)[::int(g['pixel_stride'])]
    l2 = np.sqrt(((pix.astype(np.float64) - g['gen_pixels']) ** 2).sum(axis=1))
    ref32 = g['fp32_oracle_pixels_l2'].astype(np.float64)
    n = (l2.size // 19) * 19                                               # flat order is step-major
    per_step = lambda v, f: f(v[:n].reshape(19, -1), axis=1)
    mx, mx32 = per_step(l2, np.max), per_step(ref32, np.max)
    rms, rms32 = np.sqrt(per_step(l2 ** 2, np.mean)), np.sqrt(per_step(ref32 ** 2, np.mean))
    print('config 5 geometry (B=2), per-step max per-pixel L2: HIP', ['%.1e' % v for v in mx])
    print('                                   float32 oracle itself', ['%.1e' % v for v in mx32])
    print('loss %.8f vs %.8f' % (loss, float(g['loss'])))
    # The north star's bound is stated for 10-step rollouts; a 20-frame sequence feeds 17 predictions back, and with random weights
    # a rounding error grows ~1.5x per fed-back step: plain float32 (NumPy, the reference's arithmetic) is itself 1.2e-4 from the
    # float64 result at step 14 and 2.8e-3 at step 18 on this fixture.  So: 1e-4 over the first 9 predicted frames (a 10-step
    # rollout), and beyond them no worse than 1.5 x plain float32 wherever that is past the gate.
    assert mx[:9].max() < 1e-4
    assert (mx < np.maximum(1e-4, 1.5 * mx32)).all()
    assert (rms < np.maximum(2e-6, 1.5 * rms32)).all()
    assert abs(loss - float(g['loss'])) < 1e-5
    assert np.abs(torch.stack(m.gen_states).cpu().numpy() - g['gen_states']).max() < 1e-5
    assert np.abs(gen.mean(axis=(2, 3, 4), dtype=np.float64) - g['frame_mean']).max() < 1e-6
    assert m.count_params() == 18059519


def test_config5_rollout_and_train_step_at_rank_size(pivp):
    """One rank of config 5: B = 32, T = 20, 128 x 128: rollout, then one train step with every activation kept for BPTT."""
    P = R.init_params(seed=1, dtype=np.float32, scale=1.0, height=128, width=128)
    imgs, acts, stas = R.synthetic_batch(32, 20, 128, 128)
    imgs[:, 31] = imgs[:, 5]; acts[:, 31] = acts[:, 5]; stas[:, 31] = stas[:, 5]
    m = _model(pivp, P)
    with pivp.using_config('train', False):
        loss = float(m([imgs, acts, stas], 0))
    gen = torch.stack(m.gen_images).cpu().numpy(); gs = torch.stack(m.gen_states).cpu().numpy()
    assert gen.shape == (19, 32, 3, 128, 128) and np.isfinite(gen).all()
    assert np.array_equal(gen[:, 31], gen[:, 5])
    assert abs(loss - _host_loss(imgs, stas, gen, gs)) < 1e-6
    m2 = _model(pivp, P)
    with pivp.using_config('train', False):
        m2([imgs[:, 6:8], acts[:, 6:8], stas[:, 6:8]], 0)
    l2 = R.per_pixel_l2(gen[:, 6:8], torch.stack(m2.gen_images).cpu().numpy())
    print('config 5: B=32 rows vs the same sequences as B=2: max per-pixel L2 per step', ['%.1e' % v for v in l2.max(axis=(1, 2, 3))])
    assert l2[:9].max() < 5e-5           # other tiles, other summation order; both within 1e-4 of the oracle over a 10-step rollout
    assert l2.max() < 5e-3               # 17 fed-back steps amplify a last-bit difference ~1.5x each (test_config5_geometry_matches_golden)
    ws_infer = m._active.lib.pivp_plan_workspace_bytes(m._active.h)
    del m, m2, gen
    torch.cuda.empty_cache()

    mt = _model(pivp, P, keep=True)
    opt = pivp.Adam(alpha=0.001).setup(mt)
    with pivp.using_config('train', True):
        tloss = float(opt.update(mt, [imgs, acts, stas], 0))
    ws_train = mt._active.lib.pivp_plan_workspace_bytes(mt._active.h)
    print('config 5: workspace %.2f GB rollout, %.2f GB with the activations of 19 steps kept; train loss %.6f' % (ws_infer / 2**30, ws_train / 2**30, tloss))
    assert abs(tloss - loss) < 1e-5                                        # same forward arithmetic with the slabs kept
    g = mt._flat_grads
    assert g.numel() >= 18059519 and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
    assert bool(torch.isfinite(mt._flat_params).all())
    # 19 kept per-step slabs (activations, gate activations, LayerNorm statistics, head tensors) of ~1 GB each at
    # B = 32, 128 x 128, plus gradients-of-activations scratch: 18.4 GiB measured.  288 GB of HBM holds it 15 times over.
    assert 10 * 2**30 < ws_train < 24 * 2**30
    assert ws_infer < 3 * 2**30


def test_config2_and_3_training_trajectories(pivp):
    """Thirty optimizer.update steps (TM:950) on one fixed video-like batch at the per-GPU size of configs 2 and 3 (B = 32, T = 10): the
    loss falls steadily in fp32 and in the bf16 mode, and the two trajectories stay together -- the train step as the bench times it
    (side stream, K-split data gradients, partial-sum planes, fused Adam) does what an optimizer step is for."""
    P = R.init_params(seed=1, dtype=np.float32, scale=1.0)
    imgs, acts, stas = R.smooth_batch(32, 10)
    traj = {}
    for prec in ('fp32', 'bf16'):
        m = _model(pivp, P, precision=prec, keep=True)
        opt = pivp.Adam(alpha=0.001); opt.setup(m)
        losses = []
        for it in range(30):
            losses.append(float(opt.update(m, [imgs, acts, stas], it)))
            m.reset_state()
        traj[prec] = np.array(losses)
        assert np.isfinite(traj[prec]).all()
        assert traj[prec][-1] < 0.5 * traj[prec][0], (prec, traj[prec][0], traj[prec][-1])          # it learns the batch
        assert (np.diff(traj[prec]) < 0.05 * traj[prec][:-1]).all(), (prec, traj[prec])           # no step blows the loss up
    print('loss, 30 steps: fp32 %.5f -> %.5f, bf16 %.5f -> %.5f' % (traj['fp32'][0], traj['fp32'][-1], traj['bf16'][0], traj['bf16'][-1]))
    assert abs(traj['bf16'][0] - traj['fp32'][0]) < 0.02 * traj['fp32'][0]
    assert np.abs(traj['bf16'] - traj['fp32']).max() < 0.15 * traj['fp32'][0]
