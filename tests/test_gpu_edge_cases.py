"""Edges of the hot path's input domain through the reference's Model surface: non-square frames, odd and ragged batch sizes,
the shortest sequence, other context lengths, the ends of the sampling schedule, and the error behaviour at the boundary
(TM:484-542 constructor, TM:620-657 call, include/pivp_hip.h status codes).  Oracle = the float64 restatement, run live."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import restatement as R

pytestmark = pytest.mark.gpu
GATE = 1e-4


@pytest.fixture(scope='module')
def pivp():
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    import pivp_amd
    return pivp_amd


def _both(pivp, B, T, H=64, W=64, mt='CDNA', nm=10, ctx=2, seed=1, **kw):
    P = R.init_params(seed=seed, dtype=np.float32, scale=1.0, num_masks=nm, model_type=mt, height=H, width=W)
    imgs, acts, stas = R.synthetic_batch(B, T, H, W)
    kinds = dict(is_cdna=mt == 'CDNA', is_stp=mt == 'STP', is_dna=mt == 'DNA')
    ref = R.Model(nm, params=P, dtype=np.float64, prefix='x', num_frame_before_prediction=ctx, **kinds); ref.train = False
    ref_loss = ref([imgs, acts, stas], 0)
    m = pivp.Model(nm, prefix='x', num_frame_before_prediction=ctx, **kinds, **kw)
    m.load_state_dict_reference(P)
    with pivp.using_config('train', False):
        loss = float(m([imgs, acts, stas], 0))
    gen = torch.stack(m.gen_images).cpu().numpy()
    return m, loss, gen, ref, float(ref_loss)


@pytest.mark.parametrize('H,W', [(32, 96), (72, 40), (16, 16), (64, 128)])
def test_non_square_and_small_frames(pivp, H, W):
    # the reference hard-codes 64 x 64 through its deconv outsize (TM:505-507); the generalisation is outsize = 2 x in, any H, W % 8 == 0
    m, loss, gen, ref, ref_loss = _both(pivp, 2, 3, H, W)
    assert gen.shape == (2, 2, 3, H, W)
    assert R.per_pixel_l2(gen, np.stack(ref.gen_images)).max() < GATE
    assert abs(loss - ref_loss) < 1e-5


@pytest.mark.parametrize('B', [1, 3, 5, 7, 33])
def test_odd_and_ragged_batch_sizes(pivp, B):
    # tiles hold 32 / 64 / 128 anchors: batch sizes that leave partial tiles on every map, and one sample more than config 2
    if B <= 7:
        m, loss, gen, ref, ref_loss = _both(pivp, B, 3)
        assert R.per_pixel_l2(gen, np.stack(ref.gen_images)).max() < GATE
        assert abs(loss - ref_loss) < 1e-5
    else:                      # the oracle would need minutes: rows of the big batch against the same sequences as small batches
        P = R.init_params(seed=1, dtype=np.float32, scale=1.0)
        imgs, acts, stas = R.synthetic_batch(B, 4)
        m = pivp.Model(10, prefix='x'); m.load_state_dict_reference(P)
        with pivp.using_config('train', False):
            m([imgs, acts, stas], 0)
        gen = torch.stack(m.gen_images).cpu().numpy()
        for sl in (slice(0, 2), slice(B - 2, B)):
            m2 = pivp.Model(10, prefix='x'); m2.load_state_dict_reference(P)
            with pivp.using_config('train', False):
                m2([imgs[:, sl], acts[:, sl], stas[:, sl]], 0)
            assert R.per_pixel_l2(gen[:, sl], torch.stack(m2.gen_images).cpu().numpy()).max() < 2e-5


def test_odd_batch_gradients(pivp):
    # B = 3 through the whole backward (partial tiles in the data / weight gradient kernels, K-split choices of small M)
    from oracle.torch_restatement import TorchModel
    P = R.init_params_widened(seed=1, scale=1.0)
    imgs, acts, stas = R.synthetic_batch(3, 4)
    tm = TorchModel(10, params=P, requires_grad=True)
    lt = tm([imgs, acts, stas], 0); lt.backward()
    m = pivp.Model(10, prefix='x', keep_activations=True); m.load_state_dict_reference(P)
    loss = float(m([imgs, acts, stas], 0)); m.cleargrads(); m.backward()
    assert abs(loss - float(lt)) < 1e-6
    got = m.grads_reference()
    for k, v in tm.p.items():
        g = v.grad.numpy()
        rel = np.linalg.norm(got[k] - g) / (np.linalg.norm(g) + 1e-30)
        assert rel < 2e-3, (k, rel)


@pytest.mark.parametrize('ctx,T', [(1, 2), (1, 4), (3, 5), (2, 3)])
def test_context_lengths_and_shortest_sequences(pivp, ctx, T):
    # num_frame_before_prediction (TM:484): frames fed as ground truth; T = ctx + 1 is the shortest sequence with a loss term (TM:739, TM:758)
    m, loss, gen, ref, ref_loss = _both(pivp, 2, T, ctx=ctx)
    assert gen.shape[0] == T - 1
    assert R.per_pixel_l2(gen, np.stack(ref.gen_images)).max() < GATE
    assert abs(loss - ref_loss) < 1e-5
    assert len(m.summaries) == 3 * (T - ctx) + 2


def test_schedule_extremes(pivp):
    # TM:654-656: num_ground_truth = round(B * k / (k + exp(iter / k))): iter = 0 -> all ground truth; iter >> k -> none (= feed-self)
    P = R.init_params(seed=1, dtype=np.float32, scale=1.0)
    imgs, acts, stas = R.synthetic_batch(4, 5)

    def run(k, it, train):
        m = pivp.Model(10, prefix='x', scheduled_sampling_k=k); m.load_state_dict_reference(P)
        np.random.seed(5)
        with pivp.using_config('train', train):
            m([imgs, acts, stas], it)
        return torch.stack(m.gen_images).cpu().numpy()
    feedself = run(-1, 0, True)
    none_gt = run(10.0, 1e4, True)          # exp(1000) overflows to inf exactly as in the reference: count = 0
    assert np.array_equal(none_gt, feedself)
    all_gt = run(900.0, 0.0, True)          # 4 * 900 / 901 rounds to 4: every sample gets the ground-truth frame
    ref = R.Model(10, params=P, dtype=np.float64, prefix='x', scheduled_sampling_k=900.0)
    np.random.seed(5)
    ref([imgs, acts, stas], 0.0)
    assert R.per_pixel_l2(all_gt, np.stack(ref.gen_images)).max() < GATE
    assert not np.array_equal(all_gt[2:], feedself[2:])


def test_boundary_errors(pivp):
    imgs, acts, stas = R.synthetic_batch(2, 3)
    with pytest.raises(ValueError, match='No network specified'):              # TM:540
        pivp.Model(10, is_cdna=False, is_dna=False, is_stp=False)
    m = pivp.Model(10, prefix='x')
    with pytest.raises(TypeError):                                             # TM:646 `states[0]` on a 1-element input list
        m([imgs])
    with pytest.raises(ValueError):                                            # not (T, B, 3, H, W)
        m([imgs[:, :, :2], acts, stas])
    with pytest.raises(ValueError):                                            # actions must end in 5
        m([imgs, acts[..., :4], stas])
    with pytest.raises(RuntimeError):                                          # backward needs the activations
        m2 = pivp.Model(10, prefix='x'); m2([imgs, acts, stas], 0); m2.backward()
    with pytest.raises(Exception):                                             # frame sizes must be multiples of 8 (three stride-2 levels)
        pivp.Model(10, prefix='x')([R.synthetic_batch(1, 3, 60, 60)[0], acts[:, :1], stas[:, :1]], 0)
    dna = pivp.Model(2, is_cdna=False, is_dna=True, prefix='x')                # TM:389-390: DNA supports one mask only
    with pytest.raises(Exception):
        dna([imgs, acts, stas], 0)


def test_c_abi_status_codes(pivp):
    from pivp_amd import _lib
    lib = _lib.load()
    cfg = _lib.PivpConfig(batch=2, seq_len=3, height=64, width=64, num_masks=10, model_type=0, use_state=1, context_frames=2,
                          keep_activations=0, ln_eps=1e-6, stp_zero_border=0)
    plan = ctypes.c_void_p()
    assert lib.pivp_plan_create(ctypes.byref(cfg), ctypes.byref(plan)) == 0
    try:
        buf = torch.zeros(16, device='cuda')
        ptr = ctypes.c_void_p(buf.data_ptr())
        # no workspace / parameters bound yet: a STATE error, not a crash
        assert lib.pivp_rollout_forward(plan, ptr, ptr, ptr, None, ptr, ptr, ptr, None) == -3
        assert lib.pivp_rollout_forward(plan, None, ptr, ptr, None, ptr, ptr, ptr, None) == -1      # null pointer: BADARG
        assert lib.pivp_rollout_backward(plan, ptr, ptr, ptr, None, ptr, ptr, None) == -3           # inference plan: no gradients
        assert lib.pivp_plan_set_precision(plan, 7) != 0
    finally:
        lib.pivp_plan_destroy(plan)
    for bad in (dict(height=60), dict(batch=0), dict(seq_len=1), dict(num_masks=12), dict(context_frames=3), dict(model_type=5)):
        kw = dict(batch=2, seq_len=3, height=64, width=64, num_masks=10, model_type=0, use_state=1, context_frames=2,
                  keep_activations=0, ln_eps=1e-6, stp_zero_border=0)
        kw.update(bad)
        c2 = _lib.PivpConfig(**kw)
        p2 = ctypes.c_void_p()
        assert lib.pivp_plan_create(ctypes.byref(c2), ctypes.byref(p2)) == -1, bad
