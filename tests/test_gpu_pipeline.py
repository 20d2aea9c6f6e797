"""GPU tests of the rows SURVEY 8(f) marks "next": predict-side resize, training from the reference's on-disk dataset
format with its checkpoint files, optimizer-state round trip, and the predict entry point."""
import os

import numpy as np
import pytest
import torch

from oracle import restatement as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def pivp():
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    import pivp_amd
    return pivp_amd


def _make_dataset(root, n=6, T=4):
    from pivp_amd import dataset as ds
    rs = np.random.RandomState(0)
    rows = []
    for j in range(n):
        np.save(os.path.join(root, 'image_batch_%d' % j), rs.rand(T, 64, 64, 3).astype(np.float32))
        np.save(os.path.join(root, 'action_batch_%d' % j), (rs.randn(T, 5) * 0.1).astype(np.float32))
        np.save(os.path.join(root, 'state_batch_%d' % j), (rs.randn(T, 5) * 0.1).astype(np.float32))
        np.save(os.path.join(root, 'image_batch_pred_%d' % j), (rs.rand(T, 96, 120, 3) * 255).astype(np.uint8))
        rows.append([j, '', 'image_batch_%d.npy' % j, 'action_batch_%d.npy' % j, 'state_batch_%d.npy' % j, '', 'image_batch_pred_%d.npy' % j])
    ds.write_map(root, rows)


def test_resize_images_matches_oracle(pivp):
    from pivp_amd.predict import resize_images
    rs = np.random.RandomState(2)
    x = (rs.rand(2, 3, 512, 640) * 255).astype(np.float32)            # the reference's raw frame size (predict_model.py:71-72)
    got = resize_images(x, (64, 64), scale=1.0 / 255.0).cpu().numpy()
    ref = R.resize_images(x.astype(np.float64), (64, 64)) / 255.0
    assert np.abs(got - ref).max() < 1e-5


def test_train_checkpoint_resume_predict(pivp, tmp_path):
    from pivp_amd import train as T, predict as Pm
    data = tmp_path / 'data'; data.mkdir(); out = tmp_path / 'models'; out.mkdir()
    _make_dataset(str(data))
    np.random.seed(0)
    save_dir = T.main(['--data_dir', str(data), '--output_dir', str(out), '--num_iterations', '6', '--batch_size', '2',
                       '--schedsamp_k', '-1', '--save_interval', '1', '--validation_interval', '1', '--train_val_split', '0.7'])
    files = sorted(os.listdir(save_dir))
    assert 'version' in files and 'training-0' in files and 'state-0' in files and 'training-global_losses.npy' in files
    losses = np.load(os.path.join(save_dir, 'training-global_losses.npy'))
    assert losses.ndim == 2 and losses.shape[1] == 5 and np.isfinite(losses).all()      # mean, std, min, max, median per epoch
    # model checkpoint: the reference's keys and shapes
    last = sorted(f for f in files if f.startswith('training-') and f[9:].isdigit())[-1]
    with np.load(os.path.join(save_dir, last)) as z:
        assert sorted(z.files) == sorted(R.param_shapes())
        assert z['lstm5/conv/W'].shape == (512, 192, 5, 5)
    # optimizer state: t, epoch, per-parameter m / v
    with np.load(os.path.join(save_dir, 'state-' + last[9:])) as z:
        assert int(z['t']) >= 1 and 'epoch' in z.files and 'enc0/W/m' in z.files and 'enc0/W/v' in z.files
        assert z['lstm1/conv/W/m'].shape == (128, 64, 5, 5)
    # resume: model + optimizer state load and training continues from them
    m = pivp.Model(10, prefix='r', keep_activations=True)
    pivp.load_npz(os.path.join(save_dir, last), m)
    imgs, acts, stas = R.synthetic_batch(2, 4)
    with pivp.using_config('train', False):
        m([imgs, acts, stas], 0)
    m.reset_state()
    opt = pivp.Adam().setup(m)
    pivp.load_optimizer_npz(os.path.join(save_dir, 'state-' + last[9:]), opt)
    t0 = opt.t
    l1 = float(opt.update(m, [imgs, acts, stas], 0)); m.reset_state()
    l2 = float(opt.update(m, [imgs, acts, stas], 1))
    assert opt.t == t0 + 2 and np.isfinite(l1) and l2 < l1                      # two Adam steps on one batch lower its loss
    # predict entry point: model type from the directory name (predict_model.py:91-95), one rollout, uint8 frames
    args = Pm.build_parser().parse_args([os.path.basename(save_dir), last, '1', '--models_dir', str(out), '--data_dir', str(data)])
    loss, frames = Pm.predict(args)
    assert frames.shape == (3, 3, 64, 64) and frames.dtype == np.uint8 and np.isfinite(loss)
    assert frames.reshape(3, -1).min(axis=1).tolist() == [0, 0, 0] and frames.reshape(3, -1).max(axis=1).tolist() == [255, 255, 255]


def test_device_feeder_matches_the_synchronous_loop(pivp):
    """dataset.DeviceFeeder hands the model its own reusable device buffers while the copy stream refills the other slot: five
    optimizer steps fed through it give the losses and the parameters of the synchronous loop (concat_examples + blocking upload per step),
    bit for bit in the forward (the loss of step k depends on every earlier batch having been read intact)."""
    from pivp_amd import dataset as ds
    rs = np.random.RandomState(3)
    data = [(rs.rand(4, 64, 64, 3).astype(np.float32), (rs.randn(4, 5) * 0.1).astype(np.float32), (rs.randn(4, 5) * 0.1).astype(np.float32))
            for _ in range(7)]
    P = R.init_params_widened(seed=1, scale=1.0)

    def run(fed):
        m = pivp.Model(10, prefix='f', keep_activations=True)
        m.load_state_dict_reference(P)
        opt = pivp.Adam(alpha=1e-3).setup(m)
        it = ds.SerialIterator(data, 2, repeat=True, shuffle=False)
        feeder = ds.DeviceFeeder(it, device='cuda:0') if fed else None
        losses = []
        for k in range(5):
            if fed:
                x, _, _ = feeder.get()
            else:
                x = list(pivp.concat_examples(it.next()))
            loss = opt.update(m, x, k)
            if fed:
                feeder.prefetch()                  # the refill of the OTHER slot runs under this step
            losses.append(loss.clone())            # no host synchronisation inside the loop: the copy stream really overlaps the steps
            m.reset_state()
        torch.cuda.synchronize()
        return [float(l) for l in losses], m._flat_params.clone()

    l_sync, p_sync = run(False)
    l_fed, p_fed = run(True)
    assert l_fed[0] == l_sync[0]                                   # same first forward, bit for bit
    assert np.allclose(l_fed, l_sync, rtol=1e-5, atol=0)           # later steps: weight-gradient atomics reorder fp32 sums
    # parameters: Adam's step is alpha * m / (sqrt(v) + eps), so an element whose gradient is itself rounding noise can move by up to alpha
    # per step either way between two runs of the SAME loop; the mean over the 9.2 M parameters is what a corrupted batch would move
    assert float((p_fed - p_sync).abs().mean()) < 1e-5
    assert len(set(l_sync)) == 5


def test_bench_two_ranks_rehearsal_on_one_gpu(pivp):
    """`python bench.py --gpus 2` as the driver types it, except that both ranks share this box's one GPU (--share-gpu, gloo): the
    launcher, the rank bookkeeping, both legs with real kernels and the data-parallel step with its overlapped all-reduce."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--share-gpu', '--batch', '4',
                        '--steps', '2', '--warmup', '1', '--no-cpu-baseline'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 8 and 'REHEARSAL' in out['data']
    assert out['value'] > 0 and out['roofline']['frac'] > 0
    tr = out['train']
    assert tr['rccl_ranks'] == 2 and tr['backend'] == 'gloo' and tr['ms_per_step'] > 0 and tr['ms_per_step_without_allreduce'] > 0
    assert np.isfinite(tr['loss']) and np.isfinite(out['config']['loss'])
    assert tr['allreduce_algo'] == 'allreduce' and out['train_bf16']['allreduce_algo'] == 'rs_ag'      # fp32 payload: ring; bf16 payload: all-links, fp32 local sum
    assert set(out['train_bf16']['allreduce_algo_ms_per_step']) == {'allreduce', 'rs_ag'}
