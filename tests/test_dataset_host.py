"""CPU tests of the host side of the "next" rows: on-disk dataset format, iterator semantics, predict-side
post-processing, resize oracle, CLI flag surface."""
import csv
import os

import numpy as np
import pytest

import pivp_amd
from pivp_amd import dataset as ds
from oracle import restatement as R


def _make_dataset(root, n=7, T=4, H=8, W=8):
    rs = np.random.RandomState(0)
    rows = []
    for j in range(n):
        np.save(os.path.join(root, 'image_batch_%d' % j), rs.rand(T, H, W, 3).astype(np.float32))
        np.save(os.path.join(root, 'action_batch_%d' % j), rs.randn(T, 5).astype(np.float32))
        np.save(os.path.join(root, 'state_batch_%d' % j), rs.randn(T, 5).astype(np.float32))
        np.save(os.path.join(root, 'image_batch_pred_%d' % j), (rs.rand(T, 16, 20, 3) * 255).astype(np.uint8))
        rows.append([j, '', 'image_batch_%d.npy' % j, 'action_batch_%d.npy' % j, 'state_batch_%d.npy' % j, '', 'image_batch_pred_%d.npy' % j])
    ds.write_map(root, rows)
    return rows


def test_map_csv_format_and_loader(tmp_path):
    root = str(tmp_path)
    _make_dataset(root)
    with open(os.path.join(root, 'map.csv')) as f:
        first = f.readline().strip()
    # header and quoting exactly as make_dataset.py:153-156 writes them
    assert first == '"id","img_bitmap_path","img_np_path","action_np_path","state_np_path","img_bitmap_pred_path","img_np_pred_path"'
    images, actions, states = ds.load_dataset(root)
    assert images.shape == (7, 4, 8, 8, 3) and actions.shape == (7, 4, 5) and states.shape == (7, 4, 5)
    assert images.dtype == np.float32
    assert np.array_equal(images[3], np.load(os.path.join(root, 'image_batch_3.npy')))
    (ti, ta, ts), (vi, va, vs) = ds.split_train_val(images, actions, states, 0.95)
    assert len(ti) == int(np.floor(0.95 * 7)) and len(vi) == 7 - len(ti)        # by index, no shuffle (TM:836-843)
    image, image_pred, bmp, action, state = ds.get_data_info(root, 2)
    assert image_pred.shape == (4, 16, 20, 3) and np.array_equal(action, actions[2])
    with pytest.raises(ValueError):
        ds.get_data_info(root, 7)
    empty = tmp_path / 'empty'; empty.mkdir()
    ds.write_map(str(empty), [])
    with pytest.raises(ValueError, match='No file map found'):
        ds.read_map(str(empty))


def test_serial_iterator_matches_chainer_semantics():
    data = list(range(10))
    np.random.seed(3)
    it = ds.SerialIterator(data, 4, repeat=True, shuffle=True)
    np.random.seed(3)
    order = np.random.permutation(10)
    st = np.random.get_state()                                 # RNG state right after the iterator's own permutation draw
    b1 = it.next(); assert b1 == [data[i] for i in order[0:4]] and not it.is_new_epoch and it.epoch == 0
    b2 = it.next(); assert b2 == [data[i] for i in order[4:8]]
    tail = [data[i] for i in order[8:10]]
    b3 = it.next()
    np.random.set_state(st)
    np.random.shuffle(order)                                   # the epoch boundary reshuffles in place with the next draw
    assert b3 == tail + [data[i] for i in order[:2]] and it.is_new_epoch and it.epoch == 1 and it.current_position == 2
    # every element appears exactly once per epoch
    it2 = ds.SerialIterator(data, 5, repeat=True, shuffle=True)
    seen = it2.next() + it2.next()
    assert sorted(seen) == data and it2.is_new_epoch
    # repeat=False stops after one epoch; reset() starts over
    it3 = ds.SerialIterator(data, 4, repeat=False, shuffle=True)
    assert sum(len(b) for b in it3) == 10
    it3.reset()
    assert len(it3.next()) == 4


def test_resize_oracle_and_rescale():
    rs = np.random.RandomState(1)
    x = rs.rand(2, 3, 16, 20)
    y = R.resize_images(x, (8, 8))
    assert y.shape == (2, 3, 8, 8)
    assert np.allclose(y[..., 0, 0], x[..., 0, 0]) and np.allclose(y[..., -1, -1], x[..., -1, -1])   # align-corners
    assert np.allclose(R.resize_images(x, (16, 20)), x)
    import torch
    t = torch.nn.functional.interpolate(torch.tensor(x), size=(8, 8), mode='bilinear', align_corners=True).numpy()
    assert np.abs(t - y).max() < 1e-12
    from pivp_amd.predict import rescale_to_uint8
    f = rs.rand(3, 8, 8).astype(np.float32) * 0.5 + 0.2
    u = rescale_to_uint8(f)
    assert u.dtype == np.uint8 and u.min() == 0 and u.max() == 255


def test_cli_flag_surface_matches_reference():
    from pivp_amd.train import build_parser
    from pivp_amd.predict import build_parser as pred_parser
    a = build_parser().parse_args([])
    ref_defaults = dict(num_iterations=100000, sequence_length=10, context_frames=2, use_state=1, model_type='CDNA', num_masks=10,
                        schedsamp_k=900.0, train_val_split=0.95, batch_size=32, learning_rate=0.001, validation_interval=200,
                        save_interval=50, debug=0, output_dir='models', event_log_dir='models', pretrained_model='', pretrained_state='')
    for k, v in ref_defaults.items():                           # train_model.py:773-791
        assert getattr(a, k) == v, k
    b = pred_parser().parse_args(['20170101-000000-CDNA-32', 'training-0', '3'])
    assert (b.time_step, b.schedsamp_k, b.context_frames, b.num_masks, b.image_height, b.image_width) == (8, -1, 2, 10, 64, 64)


def test_device_feeder_keeps_iterator_order_shards_and_epochs():
    """The overlapped host feed (dataset.DeviceFeeder, used by train.py) draws from the iterator one batch early; what the training
    loop sees -- batches, their order, each rank's shard, `epoch` at the draw and `is_new_epoch` after it, and NumPy's global RNG
    stream as interleaved with other draws between steps -- must equal the plain synchronous loop of train_model.py:937-950."""
    rs = np.random.RandomState(0)
    N, T, H = 11, 3, 8
    data = ds.group_examples(rs.rand(N, T, H, H, 3).astype(np.float32), rs.randn(N, T, 5).astype(np.float32), rs.randn(N, T, 5).astype(np.float32))
    steps, B, world = 9, 4, 2                                   # 11 sequences, batches of 4: an epoch boundary inside batches 2, 5, 8
    # the plain loop; between two steps something else draws from the global RNG (scheduled sampling does, TM:94)
    np.random.seed(5)
    it = ds.SerialIterator(data, B, repeat=True, shuffle=True)
    plain = []
    for _ in range(steps):
        epoch = it.epoch
        img, act, sta = pivp_amd.concat_examples(it.next())
        plain.append((img, act, sta, epoch, it.is_new_epoch, np.random.rand()))
    for rank in range(world):
        np.random.seed(5)
        it = ds.SerialIterator(data, B, repeat=True, shuffle=True)
        feeder = ds.DeviceFeeder(it, rank=rank, world=world, device='cpu')
        per = B // world
        for t in range(steps):
            x, epoch, new_epoch = feeder.get()
            draw = np.random.rand()                             # the step's own draws come after its batch was drawn ...
            if t + 1 < steps:
                feeder.prefetch()                               # ... and before the next batch is
            img, act, sta, e0, n0, d0 = plain[t]
            sl = slice(rank * per, (rank + 1) * per)
            assert np.array_equal(x[0].numpy(), img[:, sl]) and np.array_equal(x[1].numpy(), act[:, sl]) and np.array_equal(x[2].numpy(), sta[:, sl])
            assert (epoch, new_epoch, draw) == (e0, n0, d0), t
            assert x[0].shape == (T, per, 3, H, H) and x[0].dtype.is_floating_point
    assert [p[4] for p in plain].count(True) == 3
    # a finite iterator ends with StopIteration from get(), after serving every batch
    feeder = ds.DeviceFeeder(ds.SerialIterator(data, 4, repeat=False, shuffle=False), device='cpu')
    n = 0
    with pytest.raises(StopIteration):
        while True:
            x, _, _ = feeder.get(); n += x[0].shape[1]; feeder.prefetch()
    assert n == N
    with pytest.raises(ValueError):
        ds.DeviceFeeder(ds.SerialIterator(data, 3, repeat=True, shuffle=False), rank=0, world=2, device='cpu').get()
