"""On-disk dataset of the reference and the host-side feed of its training loop.

Format (written by the reference's src/data/make_dataset.py:130-158, read at train_model.py:813-834 and
predict_model.py:30-51): `<data_dir>/map.csv`, every field quoted, header
`id,img_bitmap_path,img_np_path,action_np_path,state_np_path,img_bitmap_pred_path,img_np_pred_path`, one row per sequence;
columns 2/3/4 name per-sequence `.npy` files relative to data_dir: images (T,H,W,3) float32 in [0,1], actions (T,5),
states (T,5); column 6 the raw uint8 frames used by predict.  The split is by index, no shuffle (train_model.py:836-843)."""
import csv
import os

import numpy as np

MAP_HEADER = ['id', 'img_bitmap_path', 'img_np_path', 'action_np_path', 'state_np_path', 'img_bitmap_pred_path', 'img_np_pred_path']


def read_map(data_dir):
    """Rows of map.csv without the header; raises ValueError("No file map found") like predict_model.py:37-38."""
    path = os.path.join(data_dir, 'map.csv')
    with open(path, 'r', newline='') as f:
        rows = [r for r in csv.reader(f)]
    if len(rows) <= 1:
        raise ValueError("No file map found")
    return rows[1:]


def write_map(data_dir, rows):
    """Write map.csv the way make_dataset.py:153-158 does (QUOTE_ALL, same header)."""
    with open(os.path.join(data_dir, 'map.csv'), 'w', newline='') as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(MAP_HEADER)
        for r in rows:
            w.writerow(r)


def load_dataset(data_dir):
    """train_model.py:826-834: every sequence into RAM as float32: images (N,T,H,W,3), actions (N,T,5), states (N,T,5)."""
    rows = read_map(data_dir)
    images = np.asarray([np.float32(np.load(os.path.join(data_dir, r[2]))) for r in rows], dtype=np.float32)
    actions = np.asarray([np.float32(np.load(os.path.join(data_dir, r[3]))) for r in rows], dtype=np.float32)
    states = np.asarray([np.float32(np.load(os.path.join(data_dir, r[4]))) for r in rows], dtype=np.float32)
    return images, actions, states


def split_train_val(images, actions, states, train_val_split=0.95):
    """train_model.py:836-843: first floor(split * N) sequences train, the rest validate."""
    k = int(np.floor(train_val_split * len(images)))
    return (images[:k], actions[:k], states[:k]), (images[k:], actions[k:], states[k:])


def group_examples(images, actions, states):
    """train_model.py:899-911: list of [images, actions, states] per sequence (what the iterator serves)."""
    return [[images[i], actions[i], states[i]] for i in range(len(images))]


def get_data_info(data_dir, data_index):
    """predict_model.py:30-51: (image, image_pred, image_bitmap_pred, action, state) of one sequence."""
    rows = read_map(data_dir)
    data_index = int(data_index)
    if data_index > len(rows) - 1:
        raise ValueError("Data index {} is out of range for available data".format(data_index + 1))
    r = rows[data_index]
    image = np.float32(np.load(os.path.join(data_dir, r[2])))
    image_pred = np.float32(np.load(os.path.join(data_dir, r[6])))
    action = np.float32(np.load(os.path.join(data_dir, r[3])))
    state = np.float32(np.load(os.path.join(data_dir, r[4])))
    return image, image_pred, r[5], action, state


class SerialIterator(object):
    """chainer.iterators.SerialIterator (2.0.x) as used at train_model.py:914-915: a permutation drawn from NumPy's global
    RNG at reset, reshuffled in place at every epoch boundary; with repeat=True the last batch of an epoch is completed
    from the start of the next one."""

    def __init__(self, dataset, batch_size, repeat=True, shuffle=True):
        self.dataset = dataset
        self.batch_size = batch_size
        self._repeat = repeat
        self._shuffle = shuffle
        self.reset()

    def reset(self):
        self._order = np.random.permutation(len(self.dataset)) if self._shuffle else None
        self.current_position = 0
        self.epoch = 0
        self.is_new_epoch = False

    def __iter__(self):
        return self

    def __next__(self):
        if not self._repeat and self.epoch > 0:
            raise StopIteration
        i = self.current_position
        i_end = i + self.batch_size
        N = len(self.dataset)
        pick = (lambda a, b: [self.dataset[j] for j in range(a, min(b, N))]) if self._order is None else \
               (lambda a, b: [self.dataset[j] for j in self._order[a:b]])
        batch = pick(i, i_end)
        if i_end >= N:
            if self._repeat:
                rest = i_end - N
                if self._order is not None:
                    np.random.shuffle(self._order)
                if rest > 0:
                    batch.extend(pick(0, rest))
                self.current_position = rest
            else:
                self.current_position = 0
            self.epoch += 1
            self.is_new_epoch = True
        else:
            self.is_new_epoch = False
            self.current_position = i_end
        return batch

    next = __next__


class DeviceFeeder(object):
    """Overlapped host feed of the training loop (the reference's loop, train_model.py:937-950, is synchronous: `concat_examples`
    and the host -> device copy of every batch sit between two `optimizer.update` calls).

    Two slots, each a set of PINNED host buffers and a set of device buffers.  `get()` hands out batch t on the device (the caller's
    stream is made to wait for its copy) together with the iterator's bookkeeping AS IT WAS when batch t was drawn; `prefetch()`,
    called right after the step of batch t has been enqueued, draws batch t + 1 from the iterator, runs `concat_examples` and the
    rank's shard slice into the other slot's pinned buffers and starts the copy on a second stream, so both run under the GPU's work
    on batch t.  The iterator is advanced exactly as a plain `iterator.next()` loop would advance it, one batch early: order of the
    batches, shards, `epoch` and `is_new_epoch` are unchanged (tests/test_dataset_host.py).  With device='cpu' (tests) the slots are
    plain host tensors and the copy is a memcpy.

    Slot reuse: the copy into a slot's DEVICE buffers waits for an event recorded on the caller's stream when the batch AFTER the
    slot's previous occupant was handed out (its step is enqueued before that), the write into its PINNED buffers for the event of
    its previous copy.

    INVARIANT the caller keeps (train.py does): `get()` returns the slot's OWN device buffers -- `Model.__call__` keeps their addresses,
    it does not copy -- and they are overwritten by the copy stream once the `get()` after next has run.  So every kernel that reads
    a batch must have been enqueued on, or joined into, the stream that is current at the following `get()`: `Model.__call__` /
    `Model.backward` / `Adam.update` satisfy that (the plan's side stream is joined into the caller's stream before
    `pivp_rollout_backward` returns, the all-reduce's stream before `update` returns).  A caller that runs the model on ANOTHER stream
    than the one current at `get()`, or that catches a failed `update()` and carries on, must `clone()` the batch (or synchronise the
    device) before it asks for the next but one.  `tests/test_gpu_pipeline.py::test_device_feeder_matches_the_synchronous_loop` runs the
    fed loop against the synchronous one."""

    def __init__(self, iterator, rank=0, world=1, device='cuda:0'):
        import torch
        self._torch = torch
        self.iterator = iterator
        self.rank, self.world = int(rank), int(world)
        self.device = torch.device(device)
        self.cuda = self.device.type == 'cuda'
        self._slots = [None, None]
        self._staged = None           # (slot index, epoch when drawn, is_new_epoch after the draw) of the batch waiting for get()
        self._count = 0
        self._copy_stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self._exhausted = False

    def _slot(self, k, arrays):
        torch = self._torch
        sl = self._slots[k]
        if sl is None or any(tuple(h.shape) != a.shape for h, a in zip(sl['host'], arrays)):
            host = [torch.empty(a.shape, dtype=torch.float32, pin_memory=self.cuda) for a in arrays]
            dev = [torch.empty(a.shape, dtype=torch.float32, device=self.device) for a in arrays] if self.cuda else host
            sl = self._slots[k] = dict(host=host, dev=dev, copied=None, consumed=None)
        return sl

    def prefetch(self):
        """Draw the next batch and start its journey to the device.  No-op when one is already staged or the iterator has ended."""
        if self._staged is not None or self._exhausted:
            return
        from .data import concat_examples
        torch = self._torch
        epoch = self.iterator.epoch
        try:
            batch = self.iterator.next()
        except StopIteration:
            self._exhausted = True
            return
        new_epoch = self.iterator.is_new_epoch
        img, act, sta = concat_examples(batch)
        B = img.shape[1]
        if B % self.world:
            raise ValueError('batch of %d sequences is not divisible by %d ranks' % (B, self.world))
        per = B // self.world
        lo = self.rank * per
        arrays = [a[:, lo:lo + per] for a in (img, act, sta)]
        k = self._count % 2
        self._count += 1
        sl = self._slot(k, arrays)
        if sl['copied'] is not None:
            sl['copied'].synchronize()              # the previous copy OUT of these pinned buffers has finished (two batches ago)
        for h, a in zip(sl['host'], arrays):
            np.copyto(h.numpy(), a)
        if self.cuda:
            with torch.cuda.stream(self._copy_stream):
                if sl['consumed'] is not None:
                    self._copy_stream.wait_event(sl['consumed'])
                for h, d in zip(sl['host'], sl['dev']):
                    d.copy_(h, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._copy_stream)
            sl['copied'] = ev
        self._staged = (k, epoch, new_epoch)

    def get(self):
        """-> ([images (T,B/world,3,H,W), actions, states] on the device, epoch at the draw, is_new_epoch after the draw)."""
        torch = self._torch
        if self._staged is None:
            self.prefetch()
        if self._staged is None:
            raise StopIteration
        k, epoch, new_epoch = self._staged
        self._staged = None
        sl = self._slots[k]
        if self.cuda:
            main = torch.cuda.current_stream(self.device)
            main.wait_event(sl['copied'])
            other = self._slots[1 - k]
            if other is not None:                   # everything enqueued so far used the OTHER slot's batch at the latest
                ev = torch.cuda.Event()
                ev.record(main)
                other['consumed'] = ev
            return list(sl['dev']), epoch, new_epoch
        return [t.clone() for t in sl['host']], epoch, new_epoch
