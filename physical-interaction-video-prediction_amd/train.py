"""Training entry point with the reference's flags (train_model.py:772-791) on the MI355X model.

    python -m pivp_amd.train --data_dir ... --output_dir models --num_iterations 100000 --batch_size 32 ...
    python -m torch.distributed.run --nproc-per-node 8 -m pivp_amd.train ...        # data parallel, one process per GPU

Follows the host loop of train_model.py:792-1049: load every sequence listed in <data_dir>/map.csv, 95/5 split by index,
shuffled repeat iterator, optimizer.update per iteration, per-epoch [mean, std, min, max, median] of loss and PSNR, and a
checkpoint directory `<output_dir>/<YYYYmmdd-HHMMSS>-<TYPE>-<B>/` holding `training-<epoch>` (model npz), `state-<epoch>`
(optimizer npz), `training-global_*.npy` and a `version` file.  Defects of the reference OFF the hot path are not reproduced
(SURVEY.md App. D): validation really runs every `validation_interval` epochs, --pretrained_state is loaded into the
optimizer, validation statistics do not overwrite the PSNR file."""
import argparse
import logging
import os
import subprocess
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')    # before torch: this pool's host driver only supports dmabuf IPC (RCCL across processes needs it)

import numpy as np
import torch

from . import dataset as ds
from .checkpoint import load_npz, load_optimizer_npz, save_npz, save_optimizer_npz
from .data import concat_examples
from .model import Model, using_config
from .optimizer import Adam
from .parallel import GradAllReduce


def build_parser():
    p = argparse.ArgumentParser(description='Train the model based on the data saved in ../processed')
    p.add_argument('--data_dir', default='data/processed/brain-robotics-data/push/push_train')
    p.add_argument('--output_dir', default='models')
    p.add_argument('--event_log_dir', default='models')                       # accepted, unused (as in the reference)
    p.add_argument('--num_iterations', type=int, default=100000)
    p.add_argument('--pretrained_model', default='')
    p.add_argument('--pretrained_state', default='')
    p.add_argument('--sequence_length', type=int, default=10)                 # accepted, unused: T comes from the data
    p.add_argument('--context_frames', type=int, default=2)
    p.add_argument('--use_state', type=int, default=1)
    p.add_argument('--model_type', default='CDNA')
    p.add_argument('--num_masks', type=int, default=10)
    p.add_argument('--schedsamp_k', type=float, default=900.0)
    p.add_argument('--train_val_split', type=float, default=0.95)
    p.add_argument('--batch_size', type=int, default=32)
    p.add_argument('--learning_rate', type=float, default=0.001)
    p.add_argument('--gpu', type=int, default=0)
    p.add_argument('--validation_interval', type=int, default=200)
    p.add_argument('--save_interval', type=int, default=50)
    p.add_argument('--debug', type=int, default=0)
    return p


def _git_version():
    try:
        ex = lambda a: subprocess.check_output(['git'] + a, stderr=subprocess.DEVNULL).decode().strip()
        return ex(['rev-parse', '--abbrev-ref', 'HEAD']) + '\n' + ex(['rev-parse', 'HEAD'])
    except Exception:
        return 'unknown'


def main(argv=None):
    args = build_parser().parse_args(argv)
    logging.basicConfig(level=logging.INFO, format='%(asctime)s - %(name)s - %(levelname)s - %(message)s')
    logger = logging.getLogger(__name__)
    world = int(os.environ.get('WORLD_SIZE', '1')); rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(args.gpu)))
    dp = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda:%d' % local_rank))
        dp = GradAllReduce()
    device = 'cuda:%d' % local_rank

    images, actions, states = ds.load_dataset(args.data_dir)
    (tr_i, tr_a, tr_s), (va_i, va_a, va_s) = ds.split_train_val(images, actions, states, args.train_val_split)
    logger.info('Data set contain %d, %d will be use for training and %d will be use for validation', len(images), len(tr_i), len(va_i))
    model = Model(num_masks=args.num_masks, is_cdna=args.model_type == 'CDNA', is_dna=args.model_type == 'DNA',
                  is_stp=args.model_type == 'STP', use_state=args.use_state, scheduled_sampling_k=args.schedsamp_k,
                  num_frame_before_prediction=args.context_frames, prefix='train', device=device, keep_activations=True)
    optimizer = Adam(alpha=args.learning_rate).setup(model, data_parallel=dp)
    if args.pretrained_model:
        load_npz(args.pretrained_model, model)
    per_rank = args.batch_size // world
    if per_rank * world != args.batch_size:
        raise SystemExit('--batch_size must be divisible by the number of ranks')
    np.random.seed(0 if world > 1 else None)      # identical shuffles on every rank; each takes its shard of the batch
    if world > 1:                                 # ... but its own scheduled-sampling draws (rank-offset stream)
        model.sampling_rng = np.random.RandomState(1 + rank)
    train_iter = ds.SerialIterator(ds.group_examples(tr_i, tr_a, tr_s), args.batch_size, repeat=True, shuffle=True)
    valid_iter = ds.SerialIterator(ds.group_examples(va_i, va_a, va_s), args.batch_size, repeat=False, shuffle=True)
    save_dir = os.path.join(args.output_dir, '%s-%s-%d' % (time.strftime('%Y%m%d-%H%M%S'), args.model_type, args.batch_size))
    local_losses, local_psnr, g_loss, g_psnr, g_loss_v, g_psnr_v = [], [], [], [], [], []
    stat = lambda a: [float(np.mean(a)), float(np.std(a)), float(np.min(a)), float(np.max(a)), float(np.median(a))]
    state_loaded = False
    itr, start = 0, None
    # Host feed (TM:937-950 is synchronous): batch t + 1 is drawn, laid out (concat_examples), sharded and copied from pinned memory on a
    # second stream while the GPU works on batch t; iterator order, shards and epoch bookkeeping are those of the plain loop
    feeder = ds.DeviceFeeder(train_iter, rank=rank, world=world, device=device)
    while itr < args.num_iterations:
        x, epoch, is_new_epoch = feeder.get()
        start = start or time.time()
        if world > 1 and itr == 0:                  # replicas start identical: rank 0's lazily initialised weights
            with using_config('train', False):
                model(x, 0)
            import torch.distributed as dist
            model.broadcast_params(src=0); model.reset_state()      # (also invalidates the precision modes' weight packs)
        if args.pretrained_state and not state_loaded:
            with using_config('train', False):
                model(x, 0)
            model.reset_state(); load_optimizer_npz(args.pretrained_state, optimizer); state_loaded = True
        optimizer.update(model, x, itr)             # enqueues the whole step; returns while the GPU is still working on it
        if itr + 1 < args.num_iterations:
            feeder.prefetch()                       # ... so the next batch's host work and copy run underneath it
        stats = torch.stack([model.loss, model.psnr_all]).to(torch.float64)
        if world > 1:                               # the logged statistics are those of the GLOBAL batch (mean over the ranks' shards)
            dist.all_reduce(stats); stats /= world
        lv, pv = stats.tolist()                     # the step's one host synchronisation
        local_losses.append(lv); local_psnr.append(pv)
        model.reset_state()
        if rank == 0:
            logger.info('%d %s', epoch + 1, local_losses[-1])
        if is_new_epoch:
            g_loss.append(stat(local_losses)); g_psnr.append(stat(local_psnr))
            if rank == 0:
                logger.info('[TRAIN] Epoch #: %d  elapsed %.2fs  loss %.6f  psnr %.3f', epoch + 1, time.time() - start, g_loss[-1][0], g_psnr[-1][0])
            local_losses, local_psnr, start = [], [], None
            if (epoch + 1) % args.validation_interval == 0 and len(va_i) > 0:
                vl, vp = [], []
                for vb in valid_iter:
                    vi, va, vs = concat_examples(vb)
                    with using_config('train', False):
                        vl.append(float(model([vi, va, vs], itr))); vp.append(float(model.psnr_all))
                    model.reset_state()
                g_loss_v.append(stat(vl)); g_psnr_v.append(stat(vp))
                valid_iter.reset()
            if epoch % args.save_interval == 0 and rank == 0:
                if not os.path.exists(save_dir):
                    os.makedirs(save_dir)
                    with open(os.path.join(save_dir, 'version'), 'w') as f:
                        f.write(_git_version() + '\n')
                save_npz(os.path.join(save_dir, 'training-' + str(epoch)), model)
                save_optimizer_npz(os.path.join(save_dir, 'state-' + str(epoch)), optimizer, epoch)
                np.save(os.path.join(save_dir, 'training-global_losses'), np.array(g_loss))
                np.save(os.path.join(save_dir, 'training-global_psnr_all'), np.array(g_psnr))
                np.save(os.path.join(save_dir, 'training-global_losses_valid'), np.array(g_loss_v))
                np.save(os.path.join(save_dir, 'training-global_psnr_all_valid'), np.array(g_psnr_v))
        itr += 1
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return save_dir


if __name__ == '__main__':
    main()
